"""Summarise the rocprofv3 csv output of tools/profile_round.sh:
   <out>/kernel_stats.md   per-kernel table of the --kernel-trace pass
   <out>/pmc_traffic.json  per-launch means of the PMC passes for the fused conv kernel (+ the gfx950 FETCH correction)."""
import csv, glob, json, os, sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
KERNEL = os.environ.get('DDK_PROFILE_KERNEL', 'conv_x2_kernel<true, true, false')      # the score model's conv layers in the default two-limb / three-product form (the two head launches are conv_x2_kernel<false, ...>); DDK_CONV_KERNEL=3 runs: conv_x3_kernel<true, true, false


def find(d, suffix):
    r = glob.glob(os.path.join(d, '**', '*' + suffix), recursive=True)
    return r[0] if r else None


# ---- kernel trace
f = find(os.path.join(src, 'trace'), 'kernel_trace.csv')
rows = list(csv.DictReader(open(f)))
agg = defaultdict(list)
for r in rows:
    agg[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in agg.values())
lines = ['| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    lines.append(f'| {k[:92]} | {len(v)} | {sum(v) / 1e3:.2f} | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} | {100 * sum(v) / tot:.1f} |')
conv = [v for k, v in agg.items() if KERNEL in k]
conv = conv[0] if conv else []
head = (f'# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-device-loop --no-extras --no-timesplit (MI355X)\n\n'
        f'{len(conv)} fused TP-conv launches, average {sum(conv) / max(len(conv), 1):.1f} us.\n\n')
open(os.path.join(out, 'kernel_stats.md'), 'w').write(head + '\n'.join(lines) + '\n')

# ---- counters
import hashlib
_root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
_h = hashlib.sha256()
for _n in ('k_conv_x.hip', 'k_conv_x2.hip', 'k_conv_x_epi_gen.inc', 'k_conv_x_epi2_gen.inc', 'k_conv_common.h', 'ddk_internal.h'):          # == bench.py CONV_KERNEL_SOURCES: bench.py quotes this profile only for these sources
    _h.update(open(os.path.join(_root, 'disco_diffdock_amd', 'csrc', _n), 'rb').read())
res = {'kernel_source_sha256': _h.hexdigest(), 'command': 'rocprofv3 --kernel-trace --pmc <group> (one group per pass) -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-device-loop --no-extras --no-timesplit',
       'kernel': 'ddk::' + KERNEL + ', false>', 'avg_launch_us_kernel_trace': sum(conv) / max(len(conv), 1)}
for d in sorted(glob.glob(os.path.join(src, 'pmc*'))):
    if not os.path.isdir(d):
        continue
    f = find(d, 'counter_collection.csv')
    if not f:
        continue
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if KERNEL in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        res[k + '_mean'] = sum(v) / len(v)
        res['launches'] = len(v)
if 'FETCH_SIZE_mean' in res and 'WRITE_SIZE_mean' in res:
    res['correction'] = ('gfx950: FETCH_SIZE (KB) tallies 64 B per 128-B request of a wide coalesced read -> doubled '
                         '(MI355X_MICROARCH.md, HBM section); WRITE_SIZE (KB) as reported')
    res['traffic_bytes_per_launch'] = (2 * res['FETCH_SIZE_mean'] + res['WRITE_SIZE_mean']) * 1024
if 'SQ_VALU_MFMA_BUSY_CYCLES_mean' in res and 'GRBM_GUI_ACTIVE_mean' in res:
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
    res['MfmaUtil_percent'] = 100 * (res['SQ_VALU_MFMA_BUSY_CYCLES_mean'] / 1024) / (res['GRBM_GUI_ACTIVE_mean'] / 8)
if 'SQ_WAVE_CYCLES_mean' in res:
    for k in ('SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY'):
        if k + '_mean' in res:
            res[k + '_over_WAVE_CYCLES'] = res[k + '_mean'] / res['SQ_WAVE_CYCLES_mean']
if 'TCC_HIT_sum_mean' in res and 'TCC_MISS_sum_mean' in res:
    res['L2_hit_rate'] = res['TCC_HIT_sum_mean'] / (res['TCC_HIT_sum_mean'] + res['TCC_MISS_sum_mean'])
json.dump(res, open(os.path.join(out, 'pmc_traffic.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
