"""print the key numbers of one or more bench.py JSON lines: python tools/show_bench.py file.json ..."""
import json, sys
for f in sys.argv[1:]:
    d = json.loads([ln for ln in open(f).read().splitlines() if ln.startswith('{')][-1])
    r = d['roofline']
    e = d.get('extra') or {}
    dl = e.get('device_loop') or {}
    print(f"{f}: value {d['value']:.2f} {d['unit']} ({d['ms_per_step']:.1f} ms/step), device_loop {dl.get('value')}, frac {r['frac']:.3f}, "
          f"fp32-equivalent {r.get('fp32_equivalent_TFLOPs', 0):.1f} TF, conv share {r['conv_share_of_wall']:.3f}, avg launch {r['avg_launch_ms']:.4f} ms, traffic {r.get('traffic')}")
    print('   per layer (ms, TF executed):', [(round(p['ms_per_launch'], 3), round(p.get('mfma_TFLOPs', 0), 1)) for p in r['per_layer']])
    for k in ('pruning_off', 'pocket_bound'):
        if e.get(k):
            print(f"   {k}: {e[k]['value']:.2f}")
    o = e.get('other_limb_form')
    if o:
        print(f"   conv_kernel = {o['conv_kernel']} ({o['limb_products']} limb products): {o['value']:.2f}, avg launch {o['avg_launch_ms']:.4f} ms, frac {o['frac_of_f16_matrix_peak']:.3f}")
    if d.get('cpu_baseline'):
        print('   cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:120])
