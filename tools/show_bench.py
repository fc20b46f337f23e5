"""print the key numbers of one or more bench.py JSON lines: python tools/show_bench.py file.json ..."""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    r = d['roofline']
    dl = (d.get('extra') or {}).get('device_loop') or {}
    print(f"{f}: value {d['value']:.2f} {d['unit']} ({d['ms_per_step']:.1f} ms/step), device_loop {dl.get('value')}, frac {r['frac']:.3f}, "
          f"frac_executed {r['frac_executed']:.3f}, conv share {r['conv_share_of_wall']:.3f}, avg launch {r['avg_launch_ms']:.3f} ms")
    print('   per layer (ms, TF executed):', [(round(p['ms_per_launch'], 3), round(p['TFLOPs_executed'], 1)) for p in r['per_layer']])
    if d.get('cpu_baseline'):
        print('   cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['sample'][:120])
