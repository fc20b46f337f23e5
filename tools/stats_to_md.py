"""rocprofv3 *_kernel_stats.csv -> a small markdown table (top kernels).  usage: stats_to_md.py <csv> "<title>" > out.md"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(f'# {sys.argv[2]}\n')
print('| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|')
for r in rows[:16]:
    print(f"| {r['Name'][:110]} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
          f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
