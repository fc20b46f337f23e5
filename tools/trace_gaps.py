import csv, sys, glob, collections
kt = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
mc = glob.glob(sys.argv[1] + '/**/*memory_copy_trace.csv', recursive=True)
rows = []
for r in csv.DictReader(open(kt)):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
for f in mc:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
rows.sort()
# sampling calls: split at gaps > 300 us before a graph_count... simpler: print every gap > 20 us in the last 30 % of the trace
t0 = rows[0][0]
n = len(rows)
last_end = rows[0][1]
big = []
for i, (s, e, k) in enumerate(rows):
    if s - last_end > 15000 and i > n * 0.6:
        big.append((s - last_end, (s - t0) / 1e6, rows[i - 1][2], k))
    last_end = max(last_end, e)
for g in big[:80]:
    print('gap %.1f us at %.2f ms  after %-50s before %s' % (g[0] / 1e3, g[1], g[2], g[3]))
