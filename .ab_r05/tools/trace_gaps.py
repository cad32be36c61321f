import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 400 kernels before the last fit_loop launch
idx = [i for i, r in enumerate(rows) if 'fit_loop_kernel' in r['Kernel_Name']]
last = idx[-1]
seg = rows[max(0, last - 330):last + 1]
t0 = int(seg[0]['Start_Timestamp'])
prev_end = None
for r in seg[-80:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print("%9.1f us  dur %7.1f  gap %6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r.get('Queue_Id', '?'), r['Kernel_Name'][:60]))
    prev_end = e
