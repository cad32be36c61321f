"""Kernel-by-kernel timeline of pipelined steps from a rocprofv3 --kernel-trace database (rocpd sqlite):
    python tools/step_gpu_timeline.py <results.db> [first_kernel_substring]
prints, for the last few steps, each dispatch on the binning stream with its duration and the idle gap in front of it."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
view = "kernels" if "kernels" in tabs else kd[0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(cur.execute("select %s, start, end, queue_id from %s order by start" % (name_col, view))) if "queue_id" in cols else \
    [r + (0,) for r in cur.execute("select %s, start, end from %s order by start" % (name_col, view))]
# the binning stream = the queue of deproject_kernel
q = [r[3] for r in rows if "deproject" in r[0]]
q = q[-1] if q else 0
seq = [r for r in rows if r[3] == q]
starts = [i for i, r in enumerate(seq) if "deproject" in r[0]]
if len(starts) < 4:
    print("too few steps")
    sys.exit(0)
a, b = starts[-3], starts[-2]
t0 = seq[a][1]
prev_end = seq[a - 1][2] if a > 0 else t0
print("step of %d dispatches, %.1f us from its first kernel to the next step's first" % (b - a, (seq[b][1] - t0) / 1e3))
busy = 0
for r in seq[a:b]:
    gap = (r[1] - prev_end) / 1e3
    dur = (r[2] - r[1]) / 1e3
    busy += dur
    print("%9.1f us  gap %7.1f  dur %7.1f  %s" % ((r[1] - t0) / 1e3, gap, dur, r[0][:70]))
    prev_end = r[2]
print("busy %.1f us" % busy)

# the fit loop launches of the timed region (the last `nlast` steps), relative to the first deproject of that region
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if len(starts) >= nlast:
    tz = seq[starts[-nlast]][1]
    print("timed region: first deproject at 0, last deproject at %.2f ms" % ((seq[starts[-1]][1] - tz) / 1e6))
    for r in rows:
        if "fit_loop_kernel" in r[0] and r[2] > tz:
            print("fit_loop launch: start %8.2f ms  end %8.2f ms" % ((r[1] - tz) / 1e6, (r[2] - tz) / 1e6))
