#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into the small JSON files kept under profiles/.

    python3 tools/pmc_summary.py OUT.json DIR [DIR ...]

Every DIR is the -d directory of one rocprofv3 run (--kernel-trace with --pmc, separate passes).  For each kernel name
(template arguments kept, namespaces and parameter lists dropped) and counter: mean value per dispatch and the number of
dispatches; from the kernel trace: mean duration.  HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md:
FETCH_SIZE (KB) is doubled (gfx950 reports half the bytes of wide streaming reads), WRITE_SIZE (KB) is taken as is."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)


def kernel_stats(db_path, csv_path):
    """The --stats summary (per-kernel calls / total / average) of a rocpd database as the CSV rocprofv3 would write."""
    import sqlite3
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(csv_path, "w") as fh:
        fh.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage"\n')
        for name, calls, tot, avg, pct in rows:
            fh.write('"%s",%d,%d,%.1f,%.4f\n' % (name, calls, tot, avg, pct))
    print("wrote", csv_path, len(rows), "kernels")


def main():
    if sys.argv[1] == "--stats":
        return kernel_stats(sys.argv[2], sys.argv[3])
    out_path, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(float)  # (dispatch, kernel, counter) -> sum over rows (one row per instance dimension)
            times = {}
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = short(r["Kernel_Name"])
                    per[(r["Dispatch_Id"], k, r["Counter_Name"])] += float(r["Counter_Value"])
                    times[(r["Dispatch_Id"], k)] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            for (disp, k, cname), v in per.items():
                acc[k][cname].append(v)
            for (disp, k), t in times.items():
                dur[k].append(t)
        # rocprofv3's default output here is a rocpd SQLite database: the pmc_events view carries the same columns
        for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
            import sqlite3
            db = sqlite3.connect(f)
            per = defaultdict(float)
            times = {}
            for disp, name, cname, val, dt in db.execute(
                    "select dispatch_id, name, counter_name, counter_value, duration from pmc_events"):
                k = short(name)
                per[(disp, k, cname)] += float(val)
                times[(disp, k)] = float(dt) * 1e-6
            for (disp, k, cname), v in per.items():
                acc[k][cname].append(v)
            for (disp, k), t in times.items():
                dur[k].append(t)
            db.close()
    res = {}
    for k in sorted(acc):
        e = {}
        for cname, vals in sorted(acc[k].items()):
            e[cname] = sum(vals) / len(vals)
            e["launches_" + cname] = len(vals)
        if dur[k]:
            e["duration_ms_mean_under_pmc"] = sum(dur[k]) / len(dur[k])
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = int(2 * 1024 * e.get("FETCH_SIZE", 0.0) + 1024 * e.get("WRITE_SIZE", 0.0))
        res[k] = e
    res["_how"] = ("rocprofv3 --kernel-trace --pmc <counters> (one pass per counter group) ; values are means per "
                   "dispatch; hbm_bytes_per_launch = 2 x FETCH_SIZE KB + WRITE_SIZE KB (gfx950 FETCH_SIZE correction)")
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    names = [k for k in res if not k.startswith("_")]
    print("wrote", out_path, len(names), "kernels:", ", ".join(k for k in names if len(k) < 60)[:600])


if __name__ == "__main__":
    main()
