"""bin_gram time against the number of workgroups given to part 0 (development tool; FRANK_AMD_K1_SPLIT per process)."""
import os, subprocess, sys
for split in sys.argv[1:]:
    env = dict(os.environ, FRANK_AMD_K1_SPLIT=split)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "k1_series_alone.py")],
                         env=env, capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("alone")][0]
    vals = sorted(float(x) for x in line.split(":")[1].split())
    print("part-0 workgroups %s: median %.2f ms" % (split, vals[len(vals) // 2]), flush=True)
