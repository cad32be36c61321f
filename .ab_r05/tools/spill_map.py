"""Where a kernel touches scratch memory: every scratch_load / scratch_store of one kernel of a `hipcc -S` listing with the
innermost loop (label .. backward branch) that contains it and that loop's size, barriers and matrix instructions.
    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S frank_amd/csrc/fit_loop.hip -o /tmp/fit_loop.s
    python3 tools/spill_map.py /tmp/fit_loop.s fit_loop_kernelILi0ELi0E
"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(key), l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []  # (head, tail)
for i, l in enumerate(body):
    m = re.search(r"\b(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if m and m.group(2) in labels and labels[m.group(2)] <= i:
        loops.append((labels[m.group(2)], i))
def is_instr(l):
    return l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;")
def stats(a, b):
    seg = body[a:b + 1]
    return dict(n=sum(is_instr(l) for l in seg), mfma=sum("v_mfma" in l for l in seg), bar=sum("s_barrier" in l for l in seg),
                sl=sum("scratch_load" in l for l in seg), ss=sum("scratch_store" in l for l in seg))
tot = stats(0, len(body) - 1)
print("kernel %s: %d instructions, %d mfma, %d barriers, %d scratch_load, %d scratch_store" % (key, tot["n"], tot["mfma"], tot["bar"], tot["sl"], tot["ss"]))
rows = {}
for i, l in enumerate(body):
    if "scratch_load" in l or "scratch_store" in l:
        inner = None
        for a, b in loops:
            if a <= i <= b and (inner is None or (b - a) < (inner[1] - inner[0])):
                inner = (a, b)
        depth = sum(a <= i <= b for a, b in loops)
        rows.setdefault((inner, depth), []).append(i)
for (inner, depth), idx in sorted(rows.items(), key=lambda kv: (kv[0][0] or (-1, -1))):
    if inner is None:
        print("  outside any loop: %d scratch instructions (lines %d..%d of the kernel)" % (len(idx), idx[0], idx[-1]))
    else:
        s = stats(*inner)
        print("  loop lines %d..%d (depth %d; %d instr, %d mfma, %d barriers): %d scratch instructions (%d loads, %d stores)" % (
            inner[0], inner[1], depth, s["n"], s["mfma"], s["bar"], len(idx),
            sum("scratch_load" in body[i] for i in idx), sum("scratch_store" in body[i] for i in idx)))
