"""The matrix instructions the register-resident fit loop issues from INLINE ASM (fit_loop.hip, rr_mfma4_*: the accumulator tied in
place) are invisible to the compiler's hazard recogniser: a vector ALU / LDS / memory instruction that reads the accumulator less
than 18 wait states behind the last v_mfma_f64_16x16x4_f64 of a block reads registers the matrix pipe has not written yet.  The
blocks end on their own wait states for that reason; this script checks the generated code: behind every block, the instructions
of the next 18 wait states must not touch the block's accumulator (another in-place product of the same accumulator may).
    python3 tools/check_mfma_hazard.py [file.s]      (default: compiles frank_amd/csrc/fit_loop_rr.hip)
Exit status 1 if a hazard is found.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
WAIT = 18


def regs_of(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def operands(line):
    body = line.split(";")[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0] if parts else "", []
    return parts[0], [t.strip() for t in re.split(r",\s*", parts[1])]


def scan(path, extra_flags=()):
    L = open(path).read().split("\n")
    bad = []
    blocks = 0
    i = 0
    while i < len(L):
        if "#ASMSTART" in L[i]:
            j = i + 1
            mf = []
            while j < len(L) and "#ASMEND" not in L[j]:
                if "v_mfma_f64" in L[j]:
                    mf.append(j)
                j += 1
            if mf:
                blocks += 1
                op, ops = operands(L[mf[-1]])
                acc = regs_of(ops[0])
                # wait states already spent inside the block behind the last product
                spent = 0
                for t in range(mf[-1] + 1, j):
                    o, a = operands(L[t])
                    if o == "s_nop":
                        spent += int(a[0]) + 1
                    elif o and not o.startswith(";"):
                        spent += 1
                t = j + 1
                while spent < WAIT and t < len(L):
                    o, a = operands(L[t])
                    if not o or o.startswith(".") or o.startswith(";") or o.endswith(":"):
                        if o.startswith("s_endpgm"):
                            break
                        t += 1
                        continue
                    if o in ("s_branch", "s_setpc_b64") or o.startswith("s_cbranch"):
                        # (a branch: the fall-through path is checked; a taken branch costs at least as many states)
                        pass
                    touched = set()
                    for tok in a:
                        touched |= regs_of(tok.lstrip("-").lstrip("|").rstrip("|"))
                    if touched & acc and not (o.startswith("v_mfma") and regs_of(a[0]) == acc and regs_of(a[-1]) == acc):
                        bad.append((t + 1, L[t].strip(), spent))
                        break
                    spent += int(a[0]) + 1 if o == "s_nop" else 1
                    t += 1
            i = j
        i += 1
    return blocks, bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1].endswith(".s"):
        path = sys.argv[1]
    else:
        path = "/tmp/fit_loop_rr_hazard.s"
        flags = sys.argv[1:]
        src = os.path.join(HERE, "..", "frank_amd", "csrc", "fit_loop_rr.hip")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I/opt/rocm/include", "-Wno-unused-variable",
                               "-Wno-unused-function", "--cuda-device-only", "-S", src, "-o", path] + flags, stderr=subprocess.DEVNULL)
    blocks, bad = scan(path)
    print("%d blocks of in-place matrix instructions, %d followed by a read of their accumulator within %d wait states" % (blocks, len(bad), WAIT))
    for ln, text, spent in bad[:20]:
        print("  line %d (%d wait states behind the block): %s" % (ln, spent, text))
    sys.exit(1 if bad else 0)
