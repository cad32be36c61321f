"""Build-time check of the hand-written blocks of tile_chol.h: in the device assembly of fit_loop.hip (and lognormal.hip) no
VALU instruction may read the result of a transcendental (v_rcp_f64, v_rsq_f64, v_sqrt_f64, v_exp/log/sin/cos ..) in the slot
directly behind it -- gfx940-class hardware needs one wait state there and the compiler's hazard recogniser does not look into
inline asm.  Prints every violation and exits non-zero if there is one.
    python3 tools/check_trans_hazard.py [file.hip ...]        (default: frank_amd/csrc/fit_loop.hip)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
TRANS = re.compile(r"^\s*(v_rcp_f64|v_rsq_f64|v_sqrt_f64|v_rcp_f32|v_rsq_f32|v_sqrt_f32|v_exp_f32|v_log_f32|v_sin_f32|v_cos_f32|v_rcp_iflag_f32)(_e32|_e64)?\s+(v\[(\d+):(\d+)\]|v(\d+))")


def regs(tok):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def scan(asm_path):
    bad = []
    lines = [l for l in open(asm_path).read().split("\n")]
    instr = [(i, l) for i, l in enumerate(lines) if l.startswith("\t") and not l.lstrip().startswith((".", ";", "//")) and l.strip()]
    for n, (i, l) in enumerate(instr[:-1]):
        m = TRANS.match(l)
        if not m:
            continue
        dst = regs(m.group(3))
        j, nxt = instr[n + 1]
        op = nxt.split()[0]
        if not op.startswith("v_"):
            continue  # s_nop, s_waitcnt, memory, ...: the wait state has passed
        operands = nxt.strip()[len(op):]
        srcs = operands.split(",", 1)[1] if "," in operands else ""
        if regs(srcs) & dst:
            bad.append((i + 1, l.strip(), nxt.strip()))
    return bad


def main():
    files = sys.argv[1:] or [os.path.join(ROOT, "frank_amd", "csrc", "fit_loop.hip")]
    total = 0
    for f in files:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                                   "-I", os.path.join(ROOT, "frank_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-w", f, "-o", out])
            bad = scan(out)
            ntrans = sum(1 for l in open(out) if TRANS.match(l))
        print("%s: %d transcendental instructions, %d read in the next slot" % (os.path.basename(f), ntrans, len(bad)))
        for ln, a, b in bad[:20]:
            print("   line %d: %s  ->  %s" % (ln, a, b))
        total += len(bad)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
