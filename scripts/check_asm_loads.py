"""Static check of the hand-issued LDS loads of mcmc_chain_tab's rejected-step loop (jtk_amd/csrc/mcmc_kernels.hip).

The loop loads a proposal's row with `ds_read_b64` from inline asm -- invisible to hipcc's wait-count scoreboard -- and waits for it
with a hand-placed `s_waitcnt lgkmcnt(1)` / `lgkmcnt(0)`.  What the compiler must not do is touch the destination registers between
the load and the wait that covers it.  This script compiles the file to assembly (no GPU needed) and checks every such load of every
instantiation:  python3 scripts/check_asm_loads.py      exit code 0 = every load is followed by its wait before any use."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jtk_amd import build  # noqa: E402


def main():
    src = os.path.join(build.CSRC, "mcmc_kernels.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "mcmc.s")
        cmd = [build.HIPCC] + build.FLAGS + ["-x", "hip", "--cuda-device-only", "-S", src, "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    loads = bad = 0
    for i, line in enumerate(lines):
        if not (line.strip().startswith("ds_read_b64") and lines[i - 1].strip() == ";;#ASMSTART"):
            continue
        m = re.match(r"\s*ds_read_b64 v\[(\d+):(\d+)\], v(\d+)", line)
        lo, hi = int(m.group(1)), int(m.group(2))
        loads += 1
        # Scan to the wait that COVERS this load: the first hand-placed lgkmcnt(0), or the first hand-placed
        # lgkmcnt(1) that follows a LATER hand-issued load (then this one is no longer the youngest).  Until there the compiler must
        # neither touch the destination registers nor issue an LDS / scalar-memory operation of its own: lgkmcnt counts those too,
        # and scalar loads return out of order, so "all but the youngest" would stop meaning what the hand-placed wait assumes.
        j = i + 1
        later_reads, covered = 0, False
        while j < len(lines) and j - i <= 96:
            t = lines[j].strip()
            hand = lines[j - 1].strip() == ";;#ASMSTART"
            if hand and t.startswith("s_waitcnt"):
                mm = re.search(r"lgkmcnt\((\d+)\)", t)
                n = int(mm.group(1)) if mm else 0
                if n == 0 or (n == 1 and later_reads >= 1):
                    covered = True
                    break
            elif hand and t.startswith("ds_read"):
                later_reads += 1
            else:
                if re.search(r"\bv\[%d:%d\]|\bv%d\b|\bv%d\b" % (lo, hi, lo, hi), t) and not t.startswith(";"):
                    bad += 1
                    print("line %d: %s  is used before its wait: line %d: %s" % (i + 1, line.strip(), j + 1, t))
                op = t.split()[0] if t and not t.startswith((";", ".")) and not t.endswith(":") else ""
                if not hand and (op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("ds_")):
                    bad += 1
                    print("line %d: %s  has a compiler-issued %s in flight beside it (line %d)" % (i + 1, line.strip(), op, j + 1))
            j += 1
        if not covered:
            bad += 1
            print("line %d: %s  has no covering wait within 96 lines" % (i + 1, line.strip()))
    print("hand-issued row loads: %d, violations: %d" % (loads, bad))
    return 1 if bad or loads == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
