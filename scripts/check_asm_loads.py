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
        j = i + 1
        while j < len(lines) and not (lines[j].strip().startswith("s_waitcnt") and lines[j - 1].strip() == ";;#ASMSTART"):
            t = lines[j]
            if "ds_read" not in t and re.search(r"\bv\[%d:%d\]|\bv%d\b|\bv%d\b" % (lo, hi, lo, hi), t):
                bad += 1
                print("line %d: %s  is used before its wait: line %d: %s" % (i + 1, line.strip(), j + 1, t.strip()))
            j += 1
        if j - i > 64:
            bad += 1
            print("line %d: %s  has no wait within 64 lines" % (i + 1, line.strip()))
    print("hand-issued row loads: %d, violations: %d" % (loads, bad))
    return 1 if bad or loads == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
