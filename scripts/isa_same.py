#!/usr/bin/env python3
"""Are the product kernels of a source file the same machine code as at a git revision?  Compiles jtk_amd/csrc/<file> of the
working tree and of <rev> (default HEAD; sources and headers from `git archive`) to gfx950 assembly with the product's flags and
compares every function's instructions (labels normalised).  Used when a change must not touch the hot kernels -- e.g. the
recording instantiations behind jtk_lc_session_trace: `same` for every product kernel, `NEW` for the added ones.  No GPU needed.

    python3 scripts/isa_same.py mcmc_kernels.hip filter_kernels.hip [--rev HEAD]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jtk_amd import build  # noqa: E402


def functions(path):
    out, cur, body = {}, None, []
    for line in open(path):
        m = re.match(r"^([A-Za-z_][\w.$]*):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            if cur:
                out[cur] = body
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if t.startswith(".Lfunc_end"):
                out[cur] = body
                cur = None
            continue
        t = re.sub(r";.*$", "", t).strip()
        body.append(re.sub(r"\.L[\w.$]+", ".L", t))
    if cur:
        out[cur] = body
    return {k: v for k, v in out.items() if not k.startswith("__hip_cuid") and not k.startswith("amdhsa.")}


def assemble(tree, src, out):
    flags = [f for f in build.FLAGS if not f.startswith("-I") and f != "-Wall"]
    cmd = [build.HIPCC] + flags + ["-I" + os.path.join(tree, "include"), "-I" + os.path.join(tree, "jtk_amd", "csrc"),
                                   "--cuda-device-only", "-S", "-x", "hip", os.path.join(tree, "jtk_amd", "csrc", src), "-o", out]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sources", nargs="+")
    ap.add_argument("--rev", default="HEAD")
    a = ap.parse_args()
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        old = os.path.join(tmp, "old")
        os.makedirs(old)
        ar = subprocess.Popen(["git", "-C", ROOT, "archive", a.rev, "jtk_amd/csrc", "include"], stdout=subprocess.PIPE)
        subprocess.check_call(["tar", "-x", "-C", old], stdin=ar.stdout)
        for src in a.sources:
            so, sn = os.path.join(tmp, "o_" + src + ".s"), os.path.join(tmp, "n_" + src + ".s")
            assemble(old, src, so)
            assemble(ROOT, src, sn)
            fo, fn = functions(so), functions(sn)
            print("== %s: working tree against %s" % (src, a.rev))
            for k in sorted(fo):
                if k not in fn:
                    print("GONE", k[:120], len(fo[k]))
                    bad += 1
                elif fo[k] != fn[k]:
                    print("DIFF", k[:120], len(fo[k]), len(fn[k]))
                    bad += 1
                else:
                    print("same", k[:120], len(fo[k]))
            for k in sorted(fn):
                if k not in fo:
                    print("NEW ", k[:120], len(fn[k]))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
