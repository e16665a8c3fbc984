#!/usr/bin/env python3
"""Static census of the gfx950 assembly of one source file of the library -- no GPU needed.

    python3 scripts/isa_census.py mcmc_kernels.hip                 # every function: size, instruction mix, scalar spills
    python3 scripts/isa_census.py mcmc_kernels.hip -f chain_tabILi3 # + every loop of the matching functions
    python3 scripts/isa_census.py phmm_sweep.hip -D JTK_PHMM_MARKS   # + the regions between `; MARK x` comments

What it is for: a lone wave pays ~5 cycles per instruction whatever the unit, so the instruction
count of a chain's step IS its cost -- and two things the counters do not show are visible here: scalar registers spilled to
lanes of a vector register inside a loop (v_writelane / v_readlane pairs), and an `s_waitcnt lgkmcnt(0)` that follows the issue of
a fresh LDS load inside a loop, i.e. a "prefetch" that waits for itself (hipcc waits with 0 at loop back edges where lgkmcnt(1)
would do)."""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jtk_amd import build  # noqa: E402


def classify(op, line):
    if op.startswith("v_"):
        if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
            return "lane"
        if "dpp" in line:
            return "v_dpp"
        if "f64" in op:
            return "v_f64"
        return "v_other"
    if op.startswith("s_"):
        if op.startswith("s_cbranch") or op.startswith("s_branch"):
            return "branch"
        if op.startswith("s_waitcnt"):
            return "waitcnt"
        if op.startswith("s_nop") or op.startswith("s_sleep"):
            return "nop"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def instructions(lines, a, b):
    for i in range(a, b):
        t = lines[i].strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":") or t.startswith(";;#"):
            continue
        yield i, t.split()[0], t


def census(lines, a, b):
    c = collections.Counter()
    for _, op, t in instructions(lines, a, b):
        c[classify(op, t)] += 1
    return c


def fmt(c):
    order = ["v_f64", "v_other", "v_dpp", "lane", "salu", "branch", "lds", "vmem", "waitcnt", "nop", "other"]
    return "%5d  " % sum(c.values()) + " ".join("%s %d" % (k, c[k]) for k in order if c[k])


def self_waits(lines, a, b, window=14):
    """`s_waitcnt lgkmcnt(0)` within `window` instructions after a ds_read was issued (nothing older can be the reason unless
    the compiler could not tell): the load's whole round trip is exposed."""
    out = []
    recent = []
    for i, op, t in instructions(lines, a, b):
        if op.startswith("ds_read") or op.startswith("ds_bpermute"):
            recent.append(i)
        elif op == "s_waitcnt" and "lgkmcnt(0)" in t:
            hits = [j for j in recent if i - j <= window * 2]
            if len(hits) >= 2:   # two loads in flight and a wait for both: the older one alone would have been lgkmcnt(1)
                out.append((i + 1, [j + 1 for j in hits]))
            recent = []
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source", help="a file under jtk_amd/csrc/")
    ap.add_argument("-f", "--func", default="", help="substring of the (mangled) names to list loops for")
    ap.add_argument("-D", action="append", default=[], help="extra -D macros (e.g. JTK_PHMM_MARKS, JTK_MCMC_STATS)")
    ap.add_argument("--asm", default="", help="keep the assembly here")
    args = ap.parse_args()
    src = os.path.join(build.CSRC, os.path.basename(args.source))
    with tempfile.TemporaryDirectory() as tmp:
        out = args.asm or os.path.join(tmp, "out.s")
        cmd = [build.HIPCC] + build.FLAGS + ["-D" + d for d in args.D] + ["-x", "hip", "--cuda-device-only", "-S", src, "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    funcs = []
    for i, line in enumerate(lines):
        m = re.match(r"\s*\.type\s+(\S+),@function", line)
        if m:
            funcs.append([m.group(1), i, None])
        m = re.match(r"\s*\.size\s+(\S+),", line)
        if m:
            for f in funcs:
                if f[0] == m.group(1) and f[2] is None:
                    f[2] = i
    print("%-64s %6s  %s" % ("function", "lines", "instructions by class | scalar spill stores / reloads"))
    for name, a, b in funcs:
        if b is None:
            continue
        c = census(lines, a, b)
        sp_w = sum(1 for _, op, t in instructions(lines, a, b) if op.startswith("v_writelane"))
        sp_r = sum(1 for _, op, t in instructions(lines, a, b) if op.startswith("v_readlane") and re.search(r", v\d+, \d+$", t))
        print("%-64s %6d  %s | %d / %d" % (name[-64:], b - a, fmt(c), sp_w, sp_r))
        if args.func and args.func in name:
            heads = [i for i in range(a, b) if "Loop Header: Depth=" in lines[i]]
            for k, h in enumerate(heads):
                depth = int(re.search(r"Depth=(\d+)", lines[h]).group(1))
                # the loop's text extent: up to the next header of the same or a smaller depth
                e = b
                for h2 in heads[k + 1:]:
                    if int(re.search(r"Depth=(\d+)", lines[h2]).group(1)) <= depth:
                        e = h2
                        break
                lc = census(lines, h, e)
                print("    loop at line %6d depth %d (to %6d): %s" % (h + 1, depth, e, fmt(lc)))
                for w, loads in self_waits(lines, h, e):
                    print("        line %d: s_waitcnt lgkmcnt(0) with the loads of lines %s in flight" % (w, loads))
            marks = [i for i in range(a, b) if "; MARK " in lines[i]]
            for m1, m2 in zip(marks, marks[1:]):
                print("    %-24s .. %-24s %s" % (lines[m1].split("MARK")[1].strip(), lines[m2].split("MARK")[1].strip(),
                                                 fmt(census(lines, m1, m2))))
    return 0


if __name__ == "__main__":
    sys.exit(main())
