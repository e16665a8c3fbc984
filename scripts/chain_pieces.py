"""Diagnostic (GPU box): the diploid chain's EVENT, piece by piece, on the stage as JTK enters it (refitted model).

Runs the first N pile-ups of cfg 3 through fit_model + estimate_gains + cluster_chunks with a -DJTK_MCMC_STATS build of the
library (JTK_LC_LIB must point at it: the build prints one K2STAT line per chain workgroup), then prints cycles per event by piece
and column count, the base cost per proposal and the slowest chunks.  `--default-model` keeps the default model (the headline's
chains).  `--dump file.npz` stores labels / posteriors / scores for a bit-for-bit diff between two libraries (`--diff a b`).

    JTK_LC_LIB=jtk_amd/_build/exp_k2stats/libjtk_lc_k2stats.so python3 scripts/chain_pieces.py --chunks 600
"""
import argparse
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PAT = re.compile(r"K2STAT chunk (\d+) n (\d+) D (\d+) cyc (\d+) walk \d+ win (\d+) event (\d+) rebuild (\d+) steps (\d+) windows (\d+) "
                 r"events (\d+) accepts (\d+) changed (\d+) head (\d+) bern (\d+) book (\d+) tables (\d+) hops (\d+) exact (\d+)")
COLS = "chunk n D cyc size_moves cyc_size cyc_general steps windows events accepts changed head bern book tables hops exact".split()


def child(args):
    from jtk_amd import api, batch as jb, ffi, synth
    b, cfg = synth.make_batch("ont_diploid", args.chunks)
    p0 = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
    p = p0
    if not args.default_model:
        n = b.chunks["n_reads"].astype(np.int64)
        cov = int(np.sort(n)[len(n) // 2])
        by_id = np.argsort(b.chunks["chunk_id"], kind="stable")
        train = [int(c) for c in by_id if max(cov, 2) - 2 <= n[c] < cov + 2][:5]
        f, r = api.fit_model(p0, b.subset(train), rounds=10)
        p = ffi.Params.from_buffer_copy(bytes(p0))
        p.forward, p.reverse = f, r
        p.gains = api.estimate_gains(f, r)
    if args.solo:   # every listed chunk alone: the chain kernel's time is then that chunk's chain
        for c in [int(x) for x in args.solo.split(",")]:
            best = 1e30
            for _ in range(2):
                o = api.cluster_chunks(p, b.subset([c]))
                best = min(best, api.last_timing()["kernel_ms"]["mcmc"])
            print("SOLO chunk %d n_variants %d mcmc_ms %.2f" % (c, int(o["result"]["n_variants"][0]), best))
        return
    out = api.cluster_chunks(p, b)
    tm = api.last_timing()
    sys.stderr.write("PIECES kernel_ms %s\n" % {k: round(v, 1) for k, v in tm["kernel_ms"].items()})
    if args.dump:
        np.savez(args.dump, label=out["label"], log_post=out["log_post"], score=out["result"]["score"], k=out["result"]["cluster_num"],
                 nv=out["result"]["n_variants"])


def summarize(text, out):
    rows = [[int(x) for x in m.groups()] for m in PAT.finditer(text)]
    if not rows:
        out.write("no K2STAT lines (not a -DJTK_MCMC_STATS library?)\n")
        return
    a = np.array(rows, dtype=float)
    ix = {k: i for i, k in enumerate(COLS)}
    out.write("chain workgroups with stats: %d\n" % len(a))
    for D in sorted(set(a[:, ix["D"]])):
        s = a[a[:, ix["D"]] == D]
        ev = s[:, ix["events"]].sum()
        X = np.stack([s[:, ix["steps"]], s[:, ix["events"]]], 1)
        coef, *_ = np.linalg.lstsq(X, s[:, ix["cyc"]], rcond=None)
        per = lambda k: s[:, ix[k]].sum() / max(ev, 1)
        out.write("D=%d chains=%d: %.1f cycles per proposal + %.0f per event (least squares); %.2f %% of the proposals are events, %.2f %% accepted, "
                  "%.2f %% of the events change the sums | per event: head %.0f, Bernoulli %.0f (exact exp in %.2f %%), accept / flip-back %.0f, tables %.0f, "
                  "hop words / window %.0f = %.0f | max cycles %.3g\n"
                  % (D, len(s), coef[0], coef[1], 100 * ev / s[:, ix["steps"]].sum(), 100 * s[:, ix["accepts"]].sum() / s[:, ix["steps"]].sum(),
                     100 * s[:, ix["changed"]].sum() / max(ev, 1), per("head"), per("bern"), 100 * s[:, ix["exact"]].sum() / max(ev, 1),
                     per("book"), per("tables"), per("hops"), per("head") + per("bern") + per("book") + per("tables") + per("hops"),
                     s[:, ix["cyc"]].max()))
    # per chunk: sum over its chains (20 restarts of each candidate k are ONE workgroup: a K2STAT line is already the chunk)
    order = np.argsort(-a[:, ix["cyc"]])[:8]
    for r in a[order]:
        out.write("slow chunk %d D %d: %.4g cycles, %d events (%.0f cycles each incl. the tables), %d accepted, %d windows\n"
                  % (r[ix["chunk"]], r[ix["D"]], r[ix["cyc"]], r[ix["events"]],
                     (r[ix["cyc_general"]] + r[ix["cyc_size"]]) / max(r[ix["events"]], 1), r[ix["accepts"]], r[ix["windows"]]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=600)
    ap.add_argument("--default-model", action="store_true")
    ap.add_argument("--dump")
    ap.add_argument("--solo", help="comma-separated chunk indices: run each alone and print its chain kernel's ms")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--diff", nargs=2)
    args = ap.parse_args()
    if args.diff:
        x, y = np.load(args.diff[0]), np.load(args.diff[1])
        bad = [k for k in x.files if not np.array_equal(x[k].view(np.uint8), y[k].view(np.uint8))]
        print("DIFF", "bit-equal: " + " ".join(x.files) if not bad else "DIFFERENT: " + " ".join(bad))
        return 1 if bad else 0
    if args.child:
        child(args)
        return 0
    cmd = [sys.executable, os.path.abspath(__file__), "--child", "--chunks", str(args.chunks)]
    if args.default_model:
        cmd.append("--default-model")
    if args.dump:
        cmd += ["--dump", args.dump]
    if args.solo:
        cmd += ["--solo", args.solo]
    pr = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    sys.stderr.write("\n".join(l for l in pr.stderr.splitlines() if l.startswith("PIECES") or "rror" in l) + "\n")
    if pr.returncode != 0:
        sys.stderr.write(pr.stderr[-3000:])
        return pr.returncode
    print("library: %s, %d chunks of cfg 3, %s model" % (os.environ.get("JTK_LC_LIB", "product"), args.chunks,
                                                       "default" if args.default_model else "refitted"))
    if args.solo:
        sys.stdout.write("".join(l + "\n" for l in pr.stdout.splitlines() if l.startswith("SOLO") or "K2STAT" in l or "K2PROD" in l or "K2WAIT" in l))
        return 0
    summarize(pr.stdout, sys.stdout)
    return 0


if __name__ == "__main__":
    sys.exit(main())
