#!/usr/bin/env python3
"""bench.py -- chunks clustered per second on MI355X (BASELINE.json metric).

A "step" is ONE PASS of the hot path (polish -> variant search -> clustering, jtk_lc_session_run, then the fetch of
labels / posteriors / cluster counts / scores to the host) over the whole synthetic dataset, inputs resident in HBM.
The default workload is the configuration BASELINE.json's metric is quoted on: the 5 Mb diploid 60x ONT dataset,
2,500 chunks x 60 reads x 2 kbp (configs[2]; it fits one GPU).  `--gpus N` is STRONG scaling on that fixed dataset:
one process per GPU, the chunks dealt to the ranks longest-processing-time-first (jtk_amd/sharding.py), no data-path
collective, and ONE RCCL all-gather of (label, log_post, cluster_num, score) per step.
`--scaling weak` gives every rank its own `--chunks` chunks instead.

Inside a rank the shard is cut into `--streams` slices, each a resident session with its own HIP stream and host
thread: one slice's pair-HMM passes fill the CUs another slice's chain kernel leaves idle.  Memory is that of ONE copy
of the shard.  There is no barrier between steps inside the timed region (slice i of step s+1 starts when slice i of
step s has been fetched); the region is bracketed by barrier + torch.cuda.synchronize() on both sides.

Prints ONE JSON line on rank 0.  DESIGN.md "Measurement" says how each field is obtained.
"""
import argparse
import hashlib
import json
import os
import queue
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
# torch first: it bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1 (same sonames as /opt/rocm), and the
# process must end up with ONE HIP runtime; libjtk_lc.so then binds to the already loaded one.
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from jtk_amd import api, batch as jb, build as jbuild, ffi, sharding, synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CLOCK_HZ = 2.4e9         # MI355X_MICROARCH.md: max clock
WORKLOADS = {
    "cfg2_ont_diploid_500x60x2kbp": dict(config="ont_diploid", chunks=500),
    "cfg3_ont_diploid_2500x60x2kbp": dict(config="ont_diploid", chunks=2500),
    "cfg4_ont_4copy_2500x160x2kbp": dict(config="ont_4copy", chunks=2500),
    "cfg5_hifi_diploid_2500x40x2kbp": dict(config="hifi_diploid", chunks=2500),
}
# f64 operations of the pair-HMM specification per band cell and pass (an fma counts 2):
# forward 2 emission products + 3 x (mul + 2 fma) = 17, backward the same, 22 fma into the row accumulators = 44
PHMM_FLOP_PER_CELL = 17 + 17 + 44
# the kernels behind the timing families of jtk_lc_last_timing (what rocprofv3 --stats lists); a chain "launch" is
# mcmc_kernel_light on the slice's stream with mcmc_kernel (the general one, few workgroups) beside it on a second stream
KERNELS_OF_FAMILY = {
    "mcmc": ["mcmc_kernel_light", "mcmc_kernel", "chain_split_kernel"],
    "phmm": ["phmm_kernel", "phmm_pair_kernel", "phmm_wide_kernel", "finalize_kernel", "sum_final_kernel"],
    "polish": ["select_edits_kernel", "rethread_kernel", "commit_kernel", "band_prep_kernel"],
    "filter": ["homop_kernel", "chunk_tables_kernel", "column_filter_kernel", "column_filter_fused_kernel", "pick_kernel"],
}


def make_batch_parallel(config, chunk_ids, threads=8):
    cfg = dict(synth.CONFIGS[config])
    with ThreadPoolExecutor(max_workers=threads) as ex:   # ctypes releases the GIL inside jtk_synth_pileup
        pile = list(ex.map(lambda c: synth.make_pileup(int(c), cfg), chunk_ids))
    return jb.pack(pile), cfg


def lib_sha16():
    h = hashlib.sha256()
    with open(ffi.LIB_PATH, "rb") as f:
        h.update(f.read())
    return h.hexdigest()[:16]


def usable_cpus():
    """(CPUs this process may run on, cgroup CPU quota in CPUs or None): os.cpu_count() ignores both."""
    n = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            elif float(txt[0]) > 0:
                quota = float(txt[0]) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    return n, quota


def cpu_baseline(params, batch, budget_s=100.0):
    """The oracle (CPU restatement of the path, OpenMP over chunks like the reference's rayon loop, mod.rs:64-72) on a
    bounded sample of the same workload on this box's host CPUs.  The thread count is a LADDER (1, 8, 32, 128, every CPU the
    process may use: affinity mask and cgroup quota, not os.cpu_count()), four chunks per thread with every thread busy; the
    best rung is `value`, the table shows how the host scales (the reference's own harness pins one thread,
    benchmark_clustering.rs:45-48)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    import helpers
    po = helpers.oracle_params(params)
    n_aff, quota = usable_cpus()
    usable = n_aff if quota is None else max(1, min(n_aff, int(quota + 0.5)))
    ladder = sorted({t for t in (1, 8, 32, 128, usable) if t <= usable} | {usable})
    rows, spent, best, keep = [], 0.0, None, None
    for t in ladder:
        if spent > budget_s and t != usable:
            continue
        n = min(8 if t == 1 else 4 * t, batch.n_chunks, 512)  # a few chunks per thread: start-up and first touch amortised
                                                              # (8 on one thread: with 2 the rung was noise, round 4)
        sub = batch.subset(range(n))
        t0 = time.perf_counter()
        r = O.cluster_chunks(po, sub, skip_polish=False, n_threads=t, want_record=True)
        dt = time.perf_counter() - t0
        spent += dt
        row = dict(threads=t, chunks=n, seconds=round(dt, 2), chunks_per_s=n / dt,
                   mean_record_ms=float(r["record_ms"][:, 0].mean()))
        rows.append(row)
        if best is None or row["chunks_per_s"] > best["chunks_per_s"]:
            best = row
        if keep is None or n > keep[0].n_chunks:
            keep = (sub, r)
    one = rows[0]["chunks_per_s"] if rows[0]["threads"] == 1 else None
    ratio = best["chunks_per_s"] / (best["threads"] * one) if one else None
    return dict(value=best["chunks_per_s"], unit="chunks/s", cores=usable, threads_used=best["threads"], kind="port",
                cpus=dict(os_cpu_count=os.cpu_count(), affinity=n_aff, cgroup_quota=quota),
                scaling=rows, parallel_efficiency_vs_one_thread=ratio,
                sample=f"first min(4 T, 512) chunks of the same dataset for T threads (every thread busy; 8 chunks at T = 1), full "
                       f"path (polish + variant search + clustering); best rung: {best['threads']} threads, {best['chunks']} chunks in "
                       f"{best['seconds']} s, mean RECORD {best['mean_record_ms']:.0f} ms/chunk/thread under that load"
                       + ("" if ratio is None or ratio >= 0.5 else
                          f"; the host scales to {ratio:.2f} of T x one thread: the tables of a chunk (13 MB per read pass) make "
                          f"the restatement DRAM- and allocator-bound when every CPU is busy"),
                one_thread=dict(value=one, unit="chunks/s", cores=1) if one else None), keep[0], keep[1]


def stage_e2e(params0, batch, cfg, device, record_path=None):
    """ONE call of the stage as the pipeline enters it (local_clustering/mod.rs:56-83), from host buffers, with its preambles:
    update_models_on_both_strands (model_tune.rs:96-156: the 5 training pile-ups, TRAIN_ROUND = 10, both strands) ->
    estimate_gain_default (likelihood_gains.rs:186-192) -> the batched clustering_on_pileup loop -> normalize_local_clustering
    (normalize.rs:6-51).  The same calls the C++ host mirror (jtk_amd/csrc/host/local_clustering.hpp) and dataset.py make."""
    ph = {}
    t0 = time.perf_counter()
    n = batch.chunks["n_reads"].astype(np.int64)
    cov = int(np.sort(n)[len(n) // 2])                                           # select_nth_unstable(len / 2)
    by_id = np.argsort(batch.chunks["chunk_id"], kind="stable")
    train = [int(c) for c in by_id if max(cov, 2) - 2 <= n[c] < cov + 2][:5]     # TRAIN_UNIT_SIZE
    f, r = api.fit_model(params0, batch.subset(train), rounds=10, device=device)
    ph["fit_model_ms"] = (time.perf_counter() - t0) * 1e3
    t1 = time.perf_counter()
    p = ffi.Params.from_buffer_copy(bytes(params0))
    p.forward, p.reverse = f, r
    p.gains = api.estimate_gains(f, r, device=device)
    ph["estimate_gains_ms"] = (time.perf_counter() - t1) * 1e3
    t2 = time.perf_counter()
    out = api.cluster_chunks(p, batch, device=device)
    tm = api.last_timing()
    ph["cluster_chunks_ms"] = (time.perf_counter() - t2) * 1e3
    ph["cluster_chunks_detail"] = dict(kernel_ms_summed_over_slices={k: round(v, 1) for k, v in tm["kernel_ms"].items()},
                                       h2d_ms=round(tm["h2d_ms"], 1), d2h_ms=round(tm["d2h_ms"], 1),
                                       mean_polish_rounds=float(out["result"]["polish_rounds"].mean()),
                                       mean_n_variants=float(out["result"]["n_variants"].mean()),
                                       mean_cluster_num=float(out["result"]["cluster_num"].mean()))
    raw = dict(label=out["label"].copy(), log_post=out["log_post"].copy(), score=out["result"]["score"].copy(),
               cluster_num=out["result"]["cluster_num"].copy())   # as the device returned them: what the oracle is compared with
    t3 = time.perf_counter()
    for c in range(batch.n_chunks):
        rr = batch.chunk_reads(c)
        k = int(out["result"]["cluster_num"][c])
        lab, post = api.normalize_pileup(out["label"][rr.start:rr.stop], out["log_post"][rr.start:rr.stop, :k], k)
        out["label"][rr.start:rr.stop] = lab
        out["log_post"][rr.start:rr.stop, :k] = post
    ph["normalize_ms"] = (time.perf_counter() - t3) * 1e3
    ph["total_ms"] = (time.perf_counter() - t0) * 1e3
    # RECORD rows (mod.rs:121: chunk id, elapsed ms, polish ms, consensus length, score, coverage): the device runs chunks in
    # batches, so a chunk's time is its share of the batch's kernel time per family -- pair-HMM and polish by band cells x
    # passes, the chain by proposals
    res = out["result"]
    rows = api.record_rows(batch.chunks["chunk_id"], batch.chunks["n_reads"], batch.chunks["tmpl_len"],
                           [int(batch.read_off[batch.chunk_reads(c).stop] - batch.read_off[batch.chunk_reads(c).start])
                            for c in range(batch.n_chunks)], batch.chunks["copy_num"], res,
                           np.diff(out["cons_off"]).astype(np.int64), tm)
    if record_path:
        with open(record_path, "w") as fh:
            fh.write("\n".join(rows) + "\n")
    ph["record_rows"] = rows[:3]
    ph["record_file"] = record_path
    return ph, out, p, raw


# chunks of the cfg-3 data set whose chains the refitted model makes eventful (profiles/r05_chain_pieces_before.txt: 10^5 .. 10^6
# accepted moves each): the refit-model parity sample of bench.py and tests/golden/cfg3_refit_32.npz name them explicitly
REFIT_EVENTFUL = (196, 206, 269, 332, 367, 464, 478, 498)


def refit_parity(p_refit, batch, raw, n_plain=56):
    """The stage as JTK enters it (refitted model + calibrated gains), checked: the oracle on the SAME parameters for a bounded
    sample -- the first `n_plain` chunks plus the eventful ones -- against what the device returned for them (labels, posterior
    bits, scores, cluster counts)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    import helpers
    ids = sorted(set(range(min(n_plain, batch.n_chunks))) | {c for c in REFIT_EVENTFUL if c < batch.n_chunks})
    sub = batch.subset(ids)
    n_aff, quota = usable_cpus()
    t0 = time.perf_counter()
    ora = O.cluster_chunks(helpers.oracle_params(p_refit), sub, skip_polish=False,
                           n_threads=n_aff if quota is None else max(1, min(n_aff, int(quota + 0.5))))
    rows = np.concatenate([np.arange(batch.chunk_reads(c).start, batch.chunk_reads(c).stop) for c in ids])
    k = ora["result"]["cluster_num"]
    return dict(chunks=len(ids), eventful=[c for c in REFIT_EVENTFUL if c < batch.n_chunks], oracle_seconds=round(time.perf_counter() - t0, 1),
                oracle_rc=int(ora["rc"]),
                labels_equal=bool(np.array_equal(raw["label"][rows], ora["label"])),
                cluster_num_equal=bool(np.array_equal(raw["cluster_num"][ids], k)),
                score_bits_equal=bool(np.array_equal(raw["score"][ids].view(np.uint64), ora["result"]["score"].view(np.uint64))),
                log_post_bits_equal=bool(np.array_equal(raw["log_post"][rows].view(np.uint64), ora["log_post"].view(np.uint64))),
                max_abs_dlogpost=float(np.abs(raw["log_post"][rows] - ora["log_post"]).max()))


def pmc_traffic(workload, sha):
    """HBM bytes per PASS of the workload per kernel family from the rocprofv3 PMC passes committed under profiles/ --
    (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over the profile / the passes in it (FETCH_SIZE doubled per the gfx950
    correction of MI355X_MICROARCH.md) -- but only when they were taken on THIS build of the library (same kernel sources and
    flags, or the same .so), on this workload, and with nothing but full-workload launches in the profile; otherwise None: a
    stale profile says nothing about the kernels being timed.  Every profiles/rNN_pmc_traffic.json is looked at, newest round
    first (round 6: the name was fixed to round 5's, so a line measured on a newer library could only ever say "stale")."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")), reverse=True)
    if not paths:
        return None, "no profiles/rNN_pmc_traffic.json"
    src = jbuild.source_sha16()
    why = None
    for path in paths:
        name = os.path.basename(path)
        try:
            prof = json.load(open(path))
        except (OSError, ValueError):
            continue
        if prof.get("src_sha16") != src and prof.get("lib_sha16") != sha:
            why = why or (f"profiles/{name} was taken on kernel sources {prof.get('src_sha16')} "
                          f"(library {prof.get('lib_sha16')}), this is {src} ({sha})")
            continue
        if prof.get("workload") != workload:
            why = why or f"profiles/{name} is for workload {prof.get('workload')}"
            continue
        if not prof.get("full_workload_launches_only") or "family_bytes_per_pass" not in prof:
            why = why or f"profiles/{name} mixes launches of other batch sizes"
            continue
        return dict(prof["family_bytes_per_pass"]), None
    return None, why


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the first one or two passes of a process still map workspaces and fill the stream pipeline (six slices in flight),
    # and a pass is < 1 s -- ten timed passes after three untimed ones cost 13 s of a run that takes minutes for its other legs
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3_ont_diploid_2500x60x2kbp", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="strong", choices=("strong", "weak"),
                    help="strong: the workload's fixed dataset is sharded over the ranks; weak: every rank its own chunks")
    ap.add_argument("--chunks", type=int, default=0, help="override the dataset size (strong) / chunks per GPU (weak)")
    ap.add_argument("--streams", type=int, default=6,
                    help="slices of the rank's shard in flight, each a resident session on its own HIP stream and host thread")
    ap.add_argument("--stagger-ms", type=float, default=0.0,
                    help="slice i starts its first pass i x this many ms after slice 0 (inside the timed region): slices that "
                         "start together stay in lock step (all in the pair-HMM passes, then all in their chains)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the one-shot (host buffers in, host buffers out) timing")
    ap.add_argument("--no-shard8", action="store_true", help="skip the measurement of one rank's share of an 8-GPU run (N = 1 only)")
    ap.add_argument("--weak-probe", action="store_true",
                    help="N > 1: also take a weak-scaling figure after the strong line is safe (opt-in since round 6: it builds a "
                         "second set of full-size sessions per rank)")
    ap.add_argument("--no-weak-probe", action="store_true", help="(accepted for older command lines; the probe is off by default)")
    ap.add_argument("--fetch", default="all", choices=("all", "results"),
                    help="what a step copies to the host: all = labels, posteriors, records, consensus AND re-threaded ops (every "
                         "output SURVEY 8(d)'s byte formula counts; default since round 6), results = without consensus / ops")
    ap.add_argument("--dump-labels", default="", help="rank 0 writes the whole job's labels, cluster numbers and scores in chunk-id "
                                                      "order to this .npz (tests compare an N-rank run with the 1-rank run)")
    args = ap.parse_args()
    # the bench process owns its GPU: let the library keep the workspaces of a finished one-shot call for the next one
    # (default 32 GB so that it cannot starve other users of the device in a shared process; a 2500-chunk call needs ~95 GB)
    os.environ.setdefault("JTK_LC_POOL_GB", "160")
    # HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and streams that share a queue run in order: a
    # slice's 300 ms chain kernel would hold up another slice's pair-HMM launches.  One queue per slice (+ torch's streams).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(12, 2 * args.streams + 4)))   # two streams per slice + the host's own

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # What the FIRST stage call of a fresh process costs (device workspaces are mapped for the first time: the pipeline's
    # situation), measured where the driver can see it: a child process, run to completion before this one touches the GPU.
    fresh = None
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.no_e2e and os.path.exists(ffi.LIB_PATH):
        import subprocess
        wl0 = WORKLOADS[args.workload]
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "cold_call.py"), str(args.chunks or wl0["chunks"]),
                                wl0["config"], "1"], capture_output=True, text=True, timeout=600)
            rows = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode == 0 and rows:
                c0 = json.loads(rows[-1])["calls"][0]
                fresh = dict(first_call_seconds=c0["seconds"], held_gb=c0["held_gb"], h2d_ms=c0["h2d_ms"], d2h_ms=c0["d2h_ms"],
                             note="scripts/cold_call.py as a child process before this one touched the GPU: one "
                                  "jtk_lc_cluster_chunks on the whole workload from host buffers, the process's first call")
            else:
                fresh = dict(error=(r.stderr or r.stdout)[-300:])
        except Exception as e:  # the bench line must not depend on it
            fresh = dict(error=repr(e))
    # diagnostic / CPU-box tests: JTK_BENCH_BACKEND=gloo runs the multi-rank path with every rank on GPU 0
    backend = os.environ.get("JTK_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    if args.gpus != world:
        if "WORLD_SIZE" in os.environ:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus disagree")
        # `python3 bench.py --gpus N` as typed, without a launcher: start one rank per GPU as CHILD processes (nothing here has
        # touched the GPU yet; a process that has must never be replaced by another), relay their output -- rank 0 prints the
        # JSON line -- and return their exit code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stderr.write("bench.py: launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
        raise SystemExit(subprocess.call(cmd, env=env))
    # The library is built by __graft_entry__.build() BEFORE the bench (never inside a timed or profiled process, never
    # by N ranks at once).  Only a missing library is built here, by local rank 0.
    if not os.path.exists(ffi.LIB_PATH) or not os.path.exists(ffi.SYNTH_LIB_PATH):
        if local_rank == 0:
            jbuild.build()
        else:
            while not (os.path.exists(ffi.LIB_PATH) and os.path.exists(ffi.SYNTH_LIB_PATH)):
                time.sleep(1.0)
    elif rank == 0 and jbuild.is_stale():
        sys.stderr.write("bench.py: WARNING: a source is newer than libjtk_lc.so; timing the library as built "
                         "(run __graft_entry__.build() first)\n")
    sha = lib_sha16()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a gfx950 GPU (torch sees none)")
    torch.cuda.set_device(local_rank)
    torch.cuda.synchronize()
    if not ffi.lib().jtk_lc_device_ok(local_rank):
        raise SystemExit("bench.py needs a gfx950 GPU: " + ffi.lib().jtk_lc_last_error().decode())

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wl = WORKLOADS[args.workload]
    cfg0 = synth.CONFIGS[wl["config"]]
    n_total = args.chunks or wl["chunks"]
    reads_per_chunk = cfg0["n_haps"] * cfg0["reads_per_hap"]
    if args.scaling == "strong":
        parts = sharding.strong_shards(n_total, reads_per_chunk, cfg0["tmpl_len"], cfg0["copy_num"], world)
    else:
        parts = [np.array(sharding.weak_chunk_ids(r, n_total), dtype=np.int64) for r in range(world)]
        n_total = n_total * world
    my_ids = parts[rank]
    batch, cfg = make_batch_parallel(wl["config"], my_ids)
    params = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
    stride = batch.post_stride

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- a shard whose WORKSPACES do not fit the device beside each other (cfg 4: 2,500 x 160 reads x 296 KB of row sums /
    #      tables + scratch) cannot be held as resident sessions: its step is the library's one-shot call, which streams slices
    #      through a bounded workspace (session.hip run_once).  Inputs then cross PCIe inside the step (~1 GB: noise next to the
    #      step), and the line says so.
    cap_len = int(cfg["tmpl_len"]) + int(cfg["tmpl_len"]) // 8 + 64
    est_bytes = float(batch.n_reads) * (cap_len + 1) * 144.0 + args.streams * 14e9   # 16 f64 + ops / deltas per row; scratch
    total_mem = float(torch.cuda.get_device_properties(local_rank).total_memory)
    if est_bytes > 0.80 * total_mem:
        def one_step():
            r = api.cluster_chunks(params, batch, device=local_rank)
            return r, api.last_timing()
        for _ in range(args.warmup):
            one_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out, tm = one_step()
        barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        value = n_total * args.steps / elapsed
        alg = batch.algorithmic_bytes(k_per_chunk=out["result"]["cluster_num"]) / max(1, batch.n_chunks)
        achieved = value * alg / 1e9
        km = tm["kernel_ms"]
        if rank == 0:
            print(json.dumps(dict(
                metric="chunks clustered/sec (whole node), 60x ONT 2kbp chunks", value=value, unit="chunks/s", n_gpus=world,
                steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True,
                scaling=args.scaling, vs_baseline=None, dtype="f64", data="synthetic",
                config=dict(workload=args.workload, chunks_total=int(n_total), chunks_this_rank=int(batch.n_chunks),
                            reads_per_chunk=int(reads_per_chunk), chunk_len=int(cfg["tmpl_len"]), copy_num=int(cfg["copy_num"]),
                            band_frac=cfg["band_frac"], sharding="1 gpu" if world == 1 else f"{args.scaling}: LPT over {world} ranks",
                            step="ONE-SHOT: jtk_lc_cluster_chunks from host buffers per step (the workspaces of this shard, "
                                 f"~{est_bytes / 1e9:.0f} GB, do not fit the device as resident sessions; the library streams slices "
                                 "through a bounded workspace); inputs cross PCIe inside the step"),
                roofline=dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBPS * world, unit="GB/s",
                              frac=achieved / (HBM_PEAK_GBPS * world), traffic=None, algorithmic_bytes_per_chunk=alg,
                              kernel_ms_summed_over_slices=km,
                              secondary=dict(chain_ms_share=km["mcmc"] / max(1e-9, sum(km.values())))),
                chunks_ok=int((out["result"]["status"] == 0).sum()), lib_sha16=sha,
                mean_polish_rounds=float(out["result"]["polish_rounds"].mean()),
                mean_cluster_num=float(out["result"]["cluster_num"].mean()), cpu_baseline=None)))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- the rank's shard as `streams` resident slices (equal shares of the reads, contiguous chunk ranges)
    n_streams = max(1, min(args.streams, batch.n_chunks))
    bounds = [round(i * batch.n_chunks / n_streams) for i in range(n_streams + 1)]
    slices = [batch.subset(range(bounds[i], bounds[i + 1])) for i in range(n_streams)]
    sessions = [api.Session(params, sl, device=local_rank) for sl in slices]
    read_bounds = np.cumsum([0] + [sl.n_reads for sl in slices])
    gather = None
    if dist is not None:
        sizes = [(len(p) * reads_per_chunk, len(p)) for p in parts]
        gather = sharding.ResultGather(dist, sizes, stride,
                                       device=torch.device("cuda", local_rank) if backend == "nccl" else None)
    ktime = {n: 0.0 for n in ffi.KERNEL_NAMES}
    klaunch = {n: 0 for n in ffi.KERNEL_NAMES}
    state = dict(dev_ms=0.0, gathered=None, last=None)
    fetch_bufs = [None] * n_streams   # a slice's output arrays, allocated by its first fetch and overwritten by the later ones

    def run_steps(n_steps, timed, serial=False):
        """n_steps passes over the shard.  serial: the slices run one after another (kernel breakdown pass)."""
        done = queue.Queue()

        def worker(i):
            try:
                if not serial and args.stagger_ms > 0.0 and i > 0:
                    time.sleep(i * args.stagger_ms / 1e3)
                for s in range(n_steps):
                    sessions[i].run(skip_polish=False)   # returns when the device has finished this slice's pass
                    t = api.last_timing()                # thread-local: the pass this thread just ran
                    if args.fetch == "all":              # every output of the drop-in call -> host (inside the step)
                        fetch_bufs[i] = out = sessions[i].fetch(out=fetch_bufs[i])
                    else:                                # labels, posteriors, k, score only (rounds 1-5)
                        out = sessions[i].fetch_results()
                    done.put((s, i, t, dict(label=out["label"].copy(), log_post=out["log_post"].copy(), result=out["result"].copy(),
                                            cons_bytes=int(out["cons_off"][-1]) if "cons_off" in out else 0,
                                            ops_bytes=int(out["ops_out_off"][-1]) if "ops_out_off" in out else 0)))
            except BaseException as e:  # noqa: BLE001 -- handed to the main thread
                done.put((-1, i, e, None))

        if serial:
            for i in range(n_streams):
                worker(i)
            threads = []
        else:
            threads = [threading.Thread(target=worker, args=(i,)) for i in range(n_streams)]
            for th in threads:
                th.start()
        pending = {}
        for _ in range(n_steps * n_streams):
            s, i, t, out = done.get()
            if s < 0:
                raise t
            pending.setdefault(s, {})[i] = out
            if timed:
                state["dev_ms"] += t["total_ms"]
                for n in ffi.KERNEL_NAMES:
                    ktime[n] += t["kernel_ms"][n]
                    klaunch[n] += t["kernel_launches"][n]
            if len(pending[s]) == n_streams:             # step s is complete on this rank
                outs = [pending[s][j] for j in range(n_streams)]
                del pending[s]
                merged = dict(label=np.concatenate([o["label"] for o in outs]),
                              log_post=np.concatenate([o["log_post"] for o in outs]),
                              result=np.concatenate([o["result"] for o in outs]),
                              cons_bytes=sum(o["cons_bytes"] for o in outs), ops_bytes=sum(o["ops_bytes"] for o in outs))
                state["last"] = merged
                if gather is not None:                   # the only exchange of the path: one all-gather per step
                    state["gathered"] = gather.gather(merged["label"], merged["log_post"],
                                                      merged["result"]["cluster_num"], merged["result"]["score"])
        for th in threads:
            th.join()

    for _ in range(args.warmup):
        run_steps(1, timed=False)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, timed=True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    value = n_total * args.steps / elapsed
    out = state["last"]
    ok = int((out["result"]["status"] == 0).sum())
    overl_k = {n: ktime[n] / args.steps for n in ffi.KERNEL_NAMES}
    overl_slice_ms = state["dev_ms"] / (args.steps * n_streams)

    # ---- a serial pass (slices one after another, nothing overlapped): per-kernel device time that adds up
    for n in ffi.KERNEL_NAMES:
        ktime[n], klaunch[n] = 0.0, 0
    state["dev_ms"] = 0.0
    torch.cuda.synchronize()
    ts = time.perf_counter()
    run_steps(1, timed=True, serial=True)
    serial_ms = (time.perf_counter() - ts) * 1e3
    serial_k = dict(ktime)
    serial_launch = dict(klaunch)
    serial_out = state["last"]
    steps_agree = bool(np.array_equal(serial_out["label"], out["label"]))
    # ---- where the chain kernels' time went on this rank (include/jtk_lc_debug.h): a chain launch lasts as long as its slowest
    #      chunk, so the line names it -- id, its share of the launch, how many of its proposals could not be stepped over
    cyc = np.concatenate([s_.chain_profile()[0] for s_ in sessions]).astype(np.float64)
    evs = np.concatenate([s_.chain_profile()[1] for s_ in sessions]).astype(np.int64)
    worst = int(np.argmax(cyc)) if len(cyc) else 0
    my_profile = dict(
        rank=rank, chunks=int(batch.n_chunks), pair_hmm_ms=serial_k["phmm"], polish_ms=serial_k["polish"], filter_ms=serial_k["filter"],
        chain_ms_summed_over_slices=serial_k["mcmc"], chain_launch_ms_mean=serial_k["mcmc"] / max(1, serial_launch["mcmc"]),
        slowest_chunk_id=int(my_ids[worst]) if len(cyc) else None, slowest_chunk_chain_ms=float(cyc[worst] / CLOCK_HZ * 1e3) if len(cyc) else None,
        slowest_chunk_events=int(evs[worst]) if len(cyc) else None,
        median_chain_ms=float(np.median(cyc[cyc > 0]) / CLOCK_HZ * 1e3) if (cyc > 0).any() else None,
        chunks_with_a_chain=int((cyc > 0).sum()), events_total=int(evs.sum()))
    per_rank_profiles = [my_profile]
    if dist is not None:
        per_rank_profiles = [None] * world
        dist.all_gather_object(per_rank_profiles, my_profile)

    # ---- roofline: achieved = chunks/s (whole job) x algorithmic bytes per chunk; frac against 8 TB/s per GPU
    alg_bytes_shard = batch.algorithmic_bytes(k_per_chunk=out["result"]["cluster_num"])
    alg_bytes_per_chunk = alg_bytes_shard / max(1, batch.n_chunks)
    achieved = value * alg_bytes_per_chunk / 1e9
    dom = max(serial_k, key=lambda n: serial_k[n])
    traffic, traffic_note = pmc_traffic(args.workload, sha)
    # secondary bounds: f64 rate of the banded pair-HMM and the chain's cycles per proposal
    res = out["result"]
    passes = np.minimum(res["polish_rounds"].astype(np.int64) + (res["polish_rounds"] >= 20), 21)
    cells = 0.0
    for c in range(batch.n_chunks):
        ch = batch.chunks[c]
        r0, r1 = int(ch["read_first"]), int(ch["read_first"]) + int(ch["n_reads"])
        diag = int(ch["n_reads"]) * int(ch["tmpl_len"]) + int(batch.read_off[r1] - batch.read_off[r0])   # sum of L + n
        radius = int(np.ceil(int(ch["tmpl_len"]) * cfg["band_frac"])) // 2
        cells += float(passes[c]) * diag * (2 * radius + 1)
    phmm_s = serial_k["phmm"] / 1e3
    k_tried = np.maximum(1, np.minimum(int(cfg["copy_num"]), 1 + 2 * res["n_variants"].astype(np.int64)) - 1)
    proposals_max = float((20 * 2000 * batch.chunks["n_reads"].astype(np.int64) * k_tried * (res["n_variants"] > 0)).max())
    mcmc_launches = max(1, serial_launch["mcmc"])
    secondary = dict(
        fp64_tflops=(cells * PHMM_FLOP_PER_CELL / phmm_s / 1e12) if phmm_s > 0 else None,
        fp64_peak_tflops=78.6,
        chain_cycles_per_proposal=(serial_k["mcmc"] / mcmc_launches * 1e-3 * CLOCK_HZ / proposals_max) if proposals_max else None,
        note="fp64_tflops: f64 operations of the pair-HMM specification (78 per band cell and pass, fma = 2) over the "
             "serial pass's pair-HMM device time, vs the 78.6 TFLOP/s vector f64 peak; chain_cycles_per_proposal: the "
             "chain kernel's launch duration (set by its slowest chunk) x 2.4 GHz / that chunk's 20 x 2000 x n x "
             "(#k tried) proposals")
    # the pair-HMM family's forward state: (toM, toD) of every anti-diagonal, 1 KiB, written by the forward and read by the
    # backward sweep of every read and pass -- what the kernel moves BY CONSTRUCTION, next to what the counters saw
    stripe_bytes = 0.0
    for c in range(batch.n_chunks):
        ch = batch.chunks[c]
        r0, r1 = int(ch["read_first"]), int(ch["read_first"]) + int(ch["n_reads"])
        diags = int(ch["n_reads"]) * (int(ch["tmpl_len"]) + 24) + int(batch.read_off[r1] - batch.read_off[r0])
        radius = int(np.ceil(int(ch["tmpl_len"]) * cfg["band_frac"])) // 2
        if 15 <= radius <= 30:
            # phmm_kernel (round 4): the ends of a sweep (about 2r + 24 diagonals each) stream a pair per diagonal, the groups
            # in between one 2 KiB checkpoint per 8 diagonals (the pairs are replayed from it)
            ends = min(diags, int(ch["n_reads"]) * (4 * radius + 48))
            diags = ends + (diags - ends) / 4.0
        stripe_bytes += float(passes[c]) * diags * 1024.0 * 2.0
    launches_dom = max(1, serial_launch[dom])
    fam_pass = (traffic or {}).get("phmm")
    stream = dict(
        family="phmm", kernels=KERNELS_OF_FAMILY.get("phmm"), serial_ms_per_pass=serial_k["phmm"],
        stripe_bytes_per_pass_by_construction=stripe_bytes,
        by_construction=dict(GBps=stripe_bytes / 1e9 / phmm_s if phmm_s > 0 else None,
                             frac_of_8TBps=stripe_bytes / 1e9 / phmm_s / HBM_PEAK_GBPS if phmm_s > 0 else None,
                             frac_of_achievable_6p3TBps=stripe_bytes / 1e9 / phmm_s / 6300.0 if phmm_s > 0 else None),
        pmc_bytes_per_pass=fam_pass,
        pmc=None if not fam_pass else dict(GBps=fam_pass / 1e9 / phmm_s, frac_of_8TBps=fam_pass / 1e9 / phmm_s / HBM_PEAK_GBPS,
                                           frac_of_achievable_6p3TBps=fam_pass / 1e9 / phmm_s / 6300.0),
        note="the forward state the pair-HMM family moves through HBM, next to the 124 KB per chunk of the algorithmic figure: "
             "bytes per pass of the workload over the serial pass's pair-HMM device time.  by_construction = sum over chunks of "
             "passes x (anti-diagonals of its reads: 1 KiB each at the ends of a sweep, a 2 KiB checkpoint per 8 in between "
             "since round 4) x (write + read back); pmc = (2 x FETCH_SIZE + WRITE_SIZE) of the family's kernels from the "
             "committed profile of this build (tables and row sums included); 6.3 TB/s = the float4-copy rate of "
             "MI355X_MICROARCH.md.  With the replay the family is bound by vector-instruction issue (secondary.fp64_tflops, "
             "DESIGN.md section 6), not by this stream")
    roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBPS * world, unit="GB/s",
                    frac=achieved / (HBM_PEAK_GBPS * world),
                    traffic=(traffic[dom] / launches_dom) if traffic and dom in traffic else None, traffic_note=traffic_note,
                    traffic_unit="HBM bytes per launch of the dominant family = its bytes per pass / its launches per pass",
                    stream=stream,
                    algorithmic_bytes_per_chunk=alg_bytes_per_chunk,
                    dominant_kernel=dict(
                        name=dom, kernels=KERNELS_OF_FAMILY.get(dom), launches_per_pass=serial_launch[dom], ms_per_pass=serial_k[dom],
                        avg_launch_ms=serial_k[dom] / max(1, serial_launch[dom]),
                        # per-launch form: the algorithmic bytes of the chunks one launch processes / its mean duration
                        achieved_GBps_per_launch=(alg_bytes_shard / n_streams / 1e9) /
                        (serial_k[dom] / max(1, serial_launch[dom]) / 1e3) if serial_k[dom] > 0 else None),
                    serial_pass=dict(wall_ms=serial_ms, kernel_ms=serial_k, kernel_launches=serial_launch,
                                     kernel_ms_sum=sum(serial_k.values()),
                                     note="slices run one after another: the kernel times add up to <= wall_ms"),
                    overlapped_kernel_ms_per_step=overl_k,
                    hbm_traffic_per_pass_by_family=traffic, secondary=secondary,
                    note="achieved = value x algorithmic bytes per chunk (SURVEY.md 8d: packed 4-bit bases, 2-bit ops, u32 "
                         "labels, f64 posteriors), frac = achieved / (8 TB/s x n_gpus).  Byte/integer + f64 scan work: the "
                         "compulsory HBM traffic is ~126 KB/chunk, so the HBM fraction is tiny by construction; the binding "
                         "limits are the `secondary` ones.  traffic = PMC-measured HBM bytes per launch of the dominant "
                         "kernel family, only when the committed profile was taken on this exact library build")

    line = dict(metric="chunks clustered/sec (whole node), 60x ONT 2kbp chunks", value=value, unit="chunks/s",
                n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3,
                higher_is_better=True, scaling=args.scaling, vs_baseline=None, dtype="f64", data="synthetic",
                config=dict(workload=args.workload, chunks_total=int(n_total), chunks_this_rank=int(batch.n_chunks),
                            reads_per_chunk=int(reads_per_chunk), chunk_len=int(cfg["tmpl_len"]),
                            copy_num=int(cfg["copy_num"]), band_frac=cfg["band_frac"],
                            sharding=(f"{args.scaling}: chunks dealt LPT to {world} ranks, one all-gather of "
                                      "(label, log_post, k, score) per step over RCCL") if world > 1 else "1 gpu",
                            slices_in_flight=n_streams,
                            step=("one pass over the dataset incl. the copy of EVERY output to the host: labels, posteriors, k, score, "
                                  "consensus and re-threaded ops" if args.fetch == "all" else
                                  "one pass over the dataset incl. fetch of labels, posteriors, k, score (no consensus / ops)"),
                            fetched_bytes_per_step=dict(consensus=int(out.get("cons_bytes", 0)), ops=int(out.get("ops_bytes", 0)),
                                                        labels=int(out["label"].nbytes), log_post=int(out["log_post"].nbytes))),
                roofline=roofline, slice_pass_latency_ms=overl_slice_ms,
                serial_step_agrees=steps_agree, chunks_ok=ok, lib_sha16=sha,
                mean_polish_rounds=float(out["result"]["polish_rounds"].mean()),
                mean_cluster_num=float(out["result"]["cluster_num"].mean()))
    line["per_rank"] = per_rank_profiles   # (N > 1: every rank's serial-pass breakdown and its slowest chain)
    if gather is not None and rank == 0:
        g = state["gathered"]
        line["gathered_reads"] = int(sum(len(x["label"]) for x in g))
        line["gather_ok"] = bool(np.array_equal(g[0]["label"], out["label"]))
    if args.dump_labels and rank == 0:
        # the whole job in chunk-id order: rank r's chunks are parts[r] (ascending), reads_per_chunk reads each
        per_rank = state["gathered"] if gather is not None else [dict(label=out["label"], cluster_num=out["result"]["cluster_num"],
                                                                      score=out["result"]["score"])]
        lab = np.zeros(n_total * reads_per_chunk, dtype=np.uint32)
        kk = np.zeros(n_total, dtype=np.uint32)
        sc = np.zeros(n_total, dtype=np.float64)
        for r, g in enumerate(per_rank):
            ids = np.asarray(parts[r] if args.scaling == "strong" else list(parts[r]), dtype=np.int64)
            lab.reshape(n_total, reads_per_chunk)[ids] = g["label"].reshape(len(ids), reads_per_chunk)
            kk[ids] = g["cluster_num"]
            sc[ids] = g["score"]
        np.savez(args.dump_labels, label=lab, cluster_num=kk, score=sc)

    for s in sessions:
        s.close()
    # ---- N > 1: the weak-scaling figure next to the strong one.  Strong scaling of the fixed 2,500-chunk data set ends on each
    #      shard's slowest chain; a data set that grows with the machine has no such term.  Every rank
    #      therefore also times a full-size private share: its own shard repeated N times (the same mix of pile-ups, 2,500
    #      chunks per GPU; nothing new to synthesise), same step, no gather.
    if world > 1 and args.scaling == "strong" and args.weak_probe and not args.no_weak_probe:
        # The strong-scaling line above is complete; nothing here may lose it.  Every rank says whether it could build its sessions
        # BEFORE the first timed barrier (an all-reduced flag: one rank out of memory must not leave the others waiting in a
        # collective), and any exception becomes line["weak_probe"] = {"error": ...}.
        sess_w = []
        try:
            api.trim_cache(local_rank)   # the strong run's workspaces may still sit in the block pool
            ok_w, err_w = 1.0, None
            try:
                wb = batch.subset(np.tile(np.arange(batch.n_chunks), world))
                nw = max(1, min(args.streams, wb.n_chunks))
                bw = [round(i * wb.n_chunks / nw) for i in range(nw + 1)]
                sess_w = [api.Session(params, wb.subset(range(bw[i], bw[i + 1])), device=local_rank) for i in range(nw)]
            except Exception as e:  # noqa: BLE001
                ok_w, err_w = 0.0, repr(e)
            flag = torch.tensor([ok_w], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) < 1.0:
                line["weak_probe"] = dict(error=err_w or "another rank could not build its sessions")
            else:
                def run_w(k):
                    errs = []

                    def w(i):
                        try:
                            for _ in range(k):
                                sess_w[i].run(skip_polish=False)
                                sess_w[i].fetch_results()
                        except Exception as e:  # noqa: BLE001
                            errs.append(repr(e))
                    ths = [threading.Thread(target=w, args=(i,)) for i in range(nw)]
                    for th in ths:
                        th.start()
                    for th in ths:
                        th.join()
                    return errs

                steps_w = max(1, min(args.steps, 3))
                errs = run_w(1)
                barrier()
                tw = time.perf_counter()
                errs += run_w(steps_w)
                barrier()
                el_w = time.perf_counter() - tw
                tt = torch.tensor([el_w, float(wb.n_chunks), float(len(errs))], dtype=torch.float64,
                                  device="cuda" if backend == "nccl" else "cpu")
                mx = tt.clone()
                dist.all_reduce(mx, op=dist.ReduceOp.MAX)
                sm = tt.clone()
                dist.all_reduce(sm, op=dist.ReduceOp.SUM)
                if float(sm[2].item()) > 0:
                    line["weak_probe"] = dict(error="; ".join(errs) or "a step failed on another rank")
                else:
                    line["weak_probe"] = dict(
                        chunks_per_gpu=int(wb.n_chunks), steps=steps_w, ms_per_step=float(mx[0].item()) / steps_w * 1e3,
                        chunks_per_s=float(sm[1].item()) * steps_w / float(mx[0].item()),
                        note="weak scaling beside the strong figure: every rank's own shard repeated n_gpus times (2,500 chunks per "
                             "GPU, the same mix of pile-ups), same step without the gather, max over ranks; value / this = what the "
                             "tail of the fixed data set costs at this N")
        except Exception as e:  # noqa: BLE001 -- e.g. a collective that failed: the strong line still goes out
            line["weak_probe"] = dict(error=repr(e))
        for s in sess_w:
            try:
                s.close()
            except Exception:  # noqa: BLE001
                pass
    # ---- what ONE rank of an 8-GPU run would do (north_star's 8 x MI355X target; the pool gives this process one GPU): shard
    #      0 of the 8-way LPT partition of the same dataset, as slices on this GPU, same step definition.  A measured per-GPU
    #      rate, not a scaling curve: no RCCL, no second device.
    if world == 1 and args.scaling == "strong" and not args.no_shard8 and n_total >= 64:
        parts8 = sharding.strong_shards(n_total, reads_per_chunk, cfg0["tmpl_len"], cfg0["copy_num"], 8)
        steps8 = max(1, min(args.steps, 2))
        per_rank = []
        for r8 in range(8):    # every rank's shard, one after the other on THIS GPU: the job's step is the slowest rank's
            pos = np.searchsorted(np.asarray(my_ids), parts8[r8])        # world == 1: my_ids is 0..n_total-1
            sh = batch.subset(pos)
            nb = max(1, min(args.streams, sh.n_chunks))
            bd = [round(i * sh.n_chunks / nb) for i in range(nb + 1)]
            sess8 = [api.Session(params, sh.subset(range(bd[i], bd[i + 1])), device=local_rank) for i in range(nb)]

            def run8(k):
                def w(i):
                    for _ in range(k):
                        sess8[i].run(skip_polish=False)
                        sess8[i].fetch_results()
                ths = [threading.Thread(target=w, args=(i,)) for i in range(nb)]
                for th in ths:
                    th.start()
                for th in ths:
                    th.join()

            run8(1)
            torch.cuda.synchronize()
            t8 = time.perf_counter()
            run8(steps8)
            torch.cuda.synchronize()
            per_rank.append(dict(rank=r8, chunks=int(sh.n_chunks), ms_per_step=(time.perf_counter() - t8) / steps8 * 1e3))
            for s8 in sess8:
                s8.close()
        worst = max(per_rank, key=lambda x: x["ms_per_step"])
        dt8 = worst["ms_per_step"] / 1e3
        line["shard8_projection"] = dict(
            chunks=worst["chunks"], slices=min(args.streams, worst["chunks"]), ms_per_step=worst["ms_per_step"],
            ms_per_step_by_rank=[round(x["ms_per_step"], 1) for x in per_rank], chunks_by_rank=[x["chunks"] for x in per_rank],
            chunks_per_s_one_gpu=worst["chunks"] / dt8, projected_8gpu_chunks_per_s=n_total / dt8,
            projected_efficiency_vs_8x=(n_total / dt8) / (8 * value), steps=steps8,
            note="each rank's share of the 8-way LPT partition run on THIS GPU, one after the other (same step, incl. fetch); the "
                 "job's step is the SLOWEST rank's (max over ranks, as bench.py times N > 1), so the 8-GPU figure is n_chunks / "
                 "that; the one all-gather per step (~3 MB) is left out.  NOT a measured scaling curve: no run of this code on "
                 "more than one GPU exists")
    # ---- end to end: what one stage call costs a host that hands over HOST buffers (encode + allocate + H2D + run +
    #      fetch + free): jtk_lc_cluster_chunks on this rank's shard.  PCIe-inclusive; never `value`.
    if not args.no_e2e:
        api.trim_cache(local_rank)
        t1 = time.perf_counter()
        one = api.cluster_chunks(params, batch, device=local_rank)
        cold = time.perf_counter() - t1
        t2 = time.perf_counter()
        one = api.cluster_chunks(params, batch, device=local_rank, out=one)   # (a host keeps its output buffers between calls)
        warm = time.perf_counter() - t2
        tm = api.last_timing()
        line["e2e"] = dict(chunks_per_s=batch.n_chunks / warm, seconds=warm, first_call_seconds=cold, fresh_process=fresh,
                           h2d_ms=tm["h2d_ms"], d2h_ms=tm["d2h_ms"], pool_gb=float(os.environ["JTK_LC_POOL_GB"]), matches_resident=bool(np.array_equal(one["label"], out["label"])),
                           note="jtk_lc_cluster_chunks on this rank's shard from host buffers to host buffers; the first call "
                                "also maps the device workspaces, the second reuses the pooled blocks")
        # ---- the whole stage call with its preambles (refit + gains calibration + clustering + normalisation), cold and warm
        api.trim_cache(local_rank)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        cold_ph, _, _, _ = stage_e2e(params, batch, cfg, local_rank)
        warm_ph, st_out, p_refit, st_raw = stage_e2e(params, batch, cfg, local_rank, os.path.join(ROOT, "gpurun_out", "record_r06.tsv"))
        line["stage_e2e"] = dict(
            cold=cold_ph, warm=warm_ph, chunks_per_s_warm=batch.n_chunks / (warm_ph["total_ms"] / 1e3),
            chunks_ok=int((st_out["result"]["status"] == 0).sum()),
            note="one LocalClustering::local_clustering_selected call on this rank's shard from host buffers: model refit (10 rounds "
                 "x 5 pile-ups x 2 strands) + gains calibration + jtk_lc_cluster_chunks + normalisation; cold = first call "
                 "(maps the device workspaces), warm = the next one.  The refitted model differs from the resident runs' default "
                 "model: its results are checked against the oracle on the same parameters in parity_on_refit_sample")
    api.trim_cache(local_rank)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb, sub, ora = cpu_baseline(params, batch)
        line["cpu_baseline"] = cb
        # the checker doing its job on the sample: labels bit-exact, posteriors within 1e-4
        nr = int(sub.n_reads)
        line["parity_on_cpu_sample"] = dict(
            chunks=int(sub.n_chunks), labels_equal=bool(np.array_equal(out["label"][:nr], ora["label"])),
            max_abs_dlogpost=float(np.abs(out["log_post"][:nr] - ora["log_post"]).max()))
        if not args.no_e2e:
            line["parity_on_refit_sample"] = refit_parity(p_refit, batch, st_raw)
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
