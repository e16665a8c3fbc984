#!/usr/bin/env python3
"""bench.py -- chunks clustered per second on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (polish -> variant search -> clustering, jtk_lc_session_run) over one
resident batch of synthetic pile-ups.  At N = 1 the workload is BASELINE.json configs[1]: a synthetic 1 Mb
2-haplotype region, 60x ONT error model, 500 chunks x 2 kbp, copy number 2.  With N > 1 every rank (one
process per GPU) clusters its own 500 chunks (weak scaling, chunk ids rank*500..), there is no data-path
collective, and the cluster labels are all-gathered over RCCL at the end of every step.

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement" for how each field is obtained.
"""
import argparse
import json
import os
import sys
import queue
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

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
WORKLOADS = {
    "cfg2_ont_diploid_500x60x2kbp": dict(config="ont_diploid", chunks=500),
    "cfg3_ont_diploid_2500x60x2kbp": dict(config="ont_diploid", chunks=2500),
    "cfg4_ont_4copy_2500x160x2kbp": dict(config="ont_4copy", chunks=2500),
    "cfg5_hifi_diploid_2500x40x2kbp": dict(config="hifi_diploid", chunks=2500),
}


def make_batch_parallel(config, n_chunks, first_chunk_id, threads=8):
    cfg = dict(synth.CONFIGS[config])
    with ThreadPoolExecutor(max_workers=threads) as ex:   # ctypes releases the GIL inside jtk_synth_pileup
        pile = list(ex.map(lambda c: synth.make_pileup(first_chunk_id + c, cfg), range(n_chunks)))
    return jb.pack(pile), cfg


def cpu_baseline(params, batch, sample_chunks, threads):
    """The oracle (CPU restatement, OpenMP over chunks like the reference's rayon loop) on a bounded sample of
    the same workload, on this box's host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    import helpers
    sub = batch.subset(range(sample_chunks))
    po = helpers.oracle_params(params)
    t0 = time.perf_counter()
    r = O.cluster_chunks(po, sub, skip_polish=False, n_threads=threads, want_record=True)
    dt = time.perf_counter() - t0
    return dict(value=sample_chunks / dt, unit="chunks/s", cores=threads, kind="port",
                sample=f"first {sample_chunks} chunks of the same batch, full path (polish+search+clustering), "
                       f"{dt:.1f} s wall, mean RECORD {float(r['record_ms'][:, 0].mean()):.0f} ms/chunk"), sub, r


def pmc_traffic(family, workload, n_chunks):
    """HBM bytes per launch of the dominant kernel family from the rocprofv3 PMC passes committed under profiles/
    (scripts/profile_bench.sh; FETCH_SIZE doubled per the gfx950 correction).  None when no profile of this
    workload is on file."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        prof = json.load(open(path))
    except OSError:
        return None
    if prof.get("workload") != workload or n_chunks != WORKLOADS[workload]["chunks"]:
        return None
    total = 0.0
    for k in prof["kernel_family"].get(family, []):
        e = prof["kernels"].get(k)
        if e:
            total += (2.0 * e.get("FETCH_SIZE_KiB_per_launch", 0.0) + e.get("WRITE_SIZE_KiB_per_launch", 0.0)) * 1024.0
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2_ont_diploid_500x60x2kbp", choices=sorted(WORKLOADS))
    ap.add_argument("--chunks", type=int, default=0, help="override chunks per GPU (diagnostic runs only)")
    ap.add_argument("--streams", type=int, default=4,
                    help="resident batches in flight, each on its own HIP stream and host thread (1 = strictly serial steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="chunks in the CPU baseline sample (0 = auto)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # diagnostic only: JTK_BENCH_BACKEND=gloo runs the multi-rank path with every rank on GPU 0 (1-GPU boxes)
    backend = os.environ.get("JTK_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    jbuild.build()
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
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)

    wl = WORKLOADS[args.workload]
    n_chunks = args.chunks or wl["chunks"]
    batch, cfg = make_batch_parallel(wl["config"], n_chunks, first_chunk_id=sharding.weak_chunk_ids(rank, n_chunks)[0])
    params = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # A step is one pass of the hot path over one resident 500-chunk batch.  Up to --streams batches are in flight:
    # each has its own session (HIP stream, workspaces) and host thread, so one batch's pair-HMM passes fill the
    # CUs that another batch's chain kernel leaves idle during its long tail.  Step s runs on session s % streams;
    # every session holds the same synthetic batch, so every step does the same work as with --streams 1.
    n_streams = max(1, min(args.streams, args.steps))
    free0 = torch.cuda.mem_get_info(local_rank)[0]
    sessions = [api.Session(params, batch, device=local_rank)]
    per_session = free0 - torch.cuda.mem_get_info(local_rank)[0]      # every workspace is allocated up front
    while len(sessions) < n_streams and torch.cuda.mem_get_info(local_rank)[0] > 1.25 * per_session:
        sessions.append(api.Session(params, batch, device=local_rank))
    n_streams = len(sessions)                                         # fewer in flight when HBM is the limit (cfg 3/4)
    sess = sessions[0]
    gathered = None
    ktime = {n: 0.0 for n in ffi.KERNEL_NAMES}
    klaunch = {n: 0 for n in ffi.KERNEL_NAMES}
    dev_ms = 0.0

    def run_steps(n_steps, timed):
        nonlocal gathered, dev_ms
        done = queue.Queue()

        def worker(i):
            try:
                for s in range(i, n_steps, n_streams):
                    sessions[i].run(skip_polish=False)   # synchronous: returns when the device has finished this pass
                    t = api.last_timing()                # thread-local: the pass this thread just ran
                    lab = sessions[i].fetch()["label"] if dist is not None else None
                    done.put((s, t, lab))
            except BaseException as e:  # noqa: BLE001 -- handed to the main thread
                done.put((-1, e, None))

        threads = [threading.Thread(target=worker, args=(i,)) for i in range(n_streams)]
        for th in threads:
            th.start()
        finished, nxt = {}, 0
        while nxt < n_steps:
            s, t, lab = done.get()
            if s < 0:
                raise t
            finished[s] = (t, lab)
            while nxt in finished:       # in step order on every rank
                t, lab = finished.pop(nxt)
                if dist is not None:     # the only exchange of the path: labels, RCCL all-gather
                    gathered = sharding.all_gather_labels(
                        dist, lab, device=torch.device("cuda", local_rank) if backend == "nccl" else None)
                if timed:
                    dev_ms += t["total_ms"]
                    for n in ffi.KERNEL_NAMES:
                        ktime[n] += t["kernel_ms"][n]
                        klaunch[n] += t["kernel_launches"][n]
                nxt += 1
        for th in threads:
            th.join()

    for _ in range(args.warmup):
        run_steps(n_streams, timed=False)      # one untimed pass on every session
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, timed=True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = sess.fetch()
    streams_agree = all(np.array_equal(s.fetch()["label"], out["label"]) for s in sessions[1:])
    ok = int((out["result"]["status"] == 0).sum())
    total_chunks = n_chunks * world * args.steps
    value = total_chunks / elapsed

    # ---- roofline of the dominant kernel family (HIP events on the library's own stream, summed over the
    #      timed steps): algorithmic bytes of the chunks one launch sequence processes / its device time
    dom = max(ktime, key=lambda n: ktime[n])
    alg_bytes_per_step = batch.algorithmic_bytes(k_per_chunk=out["result"]["cluster_num"])
    dom_ms_per_step = ktime[dom] / args.steps
    achieved = alg_bytes_per_step / 1e9 / (dom_ms_per_step / 1e3) if dom_ms_per_step > 0 else 0.0
    roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBPS, unit="GB/s", frac=achieved / HBM_PEAK_GBPS,
                    traffic=pmc_traffic(dom, args.workload, n_chunks), kernel=dom, kernel_ms_per_step=dom_ms_per_step,
                    launches_per_step=klaunch[dom] / args.steps,
                    algorithmic_bytes_per_step=alg_bytes_per_step,
                    all_kernels_ms_per_step={n: ktime[n] / args.steps for n in ffi.KERNEL_NAMES},
                    # measured HBM traffic (committed PMC passes) of every kernel family against its live device time:
                    # what each family actually moves, as opposed to the algorithmic bytes above
                    hbm_traffic_by_kernel={
                        n: dict(ms_per_step=ktime[n] / args.steps, launches_per_step=klaunch[n] / args.steps,
                                traffic_bytes_per_launch=pmc_traffic(n, args.workload, n_chunks),
                                traffic_GBps=(pmc_traffic(n, args.workload, n_chunks) * klaunch[n] / 1e9 / (ktime[n] / 1e3)
                                              if pmc_traffic(n, args.workload, n_chunks) and ktime[n] > 0 else None))
                        for n in ffi.KERNEL_NAMES},
                    note="byte/integer + f64 scan work: HBM-compulsory traffic is ~126 KB/chunk, so the HBM fraction "
                         "is tiny by construction; the binding limits are the serial Metropolis chain latency and "
                         "FP64 VALU in the banded pair-HMM (DESIGN.md).  kernel_ms_per_step is the mean duration of the "
                         "kernel's launches; with several batches in flight launches of different batches overlap, so it "
                         "can exceed ms_per_step")

    line = dict(metric="chunks clustered/sec (whole node), 60x ONT 2kbp chunks", value=value, unit="chunks/s",
                n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3,
                higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
                config=dict(workload=args.workload, chunks_per_gpu=n_chunks, reads_per_chunk=int(batch.chunks["n_reads"][0]),
                            chunk_len=int(cfg["tmpl_len"]), copy_num=int(cfg["copy_num"]), band_frac=cfg["band_frac"],
                            sharding=f"chunks/{world}gpu, labels all-gathered over RCCL" if world > 1 else "1 gpu",
                            batches_in_flight=n_streams),
                roofline=roofline, pass_latency_ms=dev_ms / args.steps, streams_agree=bool(streams_agree), chunks_ok=ok,
                mean_polish_rounds=float(out["result"]["polish_rounds"].mean()),
                mean_cluster_num=float(out["result"]["cluster_num"].mean()))

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = os.cpu_count() or 1
        sample = args.cpu_sample or max(2, min(n_chunks, threads // 2 if threads >= 16 else 4))
        cb, sub, ora = cpu_baseline(params, batch, sample, threads)
        line["cpu_baseline"] = cb
        # the checker doing its job on the sample: labels bit-exact, posteriors within 1e-4
        nr = int(sub.n_reads)
        line["parity_on_cpu_sample"] = dict(
            labels_equal=bool(np.array_equal(out["label"][:nr], ora["label"])),
            max_abs_dlogpost=float(np.abs(out["log_post"][:nr] - ora["log_post"]).max()))
    elif rank == 0:
        line["cpu_baseline"] = None
    for s in sessions:
        s.close()
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
