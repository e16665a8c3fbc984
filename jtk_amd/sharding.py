"""Chunk sharding across the GPUs of one node.

Chunks are independent units (each seeds its own RNG from its id, local_clustering/mod.rs:97; the reference
runs them as an unordered rayon map, mod.rs:64-72), so there is no data-path collective: every rank (one
process per GPU) clusters its own chunks and the labels are all-gathered at the end (RCCL on GPUs, gloo in
the CPU tests).
"""
import numpy as np


def weak_chunk_ids(rank, chunks_per_gpu):
    """Weak scaling (bench.py): rank r owns chunk ids r*c .. (r+1)*c-1."""
    return range(rank * chunks_per_gpu, (rank + 1) * chunks_per_gpu)


def chunk_cost(n_reads, tmpl_len, copy_num, polish_passes=3):
    """Relative cost model of one chunk: banded pair-HMM passes + Metropolis steps per candidate k
    (20 restarts x 2000 n proposals, pseudo_mcmc.rs:655,728)."""
    n_k = max(1, min(int(copy_num), 7) - 1)
    return float(n_reads) * float(tmpl_len) * polish_passes * 61 * 2 + 20.0 * 2000.0 * float(n_reads) * n_k * 40.0


def lpt_partition(costs, world):
    """Longest-processing-time-first assignment of chunks to `world` ranks. Returns a list of index arrays
    (each sorted ascending so a rank's chunks stay in input order)."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world)
    parts = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        parts[r].append(int(i))
        load[r] += costs[i]
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def all_gather_labels(dist, local_labels, device=None):
    """All-gather of per-read labels with ragged sizes: returns the list of every rank's label array.
    `dist` is torch.distributed (backend nccl == RCCL on ROCm, or gloo on CPU)."""
    import torch
    world = dist.get_world_size()
    lab = torch.as_tensor(np.ascontiguousarray(local_labels, dtype=np.int32))
    if device is not None:
        lab = lab.to(device)
    n = torch.tensor([lab.numel()], dtype=torch.int64, device=lab.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes) if sizes else 0
    pad = torch.zeros(mx, dtype=torch.int32, device=lab.device)
    pad[:lab.numel()] = lab
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return [o[:s].cpu().numpy().astype(np.uint32) for o, s in zip(out, sizes)]
