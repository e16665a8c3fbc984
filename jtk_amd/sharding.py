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


def strong_shards(n_chunks, n_reads, tmpl_len, copy_num, world):
    """Strong scaling: ONE fixed dataset of `n_chunks` chunks (ids 0..n_chunks-1), dealt to `world` ranks by
    longest-processing-time-first over `chunk_cost`.  n_reads / tmpl_len / copy_num: scalars or per-chunk
    sequences.  Every rank computes the same partition from the metadata alone, so no sizes are ever exchanged."""
    bc = lambda v: np.broadcast_to(np.asarray(v), (n_chunks,))  # noqa: E731
    costs = [chunk_cost(n, L, c) for n, L, c in zip(bc(n_reads), bc(tmpl_len), bc(copy_num))]
    return lpt_partition(costs, world)


def payload_nbytes(n_reads, n_chunks, stride):
    """label u32[n_reads] | cluster_num u32[n_chunks] | (8-byte aligned) log_post f64[n_reads*stride] | score f64[n_chunks]"""
    ints = 4 * (n_reads + n_chunks)
    return ((ints + 7) // 8) * 8 + 8 * (n_reads * stride + n_chunks)


def pack_results(label, log_post, cluster_num, score, out=None):
    n_reads, n_chunks = len(label), len(cluster_num)
    stride = log_post.shape[1] if n_reads else 1
    nb = payload_nbytes(n_reads, n_chunks, stride)
    buf = np.zeros(nb, dtype=np.uint8) if out is None else out
    ints = buf[:4 * (n_reads + n_chunks)].view(np.uint32)
    ints[:n_reads] = label
    ints[n_reads:] = cluster_num
    f = buf[nb - 8 * (n_reads * stride + n_chunks):nb].view(np.float64)
    f[:n_reads * stride] = np.ascontiguousarray(log_post, dtype=np.float64).ravel()
    f[n_reads * stride:] = score
    return buf


def unpack_results(buf, n_reads, n_chunks, stride):
    nb = payload_nbytes(n_reads, n_chunks, stride)
    ints = np.ascontiguousarray(buf[:4 * (n_reads + n_chunks)]).view(np.uint32)
    f = np.ascontiguousarray(buf[nb - 8 * (n_reads * stride + n_chunks):nb]).view(np.float64)
    return dict(label=ints[:n_reads].copy(), cluster_num=ints[n_reads:].copy(),
                log_post=f[:n_reads * stride].reshape(n_reads, stride).copy(), score=f[n_reads * stride:].copy())


class ResultGather:
    """The path's only exchange step: every rank's (label, log_post, cluster_num, score) to every rank, as
    ONE all_gather_into_tensor of equal-sized byte payloads.  The per-rank sizes follow from the partition, which every
    rank knows, so nothing is exchanged or synchronised with the host beforehand (no .item()); buffers are allocated
    once.  `sizes`: [(n_reads, n_chunks)] per rank.  backend nccl (== RCCL over xGMI on ROCm) with `device` set, gloo
    on CPU tensors otherwise."""

    def __init__(self, dist, sizes, stride, device=None):
        import torch
        self.dist, self.sizes, self.stride, self.device = dist, list(sizes), int(stride), device
        self.world = dist.get_world_size()
        assert len(self.sizes) == self.world
        self.slot = max(payload_nbytes(n, c, self.stride) for n, c in self.sizes)
        self.slot = ((self.slot + 15) // 16) * 16
        self.host_in = torch.zeros(self.slot, dtype=torch.uint8)
        self.host_out = torch.zeros(self.slot * self.world, dtype=torch.uint8)
        if device is not None:
            self.host_in, self.host_out = self.host_in.pin_memory(), self.host_out.pin_memory()
            self.dev_in = torch.zeros(self.slot, dtype=torch.uint8, device=device)
            self.dev_out = torch.zeros(self.slot * self.world, dtype=torch.uint8, device=device)

    def gather(self, label, log_post, cluster_num, score):
        rank = self.dist.get_rank()
        n, c = self.sizes[rank]
        assert len(label) == n and len(cluster_num) == c
        pack_results(label, log_post, cluster_num, score, out=self.host_in.numpy()[:payload_nbytes(n, c, self.stride)])
        if self.device is not None:
            self.dev_in.copy_(self.host_in, non_blocking=True)
            self.dist.all_gather_into_tensor(self.dev_out, self.dev_in)
            self.host_out.copy_(self.dev_out)      # synchronises: the payloads are on the host when this returns
        else:
            self.dist.all_gather_into_tensor(self.host_out, self.host_in)
        out = self.host_out.numpy()
        return [unpack_results(out[r * self.slot:(r + 1) * self.slot], n_r, c_r, self.stride)
                for r, (n_r, c_r) in enumerate(self.sizes)]


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
