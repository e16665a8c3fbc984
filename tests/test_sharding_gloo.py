"""The N > 1 path on CPU: two gloo ranks each cluster their own shard of the chunks (with the CPU oracle standing
in for the GPU) and all-gather the labels; the assembled result must equal a single-process run over all
chunks -- results do not depend on the sharding because every chunk seeds its own RNG from its id."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers
    import oracle_ffi as O
    from jtk_amd import sharding
    b, cfg, p = helpers.small_batch(n_chunks=6, tmpl_len=300, reads_per_hap=6)
    costs = [sharding.chunk_cost(int(c["n_reads"]), int(c["tmpl_len"]), int(c["copy_num"])) for c in b.chunks]
    parts = sharding.lpt_partition(costs, world)
    mine = b.subset(parts[rank].tolist())
    r = O.cluster_chunks(helpers.oracle_params(p), mine, n_threads=1)
    assert r["rc"] == 0
    gathered = sharding.all_gather_labels(dist, r["label"])
    if rank == 0:
        full = O.cluster_chunks(helpers.oracle_params(p), b, n_threads=2)
        # reassemble in original chunk order
        out = np.zeros(b.n_reads, dtype=np.uint32)
        for rk in range(world):
            off = 0
            for c in parts[rk]:
                n = int(b.chunks[c]["n_reads"])
                first = int(b.chunks[c]["read_first"])
                out[first:first + n] = gathered[rk][off:off + n]
                off += n
        np.save(os.path.join(outdir, "ok.npy"), np.array([int(np.array_equal(out, full["label"]))]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharded_run_matches_single_process(tmp_path, jtk_lib, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert np.load(tmp_path / "ok.npy")[0] == 1


def test_lpt_partition_balances_and_covers():
    from jtk_amd import sharding
    rng = np.random.default_rng(0)
    costs = rng.uniform(1, 10, 101)
    for world in (1, 2, 4, 8):
        parts = sharding.lpt_partition(costs, world)
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(101))
        loads = [costs[p].sum() for p in parts]
        assert max(loads) - min(loads) <= costs.max() + 1e-9
    assert list(sharding.weak_chunk_ids(3, 500))[:2] == [1500, 1501]


def _gather_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jtk_amd import sharding
    # ragged shards: sizes follow from the partition every rank computes for itself, nothing is exchanged beforehand
    parts = sharding.strong_shards(7, [4, 9, 3, 6, 5, 8, 2], 100, 2, world)
    stride = 3
    sizes = [(int(sum([4, 9, 3, 6, 5, 8, 2][c] for c in p)), len(p)) for p in parts]
    g = sharding.ResultGather(dist, sizes, stride)

    def payload(r, step):
        n, c = sizes[r]
        rng = np.random.default_rng(100 * step + r)
        return (rng.integers(0, 3, n).astype(np.uint32), rng.normal(size=(n, stride)), rng.integers(1, 4, c).astype(np.uint32),
                rng.normal(size=c))
    ok = True
    for step in range(3):                      # the buffers are reused from step to step
        got = g.gather(*payload(rank, step))
        for r in range(world):
            lab, post, k, sc = payload(r, step)
            ok = ok and np.array_equal(got[r]["label"], lab) and np.array_equal(got[r]["log_post"], post)
            ok = ok and np.array_equal(got[r]["cluster_num"], k) and np.array_equal(got[r]["score"], sc)
    np.save(os.path.join(outdir, f"gather{rank}.npy"), np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


def test_result_gather_is_one_collective_of_labels_posteriors_k_and_score(tmp_path):
    """(label, log_post, cluster_num, score) of every rank to every rank in ONE all_gather_into_tensor on
    sizes known from the partition"""
    world = 2
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(np.load(tmp_path / f"gather{r}.npy")[0] == 1 for r in range(world))


def _gather8_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jtk_amd import sharding
    # the headline data set's partition (2,500 chunks x 60 reads x 2 kbp, copy number 2) as bench.py --gpus 8 deals it
    n_chunks, n_reads, stride = 2500, 60, 2
    parts = sharding.strong_shards(n_chunks, n_reads, 2000, 2, world)
    sizes = [(n_reads * len(p), len(p)) for p in parts]
    g = sharding.ResultGather(dist, sizes, stride)

    def payload(r):
        n, c = sizes[r]
        rng = np.random.default_rng(r)
        return (rng.integers(0, 2, n).astype(np.uint32), rng.normal(size=(n, stride)), rng.integers(1, 3, c).astype(np.uint32),
                rng.normal(size=c))
    ok = sorted(int(c) for p in parts for c in p) == list(range(n_chunks))
    ok = ok and max(len(p) for p in parts) - min(len(p) for p in parts) <= 1      # uniform chunks: LPT deals them evenly
    for step in range(2):
        got = g.gather(*payload(rank))
        for r in range(world):
            lab, post, k, sc = payload(r)
            ok = ok and np.array_equal(got[r]["label"], lab) and np.array_equal(got[r]["log_post"], post)
            ok = ok and np.array_equal(got[r]["cluster_num"], k) and np.array_equal(got[r]["score"], sc)
    np.save(os.path.join(outdir, f"gather8_{rank}.npy"), np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


def test_result_gather_at_eight_ranks_on_the_headline_partition(tmp_path):
    """the exchange step of `bench.py --gpus 8` at its real sizes (312 / 313 chunks x 60 reads per rank, ~3 MB per step) on
    eight gloo ranks: every rank receives every rank's labels, posteriors, cluster numbers and scores; no GPU involved"""
    world = 8
    mp.spawn(_gather8_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(np.load(tmp_path / f"gather8_{r}.npy")[0] == 1 for r in range(world))


def test_predicted_cost_spread_of_the_strong_partition():
    """the partition every rank computes from the metadata alone: on the headline data set the 8 shards differ by one chunk,
    and on a ragged one (Poisson coverage, 2- and 4-copy chunks, three template lengths) the predicted per-rank cost -- pair-HMM
    cells + Metropolis steps per candidate k -- stays within 2 % of the mean.  (What the model cannot know before the filter
    has run is how EVENTFUL a chunk's chain will be: a rank's chain launch lasts as long as its slowest chunk)"""
    from jtk_amd import sharding
    parts = sharding.strong_shards(2500, 60, 2000, 2, 8)
    assert sorted(len(p) for p in parts) == [312] * 4 + [313] * 4
    assert sorted(np.concatenate(parts).tolist()) == list(range(2500))
    rng = np.random.default_rng(3)
    n = 1500
    reads = rng.poisson(60, n).clip(8, 200)
    copy = rng.choice([2, 2, 2, 4], n)
    tlen = rng.choice([1000, 2000, 4000], n)
    parts = sharding.strong_shards(n, reads * (copy // 2), tlen, copy, 8)
    cost = np.array([sharding.chunk_cost(r * (c // 2), L, c) for r, c, L in zip(reads, copy, tlen)])
    per_rank = np.array([cost[p].sum() for p in parts])
    assert per_rank.max() / per_rank.mean() < 1.02 and per_rank.min() / per_rank.mean() > 0.98
