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
