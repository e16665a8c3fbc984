"""The suite's slowest DEVICE calls, started early on other host threads.

Seven calls of the GPU suite are single chain workgroups that run for one to two minutes each (a 540-read pile-up at K = 4, 1,024 /
1,100 reads in the global-memory chain, the recursive splits): 255 compute units idle while the host waits.  The library serves
concurrent callers (own stream and workspaces per call; tests/test_gpu_parity.py::test_concurrent_sessions_match_one_shot), so those
calls are started on a small thread pool as soon as the tests that change the process environment have run (setenv must not overlap
another thread's getenv), and the test that needs a result finds it -- by CONTENT: the key is the sha256 of everything the call was
given, so a test whose inputs differ from what was started here simply makes its own call.  The result a test receives is the
return value of the same jtk_amd.api function it would have called itself.

OFF unless JTK_PREFETCH=1 (tests/conftest.py).  Measured with it on: the 24 K-way / chain cases of the suite (five of the seven calls
among them) in 240 s, every case green, the prefetched calls gone from the list of slowest tests.  The whole suite has not been timed
with it: two attempts of round 5 ended in bench.py's eight-rank test, which does not finish when the pytest process has touched the
GPU before it starts its eight child processes (prefetch threads running, or merely four GPU tests run first) -- with the switch on,
the bench tests therefore run first.  The default is the plain order the driver's run has always used."""
import threading
from concurrent.futures import ThreadPoolExecutor

import helpers

_futures = {}
_pool = None
_lock = threading.Lock()
N_THREADS = 4


def _key_chunks(p, b):
    return helpers._cache_key([b"device cluster_chunks", b.chunks, b.tmpl_bases, b.read_bases, b.read_off, b.ops, b.ops_off, b.strand,
                               bytes(p)])


def _key_features(p, chunks, var, vts, stride):
    return helpers._cache_key([b"device cluster_features", chunks, var, vts, bytes(p), bytes([stride])])


def install():
    """jtk_amd.api.cluster_chunks / cluster_features look a started call up before making their own."""
    from jtk_amd import api
    if getattr(api.cluster_chunks, "_prefetching", False):
        return
    orig_chunks, orig_features = api.cluster_chunks, api.cluster_features

    def cluster_chunks(params, batch, device=0, raise_on_chunk_failure=True, devices=None):
        if _futures and device == 0 and devices is None and raise_on_chunk_failure:
            fut = _futures.get(_key_chunks(params, batch))
            if fut is not None:
                return fut.result()
        return orig_chunks(params, batch, device=device, raise_on_chunk_failure=raise_on_chunk_failure, devices=devices)

    def cluster_features(params, feature_chunks, variants, variant_type, post_stride, device=0, **kw):
        if _futures and device == 0 and not kw:
            fut = _futures.get(_key_features(params, feature_chunks, variants, variant_type, post_stride))
            if fut is not None:
                return fut.result()
        return orig_features(params, feature_chunks, variants, variant_type, post_stride, device=device, **kw)

    cluster_chunks._prefetching = True
    cluster_chunks._orig = orig_chunks
    cluster_features._orig = orig_features
    api.cluster_chunks, api.cluster_features = cluster_chunks, cluster_features


def start(selected=None):
    """Builds the inputs of the slow cases (two seconds of host work) and submits the calls, longest first.  `selected`: the node
    ids of the tests this session runs -- a call nobody will ask for is not started."""
    global _pool
    from jtk_amd import api, batch as jb
    with _lock:
        if _pool is not None:
            return
        _pool = ThreadPoolExecutor(max_workers=N_THREADS, thread_name_prefix="jtk-prefetch")
    chunks_call, features_call = api.cluster_chunks._orig, api.cluster_features._orig

    def wanted(test_name):
        return selected is None or any(test_name in nodeid for nodeid in selected)

    p40 = jb.default_params(haploid_coverage=40.0)
    jobs = []
    for test_name, (specs, seed) in (("test_large_pileups_match_oracle", helpers.LARGE_PILEUP_SPECS),
                                     ("test_pileups_beyond_1023_reads_match_oracle", helpers.HUGE_PILEUP_SPECS)):
        if wanted(test_name):
            ch, var, vts, stride, _, _ = helpers.feature_inputs(specs, seed)
            jobs.append((_key_features(p40, ch, var, vts, stride), features_call, (p40, ch, var, vts, stride)))
    for test_name, build in (("test_pileup_beyond_511_reads_matches_oracle", helpers.pileup_540_inputs),
                             ("test_full_path_pileup_of_1100_reads_matches_oracle", helpers.full_path_1100_inputs)):
        if wanted(test_name):
            b, p = build()
            jobs.append((_key_chunks(p, b), chunks_call, (p, b)))
    if wanted("test_recursive_split_matches_oracle"):
        for case in sorted(helpers.RECURSIVE_SPLIT_CASES, key=lambda c: -c[1] * c[3]):
            b, p = helpers.recursive_split_inputs(*case)
            jobs.append((_key_chunks(p, b), chunks_call, (p, b)))
    for key, fn, args in jobs:
        _futures[key] = _pool.submit(fn, *args)


def stop():
    global _pool
    with _lock:
        pool, _pool = _pool, None
    if pool is not None:
        for f in _futures.values():
            f.cancel()
        pool.shutdown(wait=True)
    _futures.clear()
