"""tests/prefetch.py without a GPU: a started call is found by the content of its inputs, made once, and anything else goes to the
function itself."""
import threading

import numpy as np

import helpers
from jtk_amd import api, batch as jb


def test_started_calls_are_found_by_content_and_made_once(monkeypatch):
    import prefetch
    calls = []
    seen_threads = set()

    def fake_chunks(params, batch, device=0, raise_on_chunk_failure=True, devices=None):
        calls.append(("chunks", int(batch.chunks["n_reads"].sum())))
        seen_threads.add(threading.current_thread().name)
        return dict(rc=0, n=int(batch.chunks["n_reads"].sum()))

    def fake_features(params, feature_chunks, variants, variant_type, post_stride, device=0, **kw):
        calls.append(("features", int(feature_chunks["n_reads"].sum())))
        seen_threads.add(threading.current_thread().name)
        return dict(rc=0, n=int(feature_chunks["n_reads"].sum()))

    monkeypatch.setattr(api, "cluster_chunks", fake_chunks)
    monkeypatch.setattr(api, "cluster_features", fake_features)
    prefetch.install()
    try:
        prefetch.start()
        assert len(prefetch._futures) == 7
        # the test side: the same inputs, built the same way
        b, p = helpers.pileup_540_inputs()
        assert api.cluster_chunks(p, b)["n"] == int(b.chunks["n_reads"].sum())
        specs, seed = helpers.HUGE_PILEUP_SPECS
        ch, var, vts, stride, n, _ = helpers.feature_inputs(specs, seed)
        assert api.cluster_features(jb.default_params(haploid_coverage=40.0), ch, var, vts, stride)["n"] == n
        for f in list(prefetch._futures.values()):
            f.result()
        assert len(calls) == 7 and all(t.startswith("jtk-prefetch") for t in seen_threads)
        # other inputs, other parameters or other arguments: the function itself, on this thread
        b2, _, p2 = helpers.small_batch(n_chunks=1, tmpl_len=200, reads_per_hap=4)
        api.cluster_chunks(p2, b2)
        api.cluster_chunks(jb.default_params(haploid_coverage=61.0), b)
        api.cluster_chunks(p, b, raise_on_chunk_failure=False)
        assert len(calls) == 10 and threading.current_thread().name in seen_threads
    finally:
        prefetch.stop()
        monkeypatch.undo()
    assert not getattr(api.cluster_chunks, "_prefetching", False)
