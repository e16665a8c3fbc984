"""Diagnostic (GPU box): do N resident sessions, each on its own stream and host thread, overlap usefully?  The chain
kernel leaves most of the machine idle during its tail; the next batch's pair-HMM passes can run there.
`python scripts/overlap_probe.py [passes_per_session] [n_sessions ...]`"""
import sys
import threading
import time
import torch  # noqa: F401
sys.path.insert(0, "/root/repo")
from jtk_amd import api, batch as jb  # noqa: E402
from bench import make_batch_parallel  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
counts = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 4]
sessions = []
for i in range(max(counts)):
    b, cfg = make_batch_parallel("ont_diploid", 500, 500 * i)
    p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
    s = api.Session(p, b)
    s.run()   # warm-up
    sessions.append(s)


def worker(s):
    for _ in range(steps):
        s.run()


for n in counts:
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(sessions[i],)) for i in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print("%d session(s) in flight: %d passes of 500 chunks in %.3f s -> %.1f chunks/s (%.0f ms per pass)"
          % (n, n * steps, dt, 500 * n * steps / dt, dt / (n * steps) * 1e3))
for s in sessions:
    r = s.fetch()
    assert int((r["result"]["status"] == 0).sum()) == 500
