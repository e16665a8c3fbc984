"""Diagnostic (GPU box): what ONE stage call from host buffers costs a fresh process -- jtk_lc_cluster_chunks on the headline data
set (2,500 x 60 x 2 kbp), first call (device workspaces are mapped) and the two calls after it (pooled blocks), with the device
memory the library holds after each.  bench.py runs it as a child process (one call) BEFORE it touches the GPU itself:
`e2e.fresh_process` of the bench line.  usage: cold_call.py [n_chunks] [config] [n_calls]"""
import json
import os
import sys
import time

os.environ.setdefault("JTK_LC_POOL_GB", "160")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jtk_amd import api, batch as jb, synth  # noqa: E402

n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
config = sys.argv[2] if len(sys.argv) > 2 else "ont_diploid"
n_calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
b, cfg = synth.make_batch(config, n_chunks, first_chunk_id=0)
p = jb.default_params(haploid_coverage=cfg["coverage"], band_frac=cfg["band_frac"])
torch.cuda.init()
free0, total = torch.cuda.mem_get_info(0)
rows = []
ref = None
for call in range(n_calls):
    t = time.perf_counter()
    out = api.cluster_chunks(p, b, device=0)
    dt = time.perf_counter() - t
    free1, _ = torch.cuda.mem_get_info(0)
    tm = api.last_timing()
    rows.append(dict(call=call, seconds=round(dt, 3), held_gb=round((free0 - free1) / 1e9, 1), h2d_ms=round(tm["h2d_ms"], 1),
                     d2h_ms=round(tm["d2h_ms"], 1), kernel_ms={k: round(v, 1) for k, v in tm["kernel_ms"].items()}))
    if ref is None:
        ref = out
    else:
        assert np.array_equal(ref["label"], out["label"])
print(json.dumps(dict(workload=f"{config} x {n_chunks}", total_gb=round(total / 1e9, 1), calls=rows)))
