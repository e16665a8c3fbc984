import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import helpers, oracle_ffi as O, ctypes as C
from jtk_amd import api, batch as jb, ffi
import test_gpu_parity as T
p = jb.default_params(haploid_coverage=30.0)
for specs in ([(60,6,2,2)]*8, [(60,3,2,2)]*8, [(160,12,4,4)]*4):
    t0=time.time(); dev, ora, truth = T.run_features_both(p, specs, seed=3); dt=time.time()-t0
    t=api.last_timing()
    ok = np.array_equal(dev["label"], ora["label"]) and np.array_equal(helpers.bits(dev["log_post"]), helpers.bits(ora["log_post"]))
    n,d,kt,cn=specs[0]
    ks=[2] if cn==2 else list(range(2,cn+1))
    steps=20*2000*n*len(ks)
    print(specs[0], "mcmc ms", t["kernel_ms"]["mcmc"], "ns/step(upper)", t["kernel_ms"]["mcmc"]*1e6/steps, "parity", ok, "k", dev["result"]["cluster_num"].tolist())
