"""Diagnostic (GPU box): wall time of the gains calibration (estimate_gain_default) on the device path and in the
CPU oracle.  `python scripts/gains_timing.py`"""
import ctypes as C
import sys
import time
import torch  # noqa: F401
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
import helpers  # noqa: E402
import oracle_ffi as O  # noqa: E402
from jtk_amd import api, batch as jb  # noqa: E402

p = jb.default_params(haploid_coverage=30.0)
api.estimate_gains(p.forward, p.reverse, homop_len=1)   # warm-up: context, code objects
t0 = time.time()
dev = api.estimate_gains(p.forward, p.reverse)
t1 = time.time()
po = helpers.oracle_params(p)
ora = O.Gains()
O.lib().jo_estimate_gain_default(C.byref(po.forward), C.byref(po.reverse), C.byref(ora))
t2 = time.time()
same = all(getattr(dev, n)[h].gain == getattr(ora, n)[h].gain and getattr(dev, n)[h].prob == getattr(ora, n)[h].prob
           for n in ("subst", "deletions", "insertions") for h in range(3))
print("estimate_gain_default: device path %.3f s, CPU oracle (1 thread) %.3f s, identical %s" % (t1 - t0, t2 - t1, same))
