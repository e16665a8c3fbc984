"""Generates tests/golden/default_gains.json: `estimate_gain_default` (likelihood_gains.rs:186-192) for
HMMParam::default() on both strands, computed by the CPU oracle (oracle/likelihood_gains.c with the
own-spec simulator of oracle/phmm.c).  The numbers are inputs at the C-ABI boundary (a jtk host computes
them with kiley); jtk_amd/batch.py carries a copy as DEFAULT_GAINS."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle_ffi as O  # noqa: E402

L = O.lib()
h = O.default_hmm()
g = O.Gains()
L.jo_estimate_gain_default(C.byref(h), C.byref(h), C.byref(g))
out = {name: [[getattr(g, name)[i].gain, getattr(g, name)[i].prob] for i in range(g.max_homopolymer_len)]
       for name in ("subst", "deletions", "insertions")}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "default_gains.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
