"""Experiment (not part of the product): non-temporal hints on phmm_kernel's stripe stream in the unrolled middle of both
sweeps -- the pairs are written once by the forward sweep and read once by the backward sweep (254 GB per pass of 30,000
reads through a 4 MB L2 per XCD).  Patches a COPY of the sources, builds _build/exp_nt/libjtk_lc_nt.so from it and prints its
path; the product sources and library are not touched.  Run the result with JTK_LC_LIB=<path> (scripts/phmm_single_pass.py for
the timing, any parity test for the bits).  usage: python scripts/experiments/phmm_nt_stream.py [stores|loads|both]"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jtk_amd import build as jbuild  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "both"
src_dir = os.path.join(jbuild.OUT_DIR, "exp_nt_" + what, "csrc")
shutil.rmtree(src_dir, ignore_errors=True)
shutil.copytree(jbuild.CSRC, src_dir)
p = os.path.join(src_dir, "phmm_sweep.hip")
s = open(p).read()
helpers = '''
typedef double nt_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_store(double2 *p, double2 v) {
    nt_d2 w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<nt_d2 *>(p));
}
__device__ __forceinline__ double2 nt_load(const double2 *p) {
    const nt_d2 w = __builtin_nontemporal_load(reinterpret_cast<const nt_d2 *>(p));
    return make_double2(w.x, w.y);
}
'''
anchor = "__device__ __forceinline__ double rot_from_prev(double v) {"
assert s.count(anchor) == 1
s = s.replace(anchor, helpers + anchor)
if what in ("stores", "both"):
    for old, new in (("out[u * 64 + lane] = make_double2(toM_2, toD_1);", "nt_store(&out[u * 64 + lane], make_double2(toM_2, toD_1));"),):
        assert s.count(old) == 2, s.count(old)
        s = s.replace(old, new)
if what in ("loads", "both"):
    for old, new in (("qB[k] = pin[-64 * (4 + k)];", "qB[k] = nt_load(&pin[-64 * (4 + k)]);"),
                     ("qA[k] = pin[-64 * (8 + k)];", "qA[k] = nt_load(&pin[-64 * (8 + k)]);")):
        assert s.count(old) == 1, (old, s.count(old))
        s = s.replace(old, new)
open(p, "w").write(s)
out_dir = os.path.dirname(src_dir)
objs, procs = [], []
for f in jbuild.SOURCES:
    obj = os.path.join(out_dir, os.path.splitext(f)[0] + ".o")
    objs.append(obj)
    flags = [x if x != "-I" + jbuild.CSRC else "-I" + src_dir for x in jbuild.FLAGS]
    cmd = [jbuild.HIPCC] + flags + (["-x", "hip"] if f.endswith(".hip") else []) + ["-c", os.path.join(src_dir, f), "-o", obj]
    procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
for f, pr in procs:
    o, _ = pr.communicate()
    if pr.returncode != 0:
        sys.stderr.write(o.decode())
        raise SystemExit("hipcc failed on " + f)
lib = os.path.join(out_dir, "libjtk_lc_nt_%s.so" % what)
subprocess.check_call([jbuild.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
