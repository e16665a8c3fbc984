"""Builds jtk_amd/_build/libjtk_lc.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OUT_DIR = os.path.join(PKG, "_build")
OUT = os.path.join(OUT_DIR, "libjtk_lc.so")
# the synthetic pile-up generator (bench.py and tests only) is NOT part of the product library
SYNTH_OUT = os.path.join(OUT_DIR, "libjtk_synth.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
EXPORTS_MAP = os.path.join(CSRC, "exports.map")  # only jtk_lc_* leaves the library

SOURCES = ["phmm_kernels.hip", "phmm_sweep.hip", "phmm_pair.hip", "phmm_wide.hip", "polish_kernels.hip", "filter_kernels.hip", "mcmc_kernels.hip", "session.hip", "gains.hip", "correction.hip",
           "io_kernels.hip", "host_api.cpp"]
SYNTH_SOURCES = ["synth.cpp"]
# -ffp-contract=off: device f64 arithmetic must round exactly like the reference (no implicit fma);
# the pair-HMM specification uses explicit fma() where it wants one.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
         "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + os.environ.get("JTK_EXTRA_HIPCC_FLAGS", "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    objs = []
    synth_objs = []
    procs = []
    for src in SOURCES + SYNTH_SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OUT_DIR, os.path.splitext(src)[0] + ".o")
        (synth_objs if src in SYNTH_SOURCES else objs).append(obj)
        if force or _stale(obj, [path] + headers):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or (verbose and out):
            sys.stderr.write(out.decode())
        if p.returncode != 0:
            failed = True
    if failed:
        raise RuntimeError("hipcc failed")
    if force or _stale(OUT, objs + [EXPORTS_MAP]):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + EXPORTS_MAP, "-o", OUT] + objs
        subprocess.check_call(cmd)
    if force or _stale(SYNTH_OUT, synth_objs):
        subprocess.check_call([HIPCC, "-shared", "-fPIC", "-o", SYNTH_OUT] + synth_objs)
    return OUT


def build_experiment(name, extra_flags):
    """A diagnostic build (statistics counters, kernel variants) of the library beside the product: objects and
    libjtk_lc_<name>.so under _build/exp_<name>/, every source compiled with `extra_flags` added.  The product library is not
    touched; scripts point JTK_LC_LIB (jtk_amd/ffi.py) at the returned path."""
    out_dir = os.path.join(OUT_DIR, "exp_" + name)
    os.makedirs(out_dir, exist_ok=True)
    extra = extra_flags.split() if isinstance(extra_flags, str) else list(extra_flags)
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(out_dir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + extra + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on " + src)
    lib = os.path.join(out_dir, "libjtk_lc_%s.so" % name)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + EXPORTS_MAP, "-o", lib] + objs)
    return lib


PROFILED_SOURCES = ["phmm_kernels.hip", "phmm_sweep.hip", "phmm_pair.hip", "phmm_wide.hip", "polish_kernels.hip", "filter_kernels.hip", "mcmc_kernels.hip",
                    "session.hip", "io_kernels.hip", "device_common.h", "finalize_common.h"]


def source_sha16():
    """sha256[:16] over the sources that determine the profiled kernels and their launches (names + contents + flags): what
    a profile under profiles/ is tied to -- unlike the .so's own hash it does not depend on where or when hipcc ran, nor
    on entry points added beside the path."""
    import hashlib
    files = [os.path.join(CSRC, f) for f in PROFILED_SOURCES] + [os.path.join(ROOT, "include", "jtk_math.h")]
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS[:6] + FLAGS[7:8]).encode())  # (visibility does not change a kernel)
    return h.hexdigest()[:16]


def is_stale():
    """True when a source is newer than the library built from it (bench.py refuses to time a stale build)."""
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    return (_stale(OUT, [os.path.join(CSRC, f) for f in SOURCES] + headers)
            or _stale(SYNTH_OUT, [os.path.join(CSRC, f) for f in SYNTH_SOURCES] + headers))


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
