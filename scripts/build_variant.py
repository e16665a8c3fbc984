#!/usr/bin/env python3
"""A diagnostic build of the library that differs from the product in ONE source file: that file is compiled with the extra
flags, every other object is the product build's (jtk_amd/_build/*.o), the result is jtk_amd/_build/exp_<name>/libjtk_lc_<name>.so
(point JTK_LC_LIB at it).  Seconds instead of the minutes of build.build_experiment, which recompiles everything.

    python3 scripts/build_variant.py wlane phmm_sweep.hip -DJTK_PHMM_X_WLANE
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jtk_amd import build  # noqa: E402


def main():
    name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    build.build()
    out_dir = os.path.join(build.OUT_DIR, "exp_" + name)
    os.makedirs(out_dir, exist_ok=True)
    obj = os.path.join(out_dir, os.path.splitext(src)[0] + ".o")
    subprocess.check_call([build.HIPCC] + build.FLAGS + extra + ["-x", "hip", "-c", os.path.join(build.CSRC, src), "-o", obj])
    objs = [obj if s == src else os.path.join(build.OUT_DIR, os.path.splitext(s)[0] + ".o") for s in build.SOURCES]
    lib = os.path.join(out_dir, "libjtk_lc_%s.so" % name)
    subprocess.check_call([build.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + build.EXPORTS_MAP,
                           "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
