"""Builds the library of ANOTHER git revision beside the product, for A/B runs on the GPU box:
    python3 scripts/build_rev.py <rev> <name> [extra hipcc flags]
checks the revision's jtk_amd/csrc and include/ out into jtk_amd/_build/rev_<name>/src and compiles them with that revision's
flags to jtk_amd/_build/rev_<name>/libjtk_lc_<name>.so (built .so files travel with gpurun; point JTK_LC_LIB at it)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jtk_amd import build as jbuild  # noqa: E402


def main():
    rev, name, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    out = os.path.join(jbuild.OUT_DIR, "rev_" + name)
    src = os.path.join(out, "src")
    os.makedirs(src, exist_ok=True)
    tar = subprocess.run(["git", "-C", ROOT, "archive", rev, "jtk_amd/csrc", "include"], stdout=subprocess.PIPE, check=True).stdout
    subprocess.run(["tar", "-x", "-C", src], input=tar, check=True)
    csrc, inc = os.path.join(src, "jtk_amd", "csrc"), os.path.join(src, "include")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-I" + inc, "-I" + csrc] + extra
    objs, procs = [], []
    for f in jbuild.SOURCES:
        obj = os.path.join(out, os.path.splitext(f)[0] + ".o")
        objs.append(obj)
        cmd = [jbuild.HIPCC] + flags + (["-x", "hip"] if f.endswith(".hip") else []) + ["-c", os.path.join(csrc, f), "-o", obj]
        procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for f, pr in procs:
        o, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(o.decode()[-3000:])
            raise SystemExit("hipcc failed on " + f)
    lib = os.path.join(out, "libjtk_lc_%s.so" % name)
    subprocess.check_call([jbuild.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
