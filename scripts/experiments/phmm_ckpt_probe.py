"""(Applies to the sources of commit f4dcf3d: round 4 then built the real kernel, jtk_amd/csrc/phmm_sweep.hip -- 54.0 -> 45.0 ms,
the bound this probe gave for C = 8.)  Experiment (not part of the product), TIMING ONLY -- the results of these builds are garbage: what a checkpointed pair stream
could buy phmm_kernel at best.  VERDICT round 3 (Next 3) asked for a timed build instead of an estimate.

A checkpoint every C-th anti-diagonal is the lane's four forward values (toM_1, toM_2, toI_1, toD_1: 32 bytes) instead of one 16-byte
pair per diagonal, so the stream shrinks to 2 / C of itself; the forward arithmetic of every segment is redone in front of the
backward sweep.  The emulation patches a COPY of the sources and keeps the kernel's structure:
  ckptC        stores 32 bytes at every C-th diagonal only; the backward sweep loads 256 / C bytes per group of 8 diagonals;
  ckptC_fwd2   the same, and the forward sweep runs a second time without stores (the recomputation: all of it, but no LDS segment
               to hold it -- occupancy unchanged --, no re-scaling of recomputed values, perfect overlap): an OPTIMISTIC bound;
  fwd2         the second forward sweep alone (what the recomputation costs);
  nostream     no stripe stores and no loads at all (the floor).
Builds _build/exp_ckpt_<variant>/libjtk_lc_<variant>.so; prints the paths.  usage: python scripts/experiments/phmm_ckpt_probe.py"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from jtk_amd import build as jbuild  # noqa: E402


def patch(s, ckpt, fwd2):
    def rep(old, new, n=1):
        nonlocal s
        assert s.count(old) == n, (old, s.count(old))
        s = s.replace(old, new)
    if ckpt:
        rep('''                    out[u * 64 + lane] = make_double2(toM_2, toD_1);  // toM of diagonal t-1 goes out in ITS block's scale ...''',
            '''                    if (X_FWD_STORES) {
                        out[u * 64 + lane] = make_double2(toM_2, toD_1);
                        out[(u + 1) * 64 + lane] = make_double2(toM_1, toI_1);
                    }''')
        rep('''                    out[u * 64 + lane] = make_double2(toM_2, toD_1);
#endif
                }
            }
            t += 8;''',
            '''                    if (X_FWD_STORES && (u %% %d) == 0) {
                        out[u * 64 + lane] = make_double2(toM_2, toD_1);
                        out[(u + 1) * 64 + lane] = make_double2(toM_1, toI_1);
                    }
#endif
                }
            }
            t += 8;''' % ckpt)
        # 8 pair loads = 128 bytes per lane and group of 8 diagonals; checkpoints: 256 / C bytes
        rep("for (int k = 0; k < 4; k++) qB[k] = pin[-64 * (4 + k)];",
            "for (int k = 0; k < 4; k++) qB[k] = k < 2 ? pin[-64 * (4 + k)] : make_double2(1e-3 * (tb + k), 0.5);")
        rep("for (int k = 0; k < 4; k++) qA[k] = pin[-64 * (8 + k)];",
            "for (int k = 0; k < 4; k++) qA[k] = k < %d ? pin[-64 * (8 + k)] : make_double2(1e-3 * (tb - k), 0.25);" % (2 if ckpt == 4 else 0))
    rep('''    double endM = 0, endI = 0, endD = 0;
    {  // t == 0: the only cell is (0, 0), on lane 0''',
        ('''    double endM = 0, endI = 0, endD = 0;
    int row = 0;
    uint32_t xrow = 0;
    for (int x_rep = 0; x_rep < 2; x_rep++) {
    c = 0, EF = 0, toM_1 = 0, toM_2 = 0, toI_1 = 0, toD_1 = 0, endM = 0, endI = 0, endD = 0, row = 0, xrow = 0;
#define X_FWD_STORES (x_rep == 0)
''' if fwd2 else '''    double endM = 0, endI = 0, endD = 0;
#define X_FWD_STORES true
''') + '''    {  // t == 0: the only cell is (0, 0), on lane 0''')
    if fwd2:
        rep('''    int row = 0;        // fast groups: the lane's template row (a spare lane: the row it takes next)
    uint32_t xrow = 0;  // fast groups: LDS address of the eM row of x[row-1]
''', '')
        rep('''    scratch[(int64_t)(T + 1) * 64 + lane] = make_double2(toM_1, 0.0);  // P_{T+1} = (toM of diagonal T, nothing)''',
            '''    }
    scratch[(int64_t)(T + 1) * 64 + lane] = make_double2(toM_1, 0.0);  // P_{T+1} = (toM of diagonal T, nothing)''')
        if not ckpt:  # stores of the second sweep off
            rep("out[u * 64 + lane] = make_double2(toM_2, toD_1);", "if (X_FWD_STORES) out[u * 64 + lane] = make_double2(toM_2, toD_1);", 2)
    return s


def build(name, ckpt, fwd2, extra=()):
    src_dir = os.path.join(jbuild.OUT_DIR, "exp_ckpt_" + name, "csrc")
    shutil.rmtree(src_dir, ignore_errors=True)
    shutil.copytree(jbuild.CSRC, src_dir)
    p = os.path.join(src_dir, "phmm_sweep.hip")
    if ckpt or fwd2:
        text = patch(open(p).read(), ckpt, fwd2)
        open(p, "w").write(text)
    out_dir = os.path.dirname(src_dir)
    objs, procs = [], []
    for f in jbuild.SOURCES:
        obj = os.path.join(out_dir, os.path.splitext(f)[0] + ".o")
        objs.append(obj)
        flags = [x if x != "-I" + jbuild.CSRC else "-I" + src_dir for x in jbuild.FLAGS] + list(extra)
        cmd = [jbuild.HIPCC] + flags + (["-x", "hip"] if f.endswith(".hip") else []) + ["-c", os.path.join(src_dir, f), "-o", obj]
        procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for f, pr in procs:
        o, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(o.decode()[-3000:])
            raise SystemExit("hipcc failed on " + f + " (" + name + ")")
    lib = os.path.join(out_dir, "libjtk_lc_%s.so" % name)
    subprocess.check_call([jbuild.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    build("ckpt4", 4, False)
    build("ckpt4_fwd2", 4, True)
    build("ckpt8", 8, False)
    build("ckpt8_fwd2", 8, True)
    build("fwd2", 0, True)
    build("nostream", 0, False, extra=("-DJTK_PHMM_X_NOSTORE", "-DJTK_PHMM_X_NOLOAD"))
