"""Diagnostic build (not part of the product; the product sources are not touched): mcmc_chain_tab with cycle counters around the
parts of an EVENT -- set-up of the tentative state, get_lk, the Bernoulli decision, accept / flip-back bookkeeping, the republish,
the hop words afterwards -- on a patched COPY of the sources, -DJTK_MCMC_STATS.  Prints the library path.
Run with scripts/experiments/tab_event/run_probe.sh on the GPU box."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from jtk_amd import build as jbuild  # noqa: E402


def patch(s):
    def rep(old, new, n=1):
        nonlocal s
        assert s.count(old) == n, (old[:70], s.count(old))
        s = s.replace(old, new)
    rep("    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // fast, events, accepts, reloads, scalars, cyc rebuild, cyc event, cyc total",
        "    unsigned long long ts[18] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};")
    rep("        const double proposed = get_lk(T, P, W, ncl);\n        const double diff = unif64(proposed - lk);",
        "        const unsigned long long x_t1 = __builtin_readcyclecounter();\n        TS_ADD(8, x_t1 - ev_t0);\n"
        "        const double proposed = get_lk(T, P, W, ncl);\n        const double diff = unif64(proposed - lk);\n"
        "        const unsigned long long x_t2 = __builtin_readcyclecounter();\n        TS_ADD(9, x_t2 - x_t1);")
    rep("                accept = ubool(bernoulli_exact(uni64(lds_ld64(&rng.ring[ring_slot(pos_v)])), diff));\n            }\n        }\n        if (accept) {\n#pragma unroll\n            for (int c = 0; c < K; c++) {\n                tg[c] = T[c];",
        "                accept = ubool(bernoulli_exact(uni64(lds_ld64(&rng.ring[ring_slot(pos_v)])), diff));\n                TS_ADD(13, 1);\n            }\n        }\n"
        "        const unsigned long long x_t3 = __builtin_readcyclecounter();\n        TS_ADD(10, x_t3 - x_t2);\n"
        "        if (accept) {\n#pragma unroll\n            for (int c = 0; c < K; c++) {\n                tg[c] = T[c];")
    rep("            if (lane == 0) m.assign[idx] = (uint8_t)nw;\n            wsync();\n            lk = proposed;\n            if (ubool(max < lk)) {\n                max = proposed;",
        "            const unsigned long long x_a1 = __builtin_readcyclecounter();\n            TS_ADD(14, x_a1 - x_t3);\n"
        "            if (lane == 0) m.assign[idx] = (uint8_t)nw;\n            wsync();\n            lk = proposed;\n"
        "            const unsigned long long x_a2 = __builtin_readcyclecounter();\n            TS_ADD(15, x_a2 - x_a1);\n"
        "            if (ubool(max < lk)) {\n                TS_ADD(16, 1);\n                max = proposed;")
    rep("        t++;\n        since_rebuild++;\n        const bool rebuilt = accept || since_rebuild >= 65536u;",
        "        TS_ADD(11, __builtin_readcyclecounter() - x_t3);\n        t++;\n        since_rebuild++;\n        const bool rebuilt = accept || since_rebuild >= 65536u;")
    rep("        const uint32_t pos_next = no_draw ? pos_v : pos_v + 1;\n        if (reload || pos_next - wd.base >= 64) {\n            gwindow_load(wd, rng, pos_next, lane);\n            p = 0;\n            hopw = hop_words(wd);\n        } else {\n            p = pos_next - wd.base;\n            if (rebuilt) hopw = hop_words(wd);\n        }\n    }",
        "        const uint32_t pos_next = no_draw ? pos_v : pos_v + 1;\n        const unsigned long long x_t5 = __builtin_readcyclecounter();\n"
        "        if (reload || pos_next - wd.base >= 64) {\n            gwindow_load(wd, rng, pos_next, lane);\n            p = 0;\n            hopw = hop_words(wd);\n        } else {\n            p = pos_next - wd.base;\n            if (rebuilt) hopw = hop_words(wd);\n        }\n"
        "        TS_ADD(12, __builtin_readcyclecounter() - x_t5);\n    }")
    rep('        printf("TABSTAT chunk %u K %d n %u D %u steps %u fast %llu events %llu accepts %llu reloads %llu scalars %llu cyc_rebuild %llu cyc_event %llu cyc_total %llu\\n",\n               blockIdx.x, K, n, D, total, ts[0], ts[1], ts[2], ts[3], ts[4], ts[5], ts[6], __builtin_readcyclecounter() - ts_t0);',
        '        printf("TABX K %d n %u D %u steps %u events %llu accepts %llu setup %llu getlk %llu bern %llu book %llu publish %llu hops %llu exact_exp %llu event %llu total %llu acc_state %llu acc_label %llu new_max %llu\\n",\n               K, n, D, total, ts[1], ts[2], ts[8], ts[9], ts[10], ts[11], ts[5], ts[12], ts[13], ts[6], __builtin_readcyclecounter() - ts_t0, ts[14], ts[15], ts[16]);')
    if "--uni" in sys.argv:   # the candidate fix: the size table is indexed with a number the compiler can see is wave-uniform
        rep("    auto size_lk = [&](uint32_t x) -> double {\n        return big ? unif64(m.size_to_lk[x])",
            "    auto size_lk = [&](uint32_t x) -> double {\n        x = uni(x);\n        return big ? unif64(m.size_to_lk[x])")
    if "--lds-size" in sys.argv:   # the candidate fix: the per-size table always in LDS (one load, no branches) instead of 4 registers
        at = s.index("double mcmc_chain_tab(LdsShape shape")
        head, tail = s[:at], s[at:]
        a = "    if (big) {\n        for (uint32_t x = lane; x <= n; x += 64) {\n            double mx = -__builtin_inf();\n            for (int c = 1; c <= K; c++) {"
        b = "        return big ? unif64(m.size_to_lk[x]) : (small ? tab_get<true>(size_to_lk, x) : tab_get<false>(size_to_lk, x));"
        assert tail.count(a) == 1 and tail.count(b) == 1
        tail = tail.replace(a, a.replace("if (big) {", "{")).replace(b, "        return unif64(m.size_to_lk[x]);")
        s = head + tail
    return s


def main():
    src_dir = os.path.join(jbuild.OUT_DIR, "exp_tabx", "csrc")
    shutil.rmtree(os.path.dirname(src_dir), ignore_errors=True)
    shutil.copytree(jbuild.CSRC, src_dir)
    p = os.path.join(src_dir, "mcmc_kernels.hip")
    text = patch(open(p).read())
    open(p, "w").write(text)
    out_dir = os.path.dirname(src_dir)
    objs, procs = [], []
    for f in jbuild.SOURCES:
        obj = os.path.join(out_dir, os.path.splitext(f)[0] + ".o")
        objs.append(obj)
        flags = [x if x != "-I" + jbuild.CSRC else "-I" + src_dir for x in jbuild.FLAGS] + ["-DJTK_MCMC_STATS"]
        cmd = [jbuild.HIPCC] + flags + (["-x", "hip"] if f.endswith(".hip") else []) + ["-c", os.path.join(src_dir, f), "-o", obj]
        procs.append((f, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for f, pr in procs:
        o, _ = pr.communicate()
        if pr.returncode != 0:
            sys.stderr.write(o.decode()[-3000:])
            raise SystemExit("hipcc failed on " + f)
    lib = os.path.join(out_dir, "libjtk_lc_tabx.so")
    subprocess.check_call([jbuild.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
