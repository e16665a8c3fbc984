"""The C++ host mirror of `LocalClustering for DataSet` (jtk_amd/csrc/host/local_clustering.hpp), end to end:
pile-up grouping + stable sort, flattening, the GPU stage, update_by_clusterings, chunk write-back and
normalize_local_clustering, against the oracle driven through the same steps in Python."""
import os
import subprocess

import numpy as np
import pytest

import helpers
import oracle_ffi as O
from jtk_amd import batch as jb, ffi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")
EXE = os.path.join(CPP, "host_mirror")


def build_driver():
    src = os.path.join(CPP, "host_mirror_main.cpp")
    hdr = os.path.join(ROOT, "jtk_amd", "csrc", "host", "local_clustering.hpp")
    if not os.path.exists(EXE) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(EXE):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"),
                               "-I" + os.path.join(ROOT, "jtk_amd", "csrc"), src,
                               "-L" + os.path.join(ROOT, "jtk_amd", "_build"), "-ljtk_lc", "-ljtk_synth",
                               "-Wl,-rpath," + os.path.join(ROOT, "jtk_amd", "_build"), "-o", EXE])
    return EXE


def test_host_mirror_compiles(jtk_lib):
    build_driver()
    # without a gains file the driver refuses (the stage needs estimate_gain_default's result)
    assert subprocess.call([EXE, "1", "200", "4", "/nonexistent", "1"]) == 3


def runs_of(ops):
    out, prev, n = [], None, 0
    for o in ops.tolist():
        k = "D" if o == 3 else ("I" if o == 2 else "M")
        if k == prev:
            n += 1
        else:
            if prev is not None:
                out.append(f"{n}{prev}")
            prev, n = k, 1
    if prev is not None:
        out.append(f"{n}{prev}")
    return "".join(out)


@pytest.mark.gpu
@pytest.mark.parametrize("n_selected,device_gains", [(3, False), (2, False), (3, True)])
def test_host_mirror_matches_oracle(jtk_lib, tmp_path, n_selected, device_gains):
    """device_gains: the stage call runs estimate_gain_default itself (mod.rs:60, jtk_lc_estimate_gains); the result
    must be the golden gains the other cases read from a file"""
    assert jtk_lib.jtk_lc_device_ok(0) == 1
    exe = build_driver()
    n_chunks, tmpl_len, rph = 3, 400, 8
    gains_file = tmp_path / "gains.txt"
    with open(gains_file, "w") as f:
        for name in ("subst", "deletions", "insertions"):
            for g, p in jb.DEFAULT_GAINS[name]:
                f.write(f"{g!r} {p!r}\n")
    out = subprocess.run([exe, str(n_chunks), str(tmpl_len), str(rph), "-" if device_gains else str(gains_file), str(n_selected)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    chunks, nodes = {}, {}
    for line in out.stdout.splitlines():
        f = line.split("\t")
        if f[0] == "CHUNK":
            chunks[int(f[1])] = (int(f[2]), float(f[3]), f[4])
        else:
            nodes[(int(f[1]), int(f[2]))] = (int(f[3]), f[4], [float(x) for x in f[5:]])
    # the same through Python + oracle
    cfg = dict(synth.CONFIGS["ont_diploid"])
    cfg.update(tmpl_len=tmpl_len, reads_per_hap=rph)
    p = jb.default_params(haploid_coverage=float(rph))
    for c in range(n_chunks):
        cid, cn, tmpl, reads, ops, strands, truth = synth.make_pileup(c, cfg, min_variants=1, sort=False)
        order = jb.pileup_sort(tmpl, reads, ops)
        if c >= n_selected:  # not selected: untouched
            assert chunks[c] == (2, 0.0, bytes(tmpl).decode())
            for r in range(2 * rph):
                assert nodes[(r, c)][0] == 0 and nodes[(r, c)][1] == runs_of(ops[r])
            continue
        b = jb.pack([(cid, cn, tmpl, [reads[i] for i in order], [ops[i] for i in order],
                      [strands[i] for i in order], None)])
        ora = O.cluster_chunks(helpers.oracle_params(p), b)
        assert ora["rc"] == 0
        k = int(ora["result"][0]["cluster_num"])
        lab = ora["label"].astype(np.uint64)
        post = np.ascontiguousarray(ora["log_post"][:, :k])
        O.lib().jo_normalize_pileup(len(lab), k, O.u64p(lab), O.f64p(post), k)
        cons = bytes(ora["cons"][:int(ora["cons_off"][1])]).decode()
        assert chunks[c][0] == k and chunks[c][2] == cons
        assert abs(chunks[c][1] - float(ora["result"][0]["score"])) < 1e-4
        for s, r in enumerate(order):
            cl, cig, ps = nodes[(r, c)]
            assert cl == int(lab[s])
            assert np.abs(np.array(ps) - post[s]).max() < 1e-4
            o = ora["ops_out"][int(ora["ops_out_off"][s]):int(ora["ops_out_off"][s + 1])]
            assert cig == runs_of(o)
