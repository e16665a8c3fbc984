"""The reference's trace! rows of a chunk's clustering (`LK`, `CAND`, `PICK`; pseudo_mcmc.rs:122-127 DUMP, :236 RANGE,
:250,:256 LK, :262 COUNTS, :467 TOTAL, :471 CAND, :539 PICK).  The oracle writes them where the reference's trace! calls stand
(oracle/pseudo_mcmc.c); the device re-runs the chunk's pick and chain in recording instantiations (jtk_lc_session_trace) and must
produce the same text, row for row.  CPU: the oracle's rows agree with the oracle's own results and with the Rust format strings.
GPU: device text == oracle text."""
import ctypes as C
import re

import numpy as np
import pytest

import helpers
import oracle_ffi as O
from jtk_amd import api, batch as jb, ffi, synth


def _params(cfg):
    return jb.default_params(cfg["coverage"], cfg["band_frac"])


def _check_rows_against_result(rows, res, n_reads):
    """what the rows must say given the chunk's result record (score, cluster_num, n_variants)"""
    kinds = [r.split("\t")[0] for r in rows]
    assert kinds[0] == "TOTAL"
    total = int(rows[0].split("\t")[1])
    cands = [r.split("\t") for r in rows if r.startswith("CAND\t")]
    picks = [r.split("\t") for r in rows if r.startswith("PICK\t")]
    dumps = [r.split("\t") for r in rows if r.startswith("DUMP\t")]
    assert len(cands) == total
    assert len(dumps) == int(res["n_variants"]) <= len(picks) <= total
    # the order the reference logs in: TOTAL, CAND*, PICK*, DUMP*, then RANGE, (LK, LK, COUNTS?)*
    assert re.fullmatch(r"T C* P* D* (R (L L N?)*)?".replace(" ", ""),
                        "".join({"TOTAL": "T", "CAND": "C", "PICK": "P", "DUMP": "D", "RANGE": "R", "LK": "L", "COUNTS": "N"}[k]
                                for k in kinds))
    cand_cols = [(int(c[1]), int(c[2])) for c in cands]
    assert cand_cols == sorted(cand_cols)                                      # candidates in column order
    assert [(int(d[2]), int(d[3])) for d in dumps] == sorted((int(d[2]), int(d[3])) for d in dumps)   # the selected ones too
    for c in cands:
        assert 0 <= int(c[2]) < 14 and re.fullmatch(r"-?\d+\.\d", c[3]) and 0 < int(c[4]) <= n_reads
    for p in picks:
        assert p[2] in ("S", "I", "D") and re.fullmatch(r"-?\d+\.\d{3}", p[3])
    for i, d in enumerate(dumps):
        assert int(d[1]) == i and (d[2], d[3]) in [(c[1], c[2]) for c in cands]
        assert d[4] == [c[3] for c in cands if (c[1], c[2]) == (d[2], d[3])][0]   # the candidate's lk, {:.1} both times
    if "RANGE" in kinds:
        lo, hi = map(int, rows[kinds.index("RANGE")].split("\t")[1].split("..="))
        lks = [r.split("\t") for r in rows if r.startswith("LK\t")]
        assert len(lks) % 2 == 0 and all(len(a) == 3 and len(b) == 5 and a[1:3] == b[1:3] for a, b in zip(lks[::2], lks[1::2]))
        ks = [int(a[1]) for a in lks[::2]]
        assert ks == list(range(lo, lo + len(ks))) and ks[-1] <= hi
        counts = [r for r in rows if r.startswith("COUNTS\t")]
        if counts:
            last = [int(x) for x in counts[-1].split("\t")[1].strip("[]").split(", ")]
            assert len(last) == int(res["cluster_num"]) and sum(last) == n_reads
            accepted_lk = lks[2 * (len(counts) - 1)][2]
            assert accepted_lk == "%.3f" % float(res["score"])                # the accepted k's score is the chunk's
        else:
            assert int(res["cluster_num"]) == 1 and float(res["score"]) == 0.0
    else:
        assert int(res["cluster_num"]) == 1 and float(res["score"]) == 0.0


def test_oracle_trace_rows_agree_with_its_results(oracle):
    b, cfg = synth.make_batch("ont_diploid", 3)
    p = helpers.oracle_params(_params(cfg))
    plain = O._cluster_chunks_live(p, b, n_threads=0)
    seen_range = False
    for c in range(3):
        out, rows = O.trace_chunk(p, b, c)
        assert out["rc"] == 0
        # tracing changes nothing
        assert np.array_equal(out["label"], plain["label"][b.chunks["read_first"][c]:][:b.chunks["n_reads"][c]])
        assert helpers.bits(out["result"]["score"])[0] == helpers.bits(plain["result"]["score"])[c]
        _check_rows_against_result(rows, out["result"][0], int(b.chunks["n_reads"][c]))
        seen_range = seen_range or any(r.startswith("RANGE") for r in rows)
    assert seen_range, "none of the three pile-ups had a variant column: the LK rows went unchecked"
    # no sink, no rows, same answers (the sink is cleared after a traced chunk)
    again = O._cluster_chunks_live(p, b, n_threads=0)
    assert np.array_equal(again["label"], plain["label"])


def test_oracle_rows_follow_the_rust_format_strings(oracle):
    """RANGE is {:?} of a RangeInclusive, COUNTS {:?} of a Vec<usize>; a copy-number-1 chunk logs nothing (pseudo_mcmc.rs:86-88).
    ({:.1} / {:.3} of the lk fields: _check_rows_against_result.)"""
    b, cfg = synth.make_batch("ont_diploid", 1)
    p = helpers.oracle_params(_params(cfg))
    one = b.subset([0])
    one.chunks["copy_num"][0] = 1
    _, rows = O.trace_chunk(p, one, 0)
    assert rows == []
    out, rows = O.trace_chunk(p, b, 0)
    rng = [r for r in rows if r.startswith("RANGE")]
    assert rng == ["RANGE\t2..=2"]                       # copy_num 2: start = max(end, 5) - 3 = 2 = end
    counts = [r for r in rows if r.startswith("COUNTS")]
    assert counts and re.fullmatch(r"COUNTS\t\[\d+, \d+\]", counts[0])
    lab = out["label"]
    # the COUNTS row is the k = 2 clustering BEFORE the re-assignment of clustering()'s tail (:98-105); the sizes still add up
    assert sum(int(x) for x in counts[0].split("\t")[1].strip("[]").split(", ")) == len(lab)


@pytest.mark.parametrize("config", ["ont_diploid", "ont_4copy"])
def test_cand_and_dump_rows_against_a_numpy_restatement(oracle, config):
    """CAND's lk and count (pseudo_mcmc.rs:457-461 + column_sum :577-588) and DUMP's sum (:124) recomputed here in numpy from the
    oracle's per-read tables: table - lk, compress_small_gains (:141-165: |x| below half the expected gain of the column's type
    and homopolymer length -> 0), per column the sum and count of the gains above POS_THR, lk = max_k Poisson(count | k * coverage)
    + sum.  A second statement of what the rows mean, independent of the C that writes them."""
    import math
    b, cfg = synth.make_batch(config, 1)
    p = _params(cfg)
    po = helpers.oracle_params(p)
    ks = range(1, int(b.chunks["copy_num"][0]) + 1)         # the cluster counts of the Poisson term (:457-460)
    _, rows = O.trace_chunk(po, b, 0, skip_polish=True)     # skip_polish: the template the tables are taken on is the one given
    cands = [r.split("\t") for r in rows if r.startswith("CAND\t")]
    dumps = [r.split("\t") for r in rows if r.startswith("DUMP\t")]
    assert cands and dumps
    tmpl = b.template(0)
    tl, n = len(tmpl), int(b.chunks["n_reads"][0])
    radius = int(math.ceil(tl * p.band_frac)) // 2          # mod.rs:96,112
    tabs = []
    for r in b.chunk_reads(0):
        hmm = po.forward if b.strand[r] else po.reverse
        tab, lk = O.modification_table(hmm, tmpl, b.read(r), b.read_ops(r), radius)
        tabs.append(tab - lk)
    prof = np.array(tabs)                                   # n x 14 (tl + 1)
    homop = np.ones(tl, dtype=np.int64)                     # homopolymer_length :195-211
    i = 0
    while i < tl:
        j = i
        while j + 1 < tl and tmpl[j + 1] == tmpl[i]:
            j += 1
        homop[i:j + 1] = j - i + 1
        i = j + 1

    def expected(homop_len, row):                           # Gains::expected, likelihood_gains.rs:79-87; difftype :168-178
        g = p.gains
        h = min(max(int(homop_len), 1), int(g.max_homopolymer_len))
        tab = g.subst if row < 4 else (g.insertions if row < 8 + 3 else g.deletions)
        return tab[h - 1].gain

    cov = float(p.haploid_coverage)
    for c in cands:
        bp, row, lk_txt, count_txt = int(c[1]), int(c[2]), c[3], int(c[4])
        mr = expected(homop[bp] if bp < tl else 1, row) * 0.5
        col = prof[:, bp * 14 + row].copy()
        col[np.abs(col) < mr] = 0.0
        gain, count = 0.0, 0
        for x in col:                                       # left to right, as the reference sums
            if 0.00001 < x:
                gain += float(x)
                count += 1
        assert count == count_txt
        pois = max(count * math.log(cov * k) - cov * k - sum(math.log(q) for q in range(1, count + 1)) for k in ks)
        assert "%.1f" % (pois + gain) == lk_txt, (bp, row, pois + gain, lk_txt)
    for d in dumps:
        bp, row = int(d[2]), int(d[3])
        mr = expected(homop[bp] if bp < tl else 1, row) * 0.5
        col = prof[:, bp * 14 + row].copy()
        col[np.abs(col) < mr] = 0.0
        tot = 0.0
        for x in col:
            tot += max(float(x), 0.0)
        assert "%.1f" % tot == d[5]
    # PICK rows: pick_filtered_profiles (:516-575) restated -- three rounds of max(copy_num, 2) picks, each the LAST maximum among
    # the candidates still open (find_next_variants :590-600), a pick closes the candidates within MASK_LENGTH bp for good and those
    # that resemble it (Sokal-Michener or |cosine| above 0.8 over the reads where both columns are non-zero) until the next round
    def column(c):
        bp, row = int(c[1]), int(c[2])
        col = prof[:, bp * 14 + row].copy()
        col[np.abs(col) < expected(homop[bp] if bp < tl else 1, row) * 0.5] = 0.0
        return col
    cols = [column(c) for c in cands]
    # the scores at full precision are not in the rows: recompute them as above (they decide the order)
    score = []
    for c, col in zip(cands, cols):
        g = float(sum(float(x) for x in col if 0.00001 < x))
        cnt = int(c[4])
        score.append(max(cnt * math.log(cov * k) - cov * k - sum(math.log(q) for q in range(1, cnt + 1)) for k in ks) + g)
    sel = [0] * len(cands)
    order = []
    for _round in range(3):
        sel = [0 if f == 3 else f for f in sel]
        for _ in range(max(int(b.chunks["copy_num"][0]), 2)):
            open_ = [i for i, f in enumerate(sel) if f == 0]
            if not open_:
                break
            nx = max(reversed(open_), key=lambda i: score[i])          # max_by: the last maximum
            order.append(nx)
            sel[nx] = 1
            for i, f in enumerate(sel):
                if f not in (0, 3):
                    continue
                if abs(int(cands[i][1]) - int(cands[nx][1])) < 7:       # MASK_LENGTH
                    sel[i] = 2
                    continue
                x, y = cols[nx], cols[i]
                both = (np.abs(x) > 0.00001) & (np.abs(y) > 0.00001)
                mat = int(((x * y > 0) & both).sum())
                tot = int(both.sum())
                sok = 0.0 if tot == 0 else max(mat, tot - mat) / tot
                isq, jsq = float((x[both] ** 2).sum()), float((y[both] ** 2).sum())
                cs = 0.0 if isq == 0.0 else float((x[both] * y[both]).sum()) / math.sqrt(isq) / math.sqrt(jsq)
                if 0.8 < sok or 0.8 < abs(cs):
                    sel[i] = 3
    picks = [r.split("\t") for r in rows if r.startswith("PICK\t")]
    letter = lambda row: "S" if row < 4 else ("I" if row < 11 else "D")
    assert [(p_[1], p_[2]) for p_ in picks] == [(cands[i][1], letter(int(cands[i][2]))) for i in order]
    assert [p_[3] for p_ in picks] == ["%.3f" % score[i] for i in order]
    assert sorted(i for i, f in enumerate(sel) if f == 1) == sorted(
        i for i, c in enumerate(cands) if (c[1], c[2]) in [(d[2], d[3]) for d in dumps])       # the selected columns are the DUMP rows
    if config == "ont_4copy":   # the case is there for what the diploid one lacks: candidates closed by a pick
        assert len(cands) > len(dumps) >= 4 and any(f in (2, 3) for f in sel)


def _golden_cases():
    import json
    import os
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, here)
    import make_cfg3_64
    import make_trace_rows as mk
    g = json.load(open(os.path.join(here, "trace_rows.json")))
    for case in g["cases"]:
        b, p = mk.make_inputs(case["config"], case["n_chunks"])
        assert make_cfg3_64.inputs_digest(b) == case["inputs_sha256"], "the generator no longer produces the golden's inputs"
        yield case, b, p


def test_oracle_reproduces_the_golden_trace_rows(oracle):
    """tests/golden/trace_rows.json (tests/golden/make_trace_rows.py): the oracle in the tree still writes the committed rows"""
    n = 0
    for case, b, p in _golden_cases():
        _, rows = O.trace_chunk(helpers.oracle_params(p), b, case["chunk"])
        assert rows == case["rows"], (case["config"], case["chunk"])
        n += 1
    assert n >= 5


@pytest.mark.gpu
def test_device_reproduces_the_golden_trace_rows(jtk_lib):
    """the device's rows against the committed ones: no oracle in the loop"""
    sessions = {}
    try:
        for case, b, p in _golden_cases():
            key = (case["config"], case["n_chunks"])
            if key not in sessions:
                sessions[key] = api.Session(p, b)
                sessions[key].run()
            assert sessions[key].trace(case["chunk"]) == case["rows"], (case["config"], case["chunk"])
    finally:
        for s in sessions.values():
            s.close()


@pytest.mark.gpu
def test_device_trace_rows_match_the_oracle(jtk_lib, oracle):
    """diploid ONT (light chain), HiFi (pair kernel) and a 4-copy pile-up (K-way chain, k = 2 .. 4 tried): the session's rows are
    the oracle's, byte for byte, and the session's results are untouched by the recording re-run"""
    # (a 4-copy chunk's recording chain takes ~23 s on the device: one here, another one in the golden test above)
    cases = [("ont_diploid", 6, [0, 1, 2, 5]), ("hifi_diploid", 2, [0, 1]), ("ont_4copy", 2, [1])]
    seen_lk = 0
    for name, n_chunks, which in cases:
        b, cfg = synth.make_batch(name, n_chunks)
        p = _params(cfg)
        with api.Session(p, b) as s:
            s.run()
            before = s.fetch()
            for c in which:
                dev_rows = s.trace(c)
                _, ora_rows = O.trace_chunk(helpers.oracle_params(p), b, c)
                assert dev_rows == ora_rows, (name, c, dev_rows, ora_rows)
                _check_rows_against_result(dev_rows, before["result"][c], int(b.chunks["n_reads"][c]))
                seen_lk += sum(r.startswith("LK\t") for r in dev_rows)
            after = s.fetch()
            for k in ("label", "log_post", "result", "cons", "cons_off", "ops_out", "ops_out_off"):
                assert before[k].tobytes() == after[k].tobytes(), (name, k)
    assert seen_lk >= 6


@pytest.mark.gpu
def test_device_trace_after_skip_polish_and_its_errors(jtk_lib, oracle):
    b, cfg = synth.make_batch("ont_diploid", 2)
    p = _params(cfg)
    L = ffi.lib()
    need = C.c_size_t(0)
    with api.Session(p, b) as s:
        # before any run: nothing to trace
        assert L.jtk_lc_session_trace(s._h, 0, None, 0, C.byref(need)) == -1          # JTK_ERR_INVALID_ARG
        s.run(skip_polish=True)                                                       # pseudo_mcmc::clustering on the draft as given
        rows = s.trace(0)
        _, ora = O.trace_chunk(helpers.oracle_params(p), b, 0, skip_polish=True)
        assert rows == ora
        assert L.jtk_lc_session_trace(s._h, 7, None, 0, C.byref(need)) == -1          # no such chunk
        small = C.create_string_buffer(4)
        assert L.jtk_lc_session_trace(s._h, 0, small, 4, C.byref(need)) == -1 and need.value == sum(len(r) + 1 for r in rows) > 4
        assert L.jtk_lc_session_trace(None, 0, small, 4, C.byref(need)) == -1
    # a session that holds a chunk of copy number >= 8 (clustering_recursive) is refused
    b8, cfg8 = synth.make_batch("ont_4copy", 1, reads_per_hap=10)
    b8.chunks["copy_num"][0] = 8
    with api.Session(_params(cfg8), b8) as s:
        s.run()
        assert L.jtk_lc_session_trace(s._h, 0, None, 0, C.byref(need)) == -3          # JTK_ERR_UNSUPPORTED
