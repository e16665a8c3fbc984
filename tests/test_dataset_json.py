"""The DataSet JSON stage (jtk_amd/dataset.py): wire-format helpers on the CPU; on the GPU the whole stage on a JSON
object against the C++ host mirror (tests/cpp/host_mirror_main.cpp) fed the same synthetic pile-ups -- two
independent restatements of mod.rs:33-83 + normalize.rs on top of the same C ABI."""
import copy
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from jtk_amd import batch as jb, dataset as D, ffi, synth

import test_host_mirror as HM


def test_cigar_round_trip_and_match_mismatch_merge():
    ops = D.cigar_to_ops("3M2D1I4M")
    assert ops.tolist() == [0, 0, 0, 3, 3, 2, 0, 0, 0, 0]
    assert D.ops_to_cigar(ops) == "3M2D1I4M"
    assert D.ops_to_cigar(np.array([0, 1, 1, 0, 3, 2, 2, 1], np.uint8)) == "4M1D2I1M"   # misc.rs:188-225
    assert D.ops_to_cigar(np.zeros(0, np.uint8)) == "" and len(D.cigar_to_ops("")) == 0
    with pytest.raises(ValueError):
        D.cigar_to_ops("3M2X")
    with pytest.raises(ValueError):
        D.cigar_to_ops("M3")


def test_update_coverage_follows_misc_rs():
    def ds_with(counts, coverage):
        reads = [{"nodes": [{"chunk": c} for c, n in counts.items() for _ in range(n)]}]
        return {"coverage": coverage, "encoded_reads": reads}
    ds = ds_with({1: 10, 2: 30, 3: 61}, "NotAvailable")
    D.update_coverage(ds)
    assert ds["coverage"] == {"Estimated": 15.0}          # median node count / 2 (misc.rs:394-407)
    ds = ds_with({1: 10, 2: 30}, {"Estimated": 3.0})
    D.update_coverage(ds)
    assert ds["coverage"] == {"Estimated": 15.0}          # sorted [10, 30][len / 2] = 30
    ds = ds_with({1: 10}, {"Protected": 22.5})
    D.update_coverage(ds)
    assert ds["coverage"] == {"Protected": 22.5}
    with pytest.raises(ValueError):
        D.update_coverage({"coverage": "NotAvailable", "encoded_reads": []})


def synthetic_dataset(n_chunks, tmpl_len, rph):
    """the DataSet host_mirror_main.cpp builds, as the JSON object JTK would write (definitions/src/lib.rs)"""
    cfg = dict(synth.CONFIGS["ont_diploid"])
    cfg.update(tmpl_len=tmpl_len, reads_per_hap=rph)
    hmm = {k: 0.97 if k.endswith("_mat") else 0.01 for k in ("mat_mat", "mat_ins", "mat_del", "ins_mat", "ins_ins",
                                                            "ins_del", "del_mat", "del_ins", "del_del")}
    hmm["mat_emit"] = [0.97 if r == q else 0.01 for r in range(4) for q in range(4)]
    hmm["ins_emit"] = [0.25] * 20
    reads = [{"id": r, "original_length": 0, "leading_gap": "", "trailing_gap": "", "edges": [], "nodes": []}
             for r in range(2 * rph)]
    chunks = []
    for c in range(n_chunks):
        cid, cn, tmpl, rs, ops, strands, truth = synth.make_pileup(c, cfg, min_variants=1, sort=False)
        chunks.append({"id": c, "seq": bytes(tmpl).decode(), "cluster_num": 2, "copy_num": 2, "score": 0.0})
        for r in range(2 * rph):
            reads[r]["nodes"].append({"position_from_start": 0, "chunk": c, "cluster": 0, "seq": bytes(rs[r]).decode(),
                                      "is_forward": bool(strands[r]), "cigar": D.ops_to_cigar(ops[r]),
                                      "posterior": [math.log(0.5)] * 2})
    # a complete DataSet (DataSet::sanity_check, definitions/src/lib.rs:296-358): consecutive nodes joined by edges, one of them
    # overlapping its successor by two bases, gaps at both ends, and the raw reads the encoded reads recover
    raw = []
    for r, read in enumerate(reads):
        read["leading_gap"], read["trailing_gap"] = "ac" * (r % 3), "G" * (r % 2)
        for a, b in zip(read["nodes"], read["nodes"][1:]):
            read["edges"].append({"from": a["chunk"], "to": b["chunk"], "offset": 3, "label": "TTg"})
        if read["edges"]:
            read["edges"][0].update(offset=-2, label="")
        seq = D.recover_raw_read(read)
        read["original_length"] = len(seq)
        raw.append({"name": f"read{r}", "desc": "", "id": r, "seq": seq})
    return {"input_file": "synthetic", "masked_kmers": {"k": 0, "thr": 0}, "coverage": {"Protected": float(rph)},
            "raw_reads": raw, "hic_pairs": [], "selected_chunks": chunks, "encoded_reads": reads, "hic_edges": [],
            "read_type": "ONT", "model_param": {"forward": hmm, "reverse": copy.deepcopy(hmm)},
            "error_rate": {"del": 0.01, "del_sd": 0.0, "ins": 0.01, "ins_sd": 0.0, "mismatch": 0.01, "mism_sd": 0.0,
                           "total": 0.03, "total_sd": 0.0},
            "processed_stages": [{"stage_name": "encode", "arg": []}]}


def test_sanity_check_accepts_a_complete_dataset_and_names_every_violation():
    """DataSet::sanity_check (definitions/src/lib.rs:296-358), which correct_clustering ends with
    (phmm_likelihood_correction.rs:29): every assert of the reference is a ValueError here"""
    ds = synthetic_dataset(3, 120, 3)
    D.sanity_check(ds)
    rev = next(n for r in ds["encoded_reads"] for n in r["nodes"] if not n["is_forward"])
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    assert D.recover_raw_read({"leading_gap": "", "trailing_gap": "", "edges": [], "nodes": [rev]}) == \
        "".join(comp[b] for b in reversed(rev["seq"]))

    def broken(change):
        bad = copy.deepcopy(ds)
        change(bad)
        with pytest.raises(ValueError, match="sanity_check"):
            D.sanity_check(bad)
    broken(lambda d: d["encoded_reads"][0]["nodes"][0].update(chunk=99))                  # node on an unselected chunk
    broken(lambda d: d["raw_reads"].pop(1))                                               # encoded read without raw read
    broken(lambda d: d["raw_reads"][2].update(seq=d["raw_reads"][2]["seq"][:-1] + "N"))   # not recovered
    broken(lambda d: d["encoded_reads"][3].update(original_length=1))
    broken(lambda d: d["encoded_reads"][0]["edges"][1].update(**{"from": 77}))            # edge does not follow its node
    broken(lambda d: d["selected_chunks"].append(copy.deepcopy(d["selected_chunks"][0])))  # duplicate chunk id
    broken(lambda d: d["selected_chunks"][1].update(cluster_num=3))                       # cluster_num > copy_num
    broken(lambda d: d["encoded_reads"][4]["nodes"][2].update(cluster=3))                 # cluster beyond cluster_num
    lower = copy.deepcopy(ds)
    lower["raw_reads"][0]["seq"] = lower["raw_reads"][0]["seq"].lower()                   # case is not compared (:333,:340)
    D.sanity_check(lower)


def test_training_pileups_are_truncated_before_unknown_chunks_are_dropped():
    """model_tune.rs:99-133: sort by id, take 5, THEN drop pile-ups whose chunk is not selected"""
    from jtk_amd import dataset
    piles = {cid: [object()] * 30 for cid in range(1, 9)}
    piles[0] = [object()] * 30          # nodes of a chunk that is not among the selected chunks
    piles[9] = [object()] * 80          # coverage far from the median: never a training pile-up
    chunk_of = {cid: {} for cid in range(1, 10)}
    assert dataset.training_pileup_ids(piles, chunk_of) == [1, 2, 3, 4]
    del piles[0]
    assert dataset.training_pileup_ids(piles, chunk_of) == [1, 2, 3, 4, 5]


def test_synthetic_dataset_has_exactly_the_reference_fields():
    """every struct of the wire format carries exactly the fields of its serde derive (definitions/src/lib.rs), in
    particular ErrorRate's `mism_sd` (:906); a missing field or an unknown enum variant is rejected like serde does"""
    ds = synthetic_dataset(2, 120, 2)
    D.validate(ds)
    assert tuple(ds) == D.SCHEMA["DataSet"]
    assert set(ds["error_rate"]) == set(D.SCHEMA["ErrorRate"])
    assert set(ds["selected_chunks"][0]) == set(D.SCHEMA["Chunk"])
    assert set(ds["encoded_reads"][0]) == set(D.SCHEMA["EncodedRead"])
    assert set(ds["encoded_reads"][0]["nodes"][0]) == set(D.SCHEMA["Node"])
    assert set(ds["model_param"]["forward"]) == set(D.SCHEMA["HMMParam"])
    assert set(ds["processed_stages"][0]) == set(D.SCHEMA["ProcessedStage"])
    assert set(ds["masked_kmers"]) == set(D.SCHEMA["MaskInfo"])
    # the derive field lists themselves, against the reference source when it is there (not on the GPU box)
    ref = "/root/reference/definitions/src/lib.rs"
    if os.path.exists(ref):
        import re
        src = open(ref).read()
        for struct, fields in D.SCHEMA.items():
            body = re.search(r"pub struct %s \{(.*?)\n\}" % struct, src, flags=re.S).group(1)
            assert tuple(re.findall(r"^\s*pub ([a-z_0-9]+):", body, flags=re.M)) == fields, struct
    for breakage in ("error_rate.mism_sd", "model_param.reverse", "selected_chunks.0.copy_num", "encoded_reads.1.nodes.0.cigar"):
        bad = copy.deepcopy(ds)
        path = breakage.split(".")
        obj = bad
        for k in path[:-1]:
            obj = obj[int(k)] if isinstance(obj, list) else obj[k]
        del obj[path[-1]]
        with pytest.raises(ValueError):
            D.validate(bad)
    bad = copy.deepcopy(ds)
    bad["read_type"] = "PacBio"
    with pytest.raises(ValueError):
        D.validate(bad)
    bad = copy.deepcopy(ds)
    bad["coverage"] = {"Guessed": 3.0}
    with pytest.raises(ValueError):
        D.validate(bad)


def test_untouched_fields_survive_the_json_round_trip(tmp_path):
    ds = synthetic_dataset(1, 120, 2)
    text = json.dumps(ds)
    again = json.loads(text)
    assert again == ds and again["encoded_reads"][0]["nodes"][0]["cigar"] == ds["encoded_reads"][0]["nodes"][0]["cigar"]


@pytest.mark.gpu
@pytest.mark.parametrize("n_selected,refit", [(3, False), (2, False), (3, True)])
def test_json_stage_matches_the_cpp_host_mirror(jtk_lib, tmp_path, capsys, n_selected, refit):
    assert jtk_lib.jtk_lc_device_ok(0) == 1
    n_chunks, tmpl_len, rph = 3, 400, 8
    ds = synthetic_dataset(n_chunks, tmpl_len, rph)
    src, dst = tmp_path / "in.json", tmp_path / "out.json"
    src.write_text(json.dumps(ds))
    argv = [str(src), str(dst)] + (["--chunks", ",".join(str(c) for c in range(n_selected))] if n_selected < n_chunks else [])
    argv += [] if refit else ["--no-refit"]           # refit: update_models_on_both_strands on the device (mod.rs:58)
    argv += ["--verbose"]                             # the reference's RECORD lines (mod.rs:121) on stderr
    argv += ["--trace"] if not refit else []          # and its trace! rows (TOTAL / CAND / PICK / DUMP / RANGE / LK / COUNTS)
    D.main(argv)                                      # gains: estimate_gain_default on the device (mod.rs:60)
    out = json.loads(dst.read_text())
    err = capsys.readouterr().err.splitlines()
    rec = [l.split("\t") for l in err if l.startswith("RECORD\t")]
    if not refit:   # one TOTAL row per clustered chunk; where a cluster count was accepted, its COUNTS row adds up to the pile-up
        assert sum(l.startswith("TOTAL\t") for l in err) == n_selected
        for l in err:
            if l.startswith("COUNTS\t"):
                assert sum(int(x) for x in l.split("\t")[1].strip("[]").split(", ")) == 2 * rph
    exe = HM.build_driver()
    ref = subprocess.run([exe, str(n_chunks), str(tmpl_len), str(rph), "-", str(n_selected), "1" if refit else "0"],
                         capture_output=True, text=True,
                         env=dict(os.environ, JTK_HOST_MIRROR_RECORD="1", **({} if refit else {"JTK_HOST_MIRROR_TRACE": "1"})))
    assert ref.returncode == 0, ref.stderr
    if not refit:   # the trace! rows of the two host mirrors are the same text
        kinds = ("TOTAL\t", "CAND\t", "PICK\t", "DUMP\t", "RANGE\t", "LK\t", "COUNTS\t")
        assert [l for l in err if l.startswith(kinds)] == [l for l in ref.stderr.splitlines() if l.startswith(kinds)]
    # RECORD\tchunk id\telapsed ms\tpolish ms\tconsensus length\tscore (3 decimals)\tcoverage -- one per clustered chunk, from
    # the stage itself, in both host mirrors (the milliseconds are a share of the call's kernel time: not compared)
    rec_cpp = [l.split("\t") for l in ref.stderr.splitlines() if l.startswith("RECORD\t")]
    assert len(rec) == n_selected == len(rec_cpp)
    for a, b in zip(rec, rec_cpp):
        assert len(a) == 7 and (a[1], a[4], a[5], a[6]) == (b[1], b[4], b[5], b[6])
        assert float(a[2]) >= float(a[3]) >= 0.0 and int(a[6]) == 2 * rph
    chunks, nodes = {}, {}
    for line in ref.stdout.splitlines():
        f = line.split("\t")
        if f[0] == "CHUNK":
            chunks[int(f[1])] = (int(f[2]), float(f[3]), f[4])
        else:
            nodes[(int(f[1]), int(f[2]))] = (int(f[3]), f[4], [float(x) for x in f[5:]])
    for c in out["selected_chunks"]:
        k, score, seq = chunks[c["id"]]
        assert (c["cluster_num"], c["seq"]) == (k, seq) and c["score"] == score
    for r, read in enumerate(out["encoded_reads"]):
        for node in read["nodes"]:
            cl, cig, ps = nodes[(r, node["chunk"])]
            assert (node["cluster"], node["cigar"]) == (cl, cig) and node["posterior"] == ps
    # everything the stage does not own is passed through
    for key in ("input_file", "masked_kmers", "raw_reads", "hic_pairs", "hic_edges", "read_type", "error_rate",
                "processed_stages", "coverage"):
        assert out[key] == ds[key]
    # the stage overwrites model_param with the refitted model (model_tune.rs:20-25); without the refit it is passed through
    assert (out["model_param"] == ds["model_param"]) == (not refit)
    if refit:
        f = out["model_param"]["forward"]
        assert abs(f["mat_mat"] + f["mat_ins"] + f["mat_del"] - 1.0) < 1e-12 and len(f["ins_emit"]) == 20


@pytest.mark.gpu
def test_json_correct_clustering_stage_matches_the_oracle(jtk_lib, tmp_path):
    """`python -m jtk_amd.dataset --stage correct_clustering`: local clustering first (so that nodes carry real posteriors),
    then AlignmentCorrection::correct_clustering (phmm_likelihood_correction.rs:14-97) on the file; the node labels,
    posteriors and cluster_num it writes equal what oracle/correction.c computes from the same DataSet."""
    import oracle_ffi as O
    from jtk_amd import api, ffi
    ds = synthetic_dataset(5, 220, 8)
    D.local_clustering(ds, refit=False)
    before = copy.deepcopy(ds)
    min_gain = api.estimate_minimum_gain(D._hmm(ds["model_param"]["forward"]), D._hmm(ds["model_param"]["reverse"]))
    assert min_gain >= 1.0
    src, dst = tmp_path / "in.json", tmp_path / "out.json"
    src.write_text(json.dumps(before))
    assert D.main(["--stage", "correct_clustering", str(src), str(dst)]) == 0
    after = json.loads(dst.read_text())
    # the same through the oracle
    nodes, post, node_off, read_id = [], [], [0], []
    for read in before["encoded_reads"]:
        read_id.append(read["id"])
        for n in read["nodes"]:
            nodes.append((n["chunk"], n["cluster"], 1 if n["is_forward"] else 0, len(n["posterior"]), len(post)))
            post.extend(n["posterior"])
        node_off.append(len(nodes))
    nodes = np.array(nodes, dtype=ffi.CC_NODE_DT)
    chunks = np.zeros(len(before["selected_chunks"]), dtype=ffi.CC_CHUNK_DT)
    for i, c in enumerate(before["selected_chunks"]):
        chunks[i] = (c["id"], c["cluster_num"], c["copy_num"], c["score"])
    sel = [c["id"] for c in before["selected_chunks"] if c["cluster_num"] > 1]
    _, cov = D.coverage_of(before)
    rc, cluster, touched, _, _ = O.correct_clustering(read_id, node_off, nodes, post, chunks, sel, cov, min_gain)
    assert rc == 0
    e = 0
    k_of = {int(c["id"]): int(c["cluster_num"]) for c in chunks}
    assert {c["id"]: c["cluster_num"] for c in after["selected_chunks"]} == k_of
    for rb, ra in zip(before["encoded_reads"], after["encoded_reads"]):
        for nb, na in zip(rb["nodes"], ra["nodes"]):
            if touched[e]:
                want = [-10000.0] * k_of[nb["chunk"]]
                want[int(cluster[e])] = 0.0
                assert na["cluster"] == int(cluster[e]) and na["posterior"] == want
            else:
                assert na["cluster"] == nb["cluster"] and na["posterior"] == nb["posterior"]
            e += 1
    assert touched.any()
