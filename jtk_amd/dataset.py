"""File-level drop-in for the stage: `DataSet` JSON in, `DataSet` JSON out.

JTK passes ONE JSON object between its stages (definitions/src/lib.rs:6-35; `jtk ... > x.encoded.json`,
`jtk local_clustering < x.encoded.json > x.clustered.json`, cli/src/pipeline.rs:153-159).  This module reads that
object, runs `LocalClustering::local_clustering{,_selected}` (haplotyper/src/local_clustering/mod.rs:17-83) through
the C ABI and writes the object back; every field the stage does not touch is passed through as parsed.

    python -m jtk_amd.dataset in.encoded.json out.clustered.json [--chunks id,id,...] [--device 0]

Wire format of the fields the stage reads or writes (serde derives of definitions/src/lib.rs):
  coverage         "NotAvailable" | {"Protected": f64} | {"Estimated": f64}          (:46-55, externally tagged enum)
  read_type        "CCS" | "CLR" | "ONT" | "None"                                     (:156-162)
  model_param      {"forward": HMMParam, "reverse": HMMParam}; HMMParam = 9 transition fields + mat_emit[16] +
                   ins_emit[20]                                                        (:95-126)
  selected_chunks  [{"id", "seq": "ACGT..", "cluster_num", "copy_num", "score"}]       (:403-415, DNASeq as a string)
  encoded_reads    [{.., "nodes": [{"position_from_start", "chunk", "cluster", "seq": "ACGT..", "is_forward",
                   "cigar": "10M2D3I", "posterior": [f64]}]}]                          (:672-683, Ops as a string :825-877)

Both preambles run on the device: `update_models_on_both_strands` (mod.rs:58, model_tune.rs:96-156: training pile-ups
picked here, ten rounds of polish + Baum-Welch in `jtk_lc_fit_model`; `--no-refit` keeps `model_param` as found in the
file) and the gains calibration (mod.rs:60, `jtk_lc_estimate_gains`).  This is the same host logic as the C++ mirror `csrc/host/local_clustering.hpp`; the two
are tested against each other (tests/test_dataset_json.py)."""
import argparse
import ctypes as C
import json
import re
import sys

import numpy as np

from . import api, ffi
from .batch import pack, pileup_sort

BAND_FRAC = {"CCS": 0.01, "ONT": 0.03, "CLR": 0.05, "None": 0.05}   # definitions/src/lib.rs:173-175

# Field names of the serde derives of definitions/src/lib.rs (no #[serde(rename/default/skip)] anywhere in that file, so
# the wire names are the Rust names and every field is required on input): struct -> fields in declaration order.
SCHEMA = {
    "DataSet": ("input_file", "masked_kmers", "coverage", "raw_reads", "hic_pairs", "selected_chunks", "encoded_reads",
                "hic_edges", "read_type", "model_param", "error_rate", "processed_stages"),                 # :6-34
    "ProcessedStage": ("stage_name", "arg"),                                                               # :38-43
    "HMMParamOnStrands": ("forward", "reverse"),                                                           # :95-99
    "HMMParam": ("mat_mat", "mat_ins", "mat_del", "ins_mat", "ins_ins", "ins_del", "del_mat", "del_ins", "del_del",
                 "mat_emit", "ins_emit"),                                                                  # :101-126
    "MaskInfo": ("k", "thr"),
    "RawRead": ("name", "desc", "id", "seq"),
    "HiCPair": ("pair1", "pair2", "pair_id", "seq1", "seq2"),
    "Chunk": ("id", "seq", "cluster_num", "copy_num", "score"),                                            # :403-415
    "EncodedRead": ("id", "original_length", "leading_gap", "trailing_gap", "edges", "nodes"),
    "Edge": ("from", "to", "offset", "label"),
    "Node": ("position_from_start", "chunk", "cluster", "seq", "is_forward", "cigar", "posterior"),        # :672-683
    "HiCEdge": ("pair_id", "pair1", "pair2"),
    "ErrorRate": ("del", "del_sd", "ins", "ins_sd", "mismatch", "mism_sd", "total", "total_sd"),           # :900-909
}
READ_TYPES = ("CCS", "CLR", "ONT", "None")                                                                 # :156-162


def _need(obj, struct, where):
    if not isinstance(obj, dict):
        raise ValueError(f"{where}: expected a {struct} object")
    missing = [k for k in SCHEMA[struct] if k not in obj]
    if missing:   # serde: "missing field `x`"
        raise ValueError(f"{where}: {struct} is missing field(s) {missing}")


def validate(ds):
    """What serde_json would reject on the way into `DataSet` (definitions/src/lib.rs): missing fields, a coverage or
    read type that is not one of the enum's variants.  Unknown extra fields are ignored, as serde does by default."""
    _need(ds, "DataSet", "dataset")
    _need(ds["masked_kmers"], "MaskInfo", "masked_kmers")
    cov = ds["coverage"]
    if not (cov == "NotAvailable" or (isinstance(cov, dict) and len(cov) == 1 and next(iter(cov)) in ("Protected", "Estimated"))):
        raise ValueError("coverage: expected \"NotAvailable\", {\"Protected\": x} or {\"Estimated\": x}")
    if ds["read_type"] not in READ_TYPES:
        raise ValueError(f"read_type: unknown variant {ds['read_type']!r}")
    _need(ds["model_param"], "HMMParamOnStrands", "model_param")
    for strand in ("forward", "reverse"):
        _need(ds["model_param"][strand], "HMMParam", f"model_param.{strand}")
    _need(ds["error_rate"], "ErrorRate", "error_rate")
    for i, st in enumerate(ds["processed_stages"]):
        _need(st, "ProcessedStage", f"processed_stages[{i}]")
    for i, r in enumerate(ds["raw_reads"]):
        _need(r, "RawRead", f"raw_reads[{i}]")
    for i, h in enumerate(ds["hic_pairs"]):
        _need(h, "HiCPair", f"hic_pairs[{i}]")
    for i, h in enumerate(ds["hic_edges"]):
        _need(h, "HiCEdge", f"hic_edges[{i}]")
    for i, c in enumerate(ds["selected_chunks"]):
        _need(c, "Chunk", f"selected_chunks[{i}]")
    for i, r in enumerate(ds["encoded_reads"]):
        _need(r, "EncodedRead", f"encoded_reads[{i}]")
        for j, e in enumerate(r["edges"]):
            _need(e, "Edge", f"encoded_reads[{i}].edges[{j}]")
        for j, n in enumerate(r["nodes"]):
            _need(n, "Node", f"encoded_reads[{i}].nodes[{j}]")


_OP_RE = re.compile(r"(\d+)([MDI])")
_OP_CODE = {"M": ffi.OP_MATCH, "D": ffi.OP_DEL, "I": ffi.OP_INS}


def cigar_to_ops(cigar):
    """Ops::from_str (definitions/src/lib.rs:857-877) + ops_to_kiley (misc.rs:177-186): one op code per base."""
    out = []
    pos = 0
    for m in _OP_RE.finditer(cigar):
        if m.start() != pos:
            raise ValueError(f"bad cigar {cigar!r}")
        out.append(np.full(int(m.group(1)), _OP_CODE[m.group(2)], dtype=np.uint8))
        pos = m.end()
    if pos != len(cigar):
        raise ValueError(f"bad cigar {cigar!r}")
    return np.concatenate(out) if out else np.zeros(0, np.uint8)


def ops_to_cigar(ops):
    """kiley_op_to_ops (misc.rs:188-225; Match and Mismatch merge into one M run) + Display for Ops (:827-838)."""
    ops = np.asarray(ops, dtype=np.uint8)
    if len(ops) == 0:
        return ""
    kind = np.where(ops == ffi.OP_DEL, 1, np.where(ops == ffi.OP_INS, 2, 0))
    cut = np.flatnonzero(np.diff(kind)) + 1
    starts = np.concatenate([[0], cut])
    ends = np.concatenate([cut, [len(kind)]])
    return "".join(f"{e - s}{'MDI'[kind[s]]}" for s, e in zip(starts.tolist(), ends.tolist()))


def coverage_of(ds):
    cov = ds["coverage"]
    if isinstance(cov, dict):
        (tag, value), = cov.items()
        return tag, float(value)
    return cov, None


def update_coverage(ds):
    """misc.rs:394-407: unless protected, haploid coverage = median node count per chunk / 2."""
    tag, _ = coverage_of(ds)
    if tag == "Protected":
        return
    counts = {}
    for read in ds["encoded_reads"]:
        for node in read["nodes"]:
            counts[node["chunk"]] = counts.get(node["chunk"], 0) + 1
    if not counts:
        raise ValueError("update_coverage: no nodes")
    c = sorted(counts.values())
    ds["coverage"] = {"Estimated": c[len(c) // 2] / 2.0}


def _hmm(d):
    h = ffi.Hmm()
    for name in ("mat_mat", "mat_ins", "mat_del", "ins_mat", "ins_ins", "ins_del", "del_mat", "del_ins", "del_del"):
        setattr(h, name, float(d[name]))
    if len(d["mat_emit"]) != 16 or len(d["ins_emit"]) != 20:
        raise ValueError("HMMParam: mat_emit must have 16 and ins_emit 20 entries")
    for i, v in enumerate(d["mat_emit"]):
        h.mat_emit[i] = float(v)
    for i, v in enumerate(d["ins_emit"]):
        h.ins_emit[i] = float(v)
    return h


def _seq(s):
    return np.frombuffer(s.encode("ascii"), dtype=np.uint8)


def training_pileup_ids(piles, chunk_of):
    """truncate_bucket + the filter_map of estimate_model_parameters_on_both_strands (model_tune.rs:99-133): pile-ups whose
    coverage lies within 2 of the median, sorted by chunk id, the first TRAIN_UNIT_SIZE = 5 of them, and only THEN without
    those whose chunk is not among the selected chunks (a stray node can cost a training pile-up)."""
    covs = sorted(len(v) for v in piles.values())
    cov = covs[len(covs) // 2]                                       # select_nth_unstable(len / 2)
    first = sorted(cid for cid, v in piles.items() if max(cov, 2) - 2 <= len(v) < cov + 2)[:5]
    return [cid for cid in first if cid in chunk_of]


def update_models_on_both_strands(ds, device=0):
    """ModelFit::update_models_on_both_strands (model_tune.rs:20-25, :96-156): pick the training pile-ups (:99-118) and
    refit the model on both strands with jtk_lc_fit_model (TRAIN_ROUND = 10)."""
    piles = {c["id"]: [] for c in ds["selected_chunks"]}
    chunk_of = {c["id"]: c for c in ds["selected_chunks"]}
    for read in ds["encoded_reads"]:
        for node in read["nodes"]:
            piles.setdefault(node["chunk"], []).append(node)
    if not piles:
        raise ValueError("update_models_on_both_strands: no pile-up")
    ids = training_pileup_ids(piles, chunk_of)
    pile = [(cid, int(chunk_of[cid]["copy_num"]), _seq(chunk_of[cid]["seq"]), [_seq(n["seq"]) for n in piles[cid]],
             [cigar_to_ops(n["cigar"]) for n in piles[cid]], [1 if n["is_forward"] else 0 for n in piles[cid]], None)
            for cid in ids]
    if not pile or not any(p[3] for p in pile):
        raise ValueError("update_models_on_both_strands: no training pile-up")     # assert!, model_tune.rs:135
    params = ffi.Params()
    params.forward = _hmm(ds["model_param"]["forward"])
    params.reverse = _hmm(ds["model_param"]["reverse"])
    params.band_frac = BAND_FRAC[ds["read_type"]]
    f, r = api.fit_model(params, pack(pile), rounds=10, device=device)
    for name, h in (("forward", f), ("reverse", r)):
        d = {k: float(getattr(h, k)) for k in ("mat_mat", "mat_ins", "mat_del", "ins_mat", "ins_ins", "ins_del", "del_mat",
                                               "del_ins", "del_del")}
        d["mat_emit"] = [float(v) for v in h.mat_emit]
        d["ins_emit"] = [float(v) for v in h.ins_emit]
        ds["model_param"][name] = d


def normalize_local_clustering(ds):
    """normalize.rs:6-51 over every node of the dataset: clusters relabelled by descending size, posteriors permuted."""
    L = ffi.lib()
    piles = {}
    for read in ds["encoded_reads"]:
        for node in read["nodes"]:
            piles.setdefault(node["chunk"], []).append(node)
    cluster_num = {c["id"]: c["cluster_num"] for c in ds["selected_chunks"]}
    for cid, nodes in piles.items():
        if cid not in cluster_num:
            raise ValueError("normalize_local_clustering: node on an unknown chunk")
        k = cluster_num[cid]
        if any(len(n["posterior"]) != k for n in nodes):   # assert_eq!, normalize.rs:27-29
            raise ValueError("normalize_local_clustering: posterior length != cluster_num")
        if k == 0:
            continue
        label = np.array([n["cluster"] for n in nodes], dtype=np.uint32)
        post = np.array([n["posterior"] for n in nodes], dtype=np.float64).reshape(len(nodes), k)
        ffi.check(L.jtk_lc_normalize_pileup(len(nodes), k, ffi.u32p(label), ffi.f64p(post), k))
        for i, n in enumerate(nodes):
            n["cluster"] = int(label[i])
            n["posterior"] = post[i].tolist()


def local_clustering_selected(ds, selection, gains=None, device=0, failed=None, refit=True, record=None, trace=None):
    """mod.rs:56-83 on the parsed JSON object `ds` (modified in place).

    The reference panics when a chunk hits one of its asserts; here such a chunk (and a chunk of a shape this build does
    not take) comes back with a status.  With `failed` = a list, those chunks are left exactly as they were, their
    (chunk id, status) pairs are appended to it and every other chunk is written back; with `failed` = None the call
    raises like the reference, before touching `ds`.  `record` = a list: the reference's per-chunk RECORD lines (mod.rs:121,
    `debug!`) of the chunks that were written back are appended to it (api.record_rows).  `trace` = a list: the reference's
    trace! rows of every clustered chunk of copy number < 8 (TOTAL / CAND / PICK / DUMP / RANGE / LK / COUNTS,
    pseudo_mcmc.rs:122-127,236,250-262,467-472,539), chunk after chunk in chunk-id order (api.Session.trace: those chunks once
    more through a resident session -- a debugging aid, like the log level it mirrors)."""
    validate(ds)
    update_coverage(ds)                                                       # mod.rs:57
    if refit:
        update_models_on_both_strands(ds, device=device)                      # mod.rs:58
    params = ffi.Params()
    params.forward = _hmm(ds["model_param"]["forward"])                       # mod.rs:59
    params.reverse = _hmm(ds["model_param"]["reverse"])
    params.gains = gains if gains is not None else api.estimate_gains(params.forward, params.reverse, device=device)
    params.haploid_coverage = coverage_of(ds)[1]
    params.band_frac = BAND_FRAC[ds["read_type"]]
    # pileup_nodes (mod.rs:33-53): nodes of the selected chunks in order of appearance, then a stable sort by the
    # number of non-'|' alignment columns against the unpolished chunk sequence
    selection = set(selection)
    chunk_of = {c["id"]: c for c in ds["selected_chunks"] if c["id"] in selection}
    piles = {cid: [] for cid in chunk_of}
    for read in ds["encoded_reads"]:
        for node in read["nodes"]:
            if node["chunk"] in piles:
                piles[node["chunk"]].append(node)
    order = sorted(cid for cid, nodes in piles.items() if nodes)
    pileups = []
    for cid in order:
        chunk, nodes = chunk_of[cid], piles[cid]
        tmpl = _seq(chunk["seq"])
        reads = [_seq(n["seq"]) for n in nodes]
        ops = [cigar_to_ops(n["cigar"]) for n in nodes]
        perm = pileup_sort(tmpl, reads, ops)
        piles[cid] = [nodes[i] for i in perm]
        pileups.append((cid, int(chunk["copy_num"]), tmpl, [reads[i] for i in perm], [ops[i] for i in perm],
                        [1 if nodes[i]["is_forward"] else 0 for i in perm], None))
    batch = pack(pileups)
    out = api.cluster_chunks(params, batch, device=device,                    # the hot loop of mod.rs:64-72
                             raise_on_chunk_failure=failed is None)
    # update_by_clusterings (mod.rs:244-260) and the chunk write-back (mod.rs:74-81)
    for c, cid in enumerate(order):
        if int(out["result"][c]["status"]) != 0:
            failed.append((cid, int(out["result"][c]["status"])))
            continue
        k = int(out["result"][c]["cluster_num"])
        for r, node in zip(batch.chunk_reads(c), piles[cid]):
            node["posterior"] = out["log_post"][r, :k].tolist()
            node["cluster"] = int(out["label"][r])
            node["cigar"] = ops_to_cigar(out["ops_out"][int(out["ops_out_off"][r]):int(out["ops_out_off"][r + 1])])
        chunk = chunk_of[cid]
        chunk["seq"] = bytes(out["cons"][int(out["cons_off"][c]):int(out["cons_off"][c + 1])]).decode("ascii")
        chunk["score"] = float(out["result"][c]["score"])
        chunk["cluster_num"] = k
    if record is not None:                                                    # mod.rs:121
        rows = api.record_rows(batch.chunks["chunk_id"], batch.chunks["n_reads"], batch.chunks["tmpl_len"],
                               [int(batch.read_off[batch.chunk_reads(c).stop] - batch.read_off[batch.chunk_reads(c).start])
                                for c in range(batch.n_chunks)], batch.chunks["copy_num"], out["result"],
                               np.diff(out["cons_off"]).astype(np.int64), api.last_timing())
        record.extend(r for c, r in enumerate(rows) if int(out["result"][c]["status"]) == 0)
    if trace is not None:
        keep = [c for c in range(batch.n_chunks) if int(out["result"][c]["status"]) == 0 and int(batch.chunks["copy_num"][c]) < 8]
        if keep:
            with api.Session(params, batch.subset(keep), device=device) as sess:
                sess.run()
                for i in range(len(keep)):
                    trace.extend(sess.trace(i))
    normalize_local_clustering(ds)                                            # mod.rs:82
    return ds


def local_clustering(ds, gains=None, device=0, failed=None, refit=True, record=None, trace=None):
    """mod.rs:23-26: every selected chunk."""
    validate(ds)
    return local_clustering_selected(ds, [c["id"] for c in ds["selected_chunks"]], gains=gains, device=device,
                                     failed=failed, refit=refit, record=record, trace=trace)


def correct_clustering_selected(ds, selection, device=0, min_gain=None):
    """AlignmentCorrection::correct_clustering_selected (phmm_likelihood_correction.rs:32-97) on the parsed JSON object `ds`
    (modified in place): jtk_lc_correct_clustering on the flattened nodes, then the write-back of :80-96.  min_gain defaults
    to estimate_minimum_gain(model) x PROTECT_FACTOR (:118) computed on the device."""
    tag, cov = coverage_of(ds)
    if cov is None:
        raise ValueError("correct_clustering: coverage is NotAvailable (the reference unwraps it, :154)")
    if min_gain is None:
        min_gain = api.estimate_minimum_gain(_hmm(ds["model_param"]["forward"]), _hmm(ds["model_param"]["reverse"]),
                                             device=device) * 1.0
    n_nodes = sum(len(r["nodes"]) for r in ds["encoded_reads"])
    nodes = np.zeros(n_nodes, dtype=ffi.CC_NODE_DT)
    node_off = np.zeros(len(ds["encoded_reads"]) + 1, dtype=np.uint64)
    read_id = np.zeros(len(ds["encoded_reads"]), dtype=np.uint64)
    post = []
    e = 0
    for r, read in enumerate(ds["encoded_reads"]):
        read_id[r] = read["id"]
        for node in read["nodes"]:
            nodes[e] = (node["chunk"], node["cluster"], 1 if node["is_forward"] else 0, len(node["posterior"]), len(post))
            post.extend(float(x) for x in node["posterior"])
            e += 1
        node_off[r + 1] = e
    chunks = np.zeros(len(ds["selected_chunks"]), dtype=ffi.CC_CHUNK_DT)
    for i, c in enumerate(ds["selected_chunks"]):
        chunks[i] = (c["id"], c["cluster_num"], c["copy_num"], c["score"])
    cluster, touched = api.correct_clustering(read_id, node_off, nodes, np.array(post, dtype=np.float64), chunks,
                                              sorted(int(x) for x in selection), cov, min_gain, device=device)
    cluster_num = {}
    for i, c in enumerate(ds["selected_chunks"]):
        c["cluster_num"] = int(chunks["cluster_num"][i])
        cluster_num[c["id"]] = c["cluster_num"]
    e = 0
    for read in ds["encoded_reads"]:
        for node in read["nodes"]:
            if touched[e]:                                           # :85-93
                node["cluster"] = int(cluster[e])
                node["posterior"] = [-10000.0] * cluster_num[node["chunk"]]
                node["posterior"][node["cluster"]] = 0.0
            e += 1


_COMPLEMENT = {"A": "T", "C": "G", "G": "C", "T": "A"}


def recover_raw_read(read):
    """EncodedRead::recover_raw_read (definitions/src/lib.rs:604-619): leading gap, then every node's original sequence
    (Node::original_seq :737-753: reverse complement of a reverse node; anything but ACGT panics) with the overlap of a
    negative edge offset popped and the edge label appended, the last node, the trailing gap."""
    def original(node):
        if node["is_forward"]:
            return node["seq"]
        try:
            return "".join(_COMPLEMENT[b] for b in reversed(node["seq"].upper()))
        except KeyError:
            raise ValueError("sanity_check: a reverse node holds a base outside ACGT (Node::original_seq panics)") from None
    out = list(read["leading_gap"])
    for node, edge in zip(read["nodes"], read["edges"]):
        if node["chunk"] != edge["from"]:
            raise ValueError(f"sanity_check: read {read['id']}: edge.from {edge['from']} after node of chunk {node['chunk']}")
        out.extend(original(node))
        drop = max(-int(edge["offset"]), 0)
        if drop:
            del out[max(len(out) - drop, 0):]
        out.extend(edge["label"])
    if read["nodes"]:
        out.extend(original(read["nodes"][-1]))
    out.extend(read["trailing_gap"])
    return "".join(out)


def sanity_check(ds):
    """DataSet::sanity_check (definitions/src/lib.rs:296-327 + encoded_reads_can_be_recovered :328-358), which
    correct_clustering ends with (phmm_likelihood_correction.rs:29).  Where the reference panics this raises ValueError."""
    ids = [c["id"] for c in ds["selected_chunks"]]
    chunks = set(ids)
    for read in ds["encoded_reads"]:
        for node in read["nodes"]:
            if node["chunk"] not in chunks:
                raise ValueError(f"sanity_check: read {read['id']} has a node on chunk {node['chunk']}, which is not selected")
    raw = {r["id"]: r["seq"] for r in ds["raw_reads"]}
    for read in ds["encoded_reads"]:
        if read["id"] not in raw:
            raise ValueError(f"sanity_check: encoded read {read['id']} has no raw read")
        orig, rec = raw[read["id"]].upper(), recover_raw_read(read).upper()
        if len(orig) != len(rec) or len(orig) != read["original_length"] or orig != rec:
            raise ValueError(f"sanity_check: encoded read {read['id']} does not recover its raw read "
                             f"({len(rec)} bases recovered, {len(orig)} raw, original_length {read['original_length']})")
    if len(chunks) != len(ids):
        raise ValueError("sanity_check: a chunk id occurs twice in selected_chunks")
    for c in ds["selected_chunks"]:
        if not c["cluster_num"] <= c["copy_num"]:
            raise ValueError(f"sanity_check: chunk {c['id']}: cluster_num {c['cluster_num']} > copy_num {c['copy_num']}")
    max_cl = {c["id"]: c["cluster_num"] for c in ds["selected_chunks"]}
    for read in ds["encoded_reads"]:
        for node in read["nodes"]:
            if not node["cluster"] <= max_cl[node["chunk"]]:
                raise ValueError(f"sanity_check: a node of chunk {node['chunk']} carries cluster {node['cluster']} of "
                                 f"{max_cl[node['chunk']]}")


def correct_clustering(ds, device=0, min_gain=None):
    """AlignmentCorrection::correct_clustering (phmm_likelihood_correction.rs:14-30): chunks no read visits are dropped,
    every chunk with more than one cluster is corrected, and the data set has to pass DataSet::sanity_check afterwards."""
    present = {n["chunk"] for r in ds["encoded_reads"] for n in r["nodes"]}
    ds["selected_chunks"] = [c for c in ds["selected_chunks"] if c["id"] in present]
    correct_clustering_selected(ds, [c["id"] for c in ds["selected_chunks"] if 1 < c["cluster_num"]], device=device,
                                min_gain=min_gain)
    sanity_check(ds)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--stage", default="local_clustering", choices=("local_clustering", "correct_clustering"),
                    help="which JTK stage to run on the file (jtk local_clustering / jtk correct_clustering)")
    ap.add_argument("input", help="DataSet JSON ('-' = stdin)")
    ap.add_argument("output", help="DataSet JSON ('-' = stdout)")
    ap.add_argument("--chunks", default="", help="comma-separated chunk ids (local_clustering_selected); default: all")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--no-refit", action="store_true",
                    help="skip update_models_on_both_strands (mod.rs:58): cluster with model_param as found in the file")
    ap.add_argument("--keep-going", action="store_true",
                    help="a chunk that fails (where the reference would panic, or an unsupported shape) is left untouched "
                         "and listed on stderr instead of aborting the stage")
    ap.add_argument("-v", "--verbose", action="store_true",
                    help="write the reference's per-chunk RECORD lines (mod.rs:121, its debug! level) to stderr")
    ap.add_argument("--trace", action="store_true",
                    help="write the reference's trace! rows of every clustered chunk (TOTAL / CAND / PICK / DUMP / RANGE / LK / "
                         "COUNTS; its -vvv) to stderr")
    args = ap.parse_args(argv)
    ds = json.load(sys.stdin if args.input == "-" else open(args.input))
    failed = [] if args.keep_going else None
    record = [] if args.verbose else None
    trace = [] if args.trace else None
    if args.stage == "correct_clustering":
        validate(ds)
        if args.chunks:
            correct_clustering_selected(ds, [int(x) for x in args.chunks.split(",")], device=args.device)
        else:
            correct_clustering(ds, device=args.device)
    elif args.chunks:
        local_clustering_selected(ds, [int(x) for x in args.chunks.split(",")], device=args.device, failed=failed,
                                  refit=not args.no_refit, record=record, trace=trace)
    else:
        local_clustering(ds, device=args.device, failed=failed, refit=not args.no_refit, record=record, trace=trace)
    for row in (trace or []) + (record or []):
        sys.stderr.write(row + "\n")
    for cid, status in failed or []:
        sys.stderr.write(f"LC\tFAILED\t{cid}\t{status}\t{ffi.lib().jtk_lc_strerror(status).decode()}\n")
    api.trim_cache(args.device)
    text = json.dumps(ds)
    if args.output == "-":
        sys.stdout.write(text + "\n")
    else:
        with open(args.output, "w") as f:
            f.write(text + "\n")
    return 3 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
