// haplotyper/src/local_clustering/gpu_shim.rs -- replaces the rayon loop of local_clustering_selected (mod.rs:63-72)
// NOT compiled in this repository: the build image has no Rust toolchain (cargo, rustc: command not found) and the
// reference's git dependencies are un-vendored.  Source a jtk maintainer adds to ban-m/jtk; INTEGRATION.md explains it and
// tests/test_rust_shim_source.py keeps it in step with include/jtk_lc.h.  The same call sequence is exercised end to end
// by the C++ host mirror (jtk_amd/csrc/host/local_clustering.hpp) and the Python harness (jtk_amd/api.py).
use super::gpu_ffi::*;
use std::collections::HashMap;

/// `pileups`: the map `pileup_nodes` returns (mod.rs:33-53); `hmm`, `gains`, `coverage`, `read_type` as computed at mod.rs:57-62.
/// Returns what the rayon loop collects: chunk id -> (consensus, score, cluster_num)  (mod.rs:64-72).
pub(crate) fn clustering_on_pileups_gpu(
    pileups: &mut HashMap<u64, (Vec<&mut definitions::Node>, &definitions::Chunk)>,
    read_type: definitions::ReadType,
    hmm: &kiley::hmm::PairHiddenMarkovModelOnStrands,
    gains: &crate::likelihood_gains::Gains,
    coverage: f64,
) -> HashMap<u64, (Vec<u8>, f64, usize)> {
    let mut order: Vec<u64> = pileups.iter().filter(|(_, (n, _))| !n.is_empty()).map(|(&id, _)| id).collect();
    order.sort_unstable();                                          // any fixed order; results do not depend on it
    let (mut chunks, mut tmpl, mut reads, mut read_off) = (vec![], vec![], vec![], vec![0u64]);
    let (mut ops, mut ops_off, mut strand) = (vec![], vec![0u64], vec![]);
    for id in order.iter() {
        let (nodes, chunk) = &pileups[id];
        chunks.push(JtkLcChunk { chunk_id: chunk.id, copy_num: chunk.copy_num as u32, n_reads: nodes.len() as u32,
            tmpl_off: tmpl.len() as u64, tmpl_len: chunk.seq().len() as u64, read_first: strand.len() as u64 });
        tmpl.extend_from_slice(chunk.seq());
        for node in nodes.iter() {
            reads.extend_from_slice(node.seq());
            read_off.push(reads.len() as u64);
            // kiley::Op order of misc.rs:167-172: 0 Match, 1 Mismatch, 2 Ins, 3 Del  (ops_to_kiley, misc.rs:177-186)
            ops.extend(crate::misc::ops_to_kiley(&node.cigar).iter().map(|op| match op {
                kiley::Op::Match => 0u8, kiley::Op::Mismatch => 1, kiley::Op::Ins => 2, kiley::Op::Del => 3 }));
            ops_off.push(ops.len() as u64);
            strand.push(node.is_forward as u8);
        }
    }
    let params = JtkLcParams { forward: to_ffi(hmm.forward()), reverse: to_ffi(hmm.reverse()), gains: gains.to_ffi(),
        haploid_coverage: coverage, band_frac: read_type.band_width(1_000_000) as f64 / 1_000_000f64 };
    let stride = chunks.iter().map(|c| c.copy_num).max().unwrap_or(1).max(1);
    let n = strand.len();
    let (mut label, mut post) = (vec![0u32; n], vec![0f64; n * stride as usize]);
    let mut result = vec![JtkLcResult::default(); chunks.len()];
    let (cons_cap, ops_cap) = (2 * tmpl.len() + 64 * chunks.len() + 64, 2 * ops.len() + 64 * n + 64);
    let (mut cons, mut cons_off) = (vec![0u8; cons_cap], vec![0u64; chunks.len() + 1]);
    let (mut ops_out, mut ops_out_off) = (vec![0u8; ops_cap], vec![0u64; n + 1]);
    let rc = unsafe { jtk_lc_cluster_chunks(&params, chunks.len(), chunks.as_ptr(), tmpl.as_ptr(), reads.as_ptr(),
        read_off.as_ptr(), ops.as_ptr(), ops_off.as_ptr(), strand.as_ptr(), label.as_mut_ptr(), post.as_mut_ptr(),
        stride, result.as_mut_ptr(), cons.as_mut_ptr(), cons_off.as_mut_ptr(), cons_cap as u64,
        ops_out.as_mut_ptr(), ops_out_off.as_mut_ptr(), ops_cap as u64, /*device*/ 0) };
    // rc == -6 (JTK_ERR_CHUNK_FAILED): result[c].status names the chunks that hit a condition on which the reference itself
    // panics (e.g. misc.rs:335, pseudo_mcmc.rs:759) or a shape this build does not take; the others are complete.
    assert!(rc == 0 || rc == -6, "{}", unsafe { std::ffi::CStr::from_ptr(jtk_lc_last_error()) }.to_string_lossy());
    // RUST_LOG=trace: the rows pseudo_mcmc.rs logs per chunk (TOTAL / CAND / PICK / DUMP / RANGE / LK / COUNTS) come from the
    // device -- the batch once more as a resident session, jtk_lc_session_trace per clustered chunk (a debugging aid, like the
    // log level; a batch that holds a chunk of copy number >= 8 gets no rows: JTK_ERR_UNSUPPORTED)
    if log_enabled!(log::Level::Trace) {
        let mut sess: *mut std::os::raw::c_void = std::ptr::null_mut();
        if unsafe { jtk_lc_session_create(&params, chunks.len(), chunks.as_ptr(), tmpl.as_ptr(), reads.as_ptr(), read_off.as_ptr(),
            ops.as_ptr(), ops_off.as_ptr(), strand.as_ptr(), stride, 0, &mut sess) } == 0 {
            let run_rc = unsafe { jtk_lc_session_run(sess, 0) };
            let mut text = vec![0u8; 1 << 16];
            for c in (0..chunks.len()).filter(|&c| (run_rc == 0 || run_rc == -6) && result[c].status == 0) {
                let mut len = 0usize;
                let mut trc = unsafe { jtk_lc_session_trace(sess, c, text.as_mut_ptr() as *mut _, text.len(), &mut len) };
                if trc != 0 && len > text.len() {
                    text.resize(len, 0);
                    trc = unsafe { jtk_lc_session_trace(sess, c, text.as_mut_ptr() as *mut _, text.len(), &mut len) };
                }
                if trc != 0 { break; }
                for row in String::from_utf8_lossy(&text[..len]).lines() { trace!("{row}"); }
            }
            unsafe { jtk_lc_session_destroy(sess) };
        }
    }
    // update_by_clusterings (mod.rs:244-260) + the tuple clustering_on_pileup returns (mod.rs:122)
    let mut consensus_and_clusternum = HashMap::new();
    for (c, id) in order.iter().enumerate() {
        if result[c].status != 0 {                                  // left untouched, as if not selected
            warn!("LC\tFAILED\t{id}\t{}", result[c].status);
            continue;
        }
        let (nodes, _) = pileups.get_mut(id).unwrap();
        let k = result[c].cluster_num as usize;
        for (r, node) in nodes.iter_mut().enumerate() {
            let g = chunks[c].read_first as usize + r;
            node.cluster = label[g] as u64;
            node.posterior.clear();
            node.posterior.extend_from_slice(&post[g * stride as usize..g * stride as usize + k]);
            let k_ops: Vec<kiley::Op> = ops_out[ops_out_off[g] as usize..ops_out_off[g + 1] as usize].iter()
                .map(|&b| [kiley::Op::Match, kiley::Op::Mismatch, kiley::Op::Ins, kiley::Op::Del][b as usize]).collect();
            node.cigar = crate::misc::kiley_op_to_ops(&k_ops);
        }
        let cons_c = cons[cons_off[c] as usize..cons_off[c + 1] as usize].to_vec();
        consensus_and_clusternum.insert(*id, (cons_c, result[c].score, k));
    }
    consensus_and_clusternum  // mod.rs:73 on is unchanged (chunk.seq/score/cluster_num write-back, normalize_local_clustering)
}

fn to_ffi(h: &kiley::hmm::PairHiddenMarkovModel) -> JtkHmm {
    let d = crate::model_tune::kiley_into_def(h);                  // model_tune.rs:65-92
    JtkHmm { mat_mat: d.mat_mat, mat_ins: d.mat_ins, mat_del: d.mat_del, ins_mat: d.ins_mat, ins_ins: d.ins_ins,
             ins_del: d.ins_del, del_mat: d.del_mat, del_ins: d.del_ins, del_del: d.del_del,
             mat_emit: d.mat_emit, ins_emit: d.ins_emit }
}
