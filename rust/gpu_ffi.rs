// haplotyper/src/local_clustering/gpu_ffi.rs -- field-for-field #[repr(C)] mirrors of include/jtk_lc.h + the extern block
// NOT compiled in this repository: the build image has no Rust toolchain (cargo, rustc: command not found) and the
// reference's git dependencies are un-vendored.  Source a jtk maintainer adds to ban-m/jtk; INTEGRATION.md explains it and
// tests/test_rust_shim_source.py keeps it in step with include/jtk_lc.h.  The same call sequence is exercised end to end
// by the C++ host mirror (jtk_amd/csrc/host/local_clustering.hpp) and the Python harness (jtk_amd/api.py).
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] #[derive(Clone, Copy)]
pub struct JtkHmm {                       // == definitions::HMMParam (definitions/src/lib.rs:101-126)
    pub mat_mat: f64, pub mat_ins: f64, pub mat_del: f64,
    pub ins_mat: f64, pub ins_ins: f64, pub ins_del: f64,
    pub del_mat: f64, pub del_ins: f64, pub del_del: f64,
    pub mat_emit: [f64; 16], pub ins_emit: [f64; 20],
}
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct JtkGainProfile { pub gain: f64, pub prob: f64 }     // likelihood_gains.rs:41-47
#[repr(C)] #[derive(Clone, Copy)]
pub struct JtkGains {                                           // likelihood_gains.rs:55-61
    pub max_homopolymer_len: u32, pub reserved: u32,
    pub subst: [JtkGainProfile; 8], pub deletions: [JtkGainProfile; 8], pub insertions: [JtkGainProfile; 8],
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct JtkLcParams {
    pub forward: JtkHmm, pub reverse: JtkHmm, pub gains: JtkGains,
    pub haploid_coverage: f64, pub band_frac: f64,
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct JtkLcChunk {
    pub chunk_id: u64, pub copy_num: u32, pub n_reads: u32,
    pub tmpl_off: u64, pub tmpl_len: u64, pub read_first: u64,
}
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct JtkLcResult {
    pub score: f64, pub cluster_num: u32, pub status: i32, pub polish_rounds: u32, pub n_variants: u32,
}

// the flattened DataSet view of jtk_lc_correct_clustering (phmm_likelihood_correction.rs:32-97)
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct JtkCcNode {
    pub chunk: u64, pub cluster: u64, pub is_forward: u32, pub post_len: u32, pub post_off: u64,
}
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct JtkCcChunk {
    pub id: u64, pub cluster_num: u32, pub copy_num: u32, pub score: f64,
}

extern "C" {
    pub fn jtk_lc_cluster_chunks(
        params: *const JtkLcParams, n_chunks: usize, chunks: *const JtkLcChunk,
        tmpl_bases: *const u8, read_bases: *const u8, read_off: *const u64,
        ops: *const u8, ops_off: *const u64, strand: *const u8,
        label: *mut u32, log_post: *mut f64, post_stride: u32, result: *mut JtkLcResult,
        cons_out: *mut u8, cons_off: *mut u64, cons_cap: u64,
        ops_out: *mut u8, ops_out_off: *mut u64, ops_cap: u64, device: c_int) -> c_int;
    // the same over several GPUs of one node (contiguous, read-balanced shares; no collective)
    pub fn jtk_lc_cluster_chunks_multi(
        params: *const JtkLcParams, n_chunks: usize, chunks: *const JtkLcChunk,
        tmpl_bases: *const u8, read_bases: *const u8, read_off: *const u64,
        ops: *const u8, ops_off: *const u64, strand: *const u8,
        label: *mut u32, log_post: *mut f64, post_stride: u32, result: *mut JtkLcResult,
        cons_out: *mut u8, cons_off: *mut u64, cons_cap: u64,
        ops_out: *mut u8, ops_out_off: *mut u64, ops_cap: u64, devices: *const c_int, n_devices: usize) -> c_int;
    pub fn jtk_lc_estimate_gains(forward: *const JtkHmm, reverse: *const JtkHmm, seed: u64, seq_len: u32, band: u32,
                                 homop_len: u32, out: *mut JtkGains, device: c_int) -> c_int;   // likelihood_gains.rs:162-192
    pub fn jtk_lc_estimate_minimum_gain(forward: *const JtkHmm, reverse: *const JtkHmm, seed: u64, sample_num: u32,
                                        seq_num: u32, len: u32, band: u32, out: *mut f64, device: c_int) -> c_int;  // likelihood_gains.rs:6-39
    // model_tune.rs:119-152 on the training pile-ups the host selected (model_tune.rs:99-118)
    pub fn jtk_lc_fit_model(
        params: *const JtkLcParams, n_chunks: usize, chunks: *const JtkLcChunk,
        tmpl_bases: *const u8, read_bases: *const u8, read_off: *const u64,
        ops: *const u8, ops_off: *const u64, strand: *const u8, rounds: u32,
        forward_out: *mut JtkHmm, reverse_out: *mut JtkHmm, device: c_int) -> c_int;
    // kiley polish_until_converge_antidiagonal on a batch of windows (consensus/mod.rs:476-483)
    pub fn jtk_lc_polish_chunks(
        params: *const JtkLcParams, n_chunks: usize, chunks: *const JtkLcChunk,
        tmpl_bases: *const u8, read_bases: *const u8, read_off: *const u64,
        ops: *const u8, ops_off: *const u64, strand: *const u8, radius: u32, take_num: u32, ignore_edge: u32,
        cons_out: *mut u8, cons_off: *mut u64, cons_cap: u64,
        ops_out: *mut u8, ops_out_off: *mut u64, ops_cap: u64, result: *mut JtkLcResult, device: c_int) -> c_int;
    // AlignmentCorrection::correct_clustering_selected (phmm_likelihood_correction.rs:32-97)
    pub fn jtk_lc_correct_clustering(
        n_reads: usize, read_id: *const u64, node_off: *const u64, nodes: *const JtkCcNode, posteriors: *const f64,
        n_chunks: usize, chunks: *mut JtkCcChunk, n_selected: usize, selection: *const u64,
        haploid_coverage: f64, min_gain: f64, cluster_out: *mut u64, touched: *mut u8, device: c_int) -> c_int;
    // the resident-batch form (jtk_lc.h: session_create + run + fetch == jtk_lc_cluster_chunks) and, on it, the reference's
    // trace! rows of one chunk (TOTAL / CAND / PICK / DUMP / RANGE / LK / COUNTS; pseudo_mcmc.rs:122-127,236,250-262,467-472,539)
    pub fn jtk_lc_session_create(
        params: *const JtkLcParams, n_chunks: usize, chunks: *const JtkLcChunk,
        tmpl_bases: *const u8, read_bases: *const u8, read_off: *const u64,
        ops: *const u8, ops_off: *const u64, strand: *const u8, post_stride: u32, device: c_int,
        out: *mut *mut c_void) -> c_int;
    pub fn jtk_lc_session_run(s: *mut c_void, skip_polish: c_int) -> c_int;
    pub fn jtk_lc_session_trace(s: *mut c_void, chunk: usize, text: *mut c_char, cap: usize, len: *mut usize) -> c_int;
    pub fn jtk_lc_session_destroy(s: *mut c_void) -> c_int;
    pub fn jtk_lc_trim_cache(device: c_int) -> c_int;          // hand pooled device workspaces back to the driver
    pub fn jtk_lc_strerror(status: c_int) -> *const c_char;
    pub fn jtk_lc_last_error() -> *const c_char;
}
