// UNVERIFIED SOURCE: no Rust toolchain exists in the image this backend was built in (no rustc / cargo), so this file has
// never been compiled. It is the module a maintainer drops into block-aligner's src/ next to avx2.rs
// (/root/reference/src/lib.rs:55-103 selects the SIMD layer by cargo feature); every declaration mirrors
// include/block_aligner_hip.h, which IS compiled and tested (tests/test_c_abi.py builds a C caller against it).
#![cfg(feature = "simd_hip")]
#![allow(non_snake_case)]

// src/hip.rs — replaces avx2.rs + the align_core / place_block bodies of scan_block.rs when `simd_hip` is enabled
use std::os::raw::c_void;
use crate::scores::Gaps;

#[repr(C)] #[derive(Copy, Clone)] pub struct SizeRange { pub min: usize, pub max: usize }
#[repr(C)] #[derive(Copy, Clone)] pub struct AlignResult { pub score: i32, pub query_idx: usize, pub reference_idx: usize }

pub const BA_TRACE: u32 = 1; pub const BA_X_DROP: u32 = 2; pub const BA_LOCAL_START: u32 = 4;
pub const BA_FREE_QUERY_START_GAPS: u32 = 8; pub const BA_FREE_QUERY_END_GAPS: u32 = 16; pub const BA_CIGAR_EQ: u32 = 32;
pub const BA_KIND_AA: i32 = 0; pub const BA_KIND_NUC: i32 = 1; pub const BA_KIND_BYTES: i32 = 2;

#[link(name = "block_aligner_hip")]
extern "C" {
    // per-pair handle: Block::<mode>::new / align::<M> / res / trace().cigar[_eq]   (scan_block.rs:798-878,1235-1244,1469-1480)
    pub fn block_new_generic(mode: u32, query_len: usize, reference_len: usize, max_size: usize) -> *mut c_void;
    pub fn block_align_generic(b: *mut c_void, kind: i32, q: *const c_void, r: *const c_void, matrix: *const c_void,
                               g: Gaps, s: SizeRange, x: i32);
    pub fn block_align_profile_generic(b: *mut c_void, q: *const c_void, profile: *const c_void, s: SizeRange, x: i32);   // scan_block.rs:942-968
    pub fn block_res_generic(b: *mut c_void) -> AlignResult;
    pub fn block_cigar_generic(b: *mut c_void, query_idx: usize, reference_idx: usize, cigar: *mut c_void);
    pub fn block_cigar_eq_generic(b: *mut c_void, q: *const c_void, r: *const c_void, query_idx: usize, reference_idx: usize, cigar: *mut c_void);
    pub fn block_free_generic(b: *mut c_void);
    // batch launcher: one persistent kernel launch, one wavefront per pair
    pub fn ba_batch_create(kind: i32, matrix: *const c_void, gaps: Gaps, size: SizeRange, x_drop: i32, mode: u32,
                           pool: *const u8, q_off: *const u64, q_len: *const u32, r_off: *const u64, r_len: *const u32,
                           n_pairs: usize) -> *mut c_void;
    pub fn ba_batch_create_profile(profiles: *const *const c_void, size: SizeRange, x_drop: i32, mode: u32,
                                   pool: *const u8, q_off: *const u64, q_len: *const u32, n_pairs: usize) -> *mut c_void;
    pub fn ba_batch_reload(batch: *mut c_void, pool: *const u8, q_off: *const u64, q_len: *const u32, r_off: *const u64,
                           r_len: *const u32, n_pairs: usize) -> i32;   // new pairs, same device buffers
    pub fn ba_batch_run(batch: *mut c_void, kernel_ms: *mut f32) -> i32;
    pub fn ba_batch_results(batch: *mut c_void, score: *mut i32, query_idx: *mut u32, reference_idx: *mut u32,
                            cells: *mut u64, cigar_len: *mut u32, status: *mut u32) -> i32;
    pub fn ba_batch_cigars(batch: *mut c_void, runs: *mut u32, capacity: u64) -> i32;
    pub fn ba_batch_destroy(batch: *mut c_void);
    // Block::align_exp over a batch (scan_block.rs:884-902): reached_min[p] = 0 where the crate returns None
    pub fn block_batch_align_exp(kind: i32, matrix: *const c_void, gaps: Gaps, size: SizeRange, x_drop: i32, target_score: i32, mode: u32,
                                 pool: *const u8, q_off: *const u64, q_len: *const u32, r_off: *const u64, r_len: *const u32,
                                 n_pairs: usize, results: *mut AlignResult, reached_min: *mut usize) -> i32;
    // Trace::blocks() (scan_block.rs:1676-1691)
    pub fn block_trace_blocks_generic(b: *mut c_void, out: *mut Rectangle, capacity: usize) -> usize;
    pub fn ba_last_error() -> *const std::os::raw::c_char;
}

#[repr(C)] #[derive(Copy, Clone)] pub struct Rectangle { pub row: usize, pub col: usize, pub width: usize, pub height: usize }

#[link(name = "block_aligner_hip")]
extern "C" {
    // one batch over several GPUs of a node: cost-balanced contiguous slices, results in the caller's order
    pub fn ba_multibatch_create(kind: i32, matrix: *const c_void, gaps: Gaps, size: SizeRange, x_drop: i32, mode: u32,
                                pool: *const u8, q_off: *const u64, q_len: *const u32, r_off: *const u64, r_len: *const u32,
                                n_pairs: usize, devices: *const i32, n_devices: i32) -> *mut c_void;
    pub fn ba_multibatch_run(batch: *mut c_void, kernel_ms: *mut f32) -> i32;
    pub fn ba_multibatch_results(batch: *mut c_void, score: *mut i32, query_idx: *mut u32, reference_idx: *mut u32,
                                 cells: *mut u64, cigar_len: *mut u32, status: *mut u32) -> i32;
    pub fn ba_multibatch_cigars(batch: *mut c_void, runs: *mut u32, capacity: u64) -> i32;
    pub fn ba_multibatch_destroy(batch: *mut c_void);
    pub fn ba_batch_surviving_cells(batch: *mut c_void, cells: *mut u64) -> i32;
    pub fn ba_device_count() -> i32;
    pub fn ba_set_device(device: i32) -> i32;      // per calling thread
    pub fn block_percent_len(len: usize, p: f32) -> usize;
}
