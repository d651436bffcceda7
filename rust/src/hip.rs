// UNCOMPILED SOURCE: no Rust toolchain exists in the image this backend was built in (no rustc / cargo). It is the module a maintainer
// drops into block-aligner's src/ next to avx2.rs (the patches in this directory add the `simd_hip` feature and the module gates:
// tests/test_rust_patches.py applies them to a copy of the reference). Every declaration mirrors include/block_aligner_hip.h, which
// IS compiled and tested (tests/test_c_abi.py builds a C caller against it; tests/test_abi.py checks every declared symbol is exported).
//
// src/hip.rs -- raw FFI of libblock_aligner_hip.so + the batch API (HipBatch), the route the device is built for: one launch over
// many pairs instead of one launch per pair. The matrix kind and bytes come from Matrix::HIP_KIND / Matrix::hip_ptr (scores.rs.patch).
#![allow(non_snake_case)]

use std::os::raw::{c_char, c_void};
use std::ffi::CStr;
use crate::scores::{Gaps, Matrix};
use crate::cigar::{OpLen, Operation};
use crate::scan_block::AlignResult;   // #[repr(C)] in the crate (scan_block.rs:1887-1893): crosses the ABI as it is

#[repr(C)] #[derive(Copy, Clone)] pub struct SizeRange { pub min: usize, pub max: usize }
#[repr(C)] #[derive(Copy, Clone)] pub struct CRectangle { pub row: usize, pub col: usize, pub width: usize, pub height: usize }   // (the crate's Rectangle is not repr(C))
#[repr(C)] #[derive(Copy, Clone)] pub struct COpLen { pub op: u8, pub len: usize }   // c/block_aligner.h:17-66 (enum Operation is one byte)

pub const BA_TRACE: u32 = 1; pub const BA_X_DROP: u32 = 2; pub const BA_LOCAL_START: u32 = 4;
pub const BA_FREE_QUERY_START_GAPS: u32 = 8; pub const BA_FREE_QUERY_END_GAPS: u32 = 16; pub const BA_CIGAR_EQ: u32 = 32;
pub const BA_KIND_AA: i32 = 0; pub const BA_KIND_NUC: i32 = 1; pub const BA_KIND_BYTES: i32 = 2;

/// The matrix as the C ABI takes it. AAMatrix / NucMatrix are #[repr(C)] byte tables (scores.rs:41-46,139-144): their own memory;
/// ByteMatrix (two private i8 fields, no repr) is passed as {match, mismatch} read through Matrix::get.
pub(crate) struct MatrixArg { bytes: [i8; 2], ptr: *const c_void }
impl MatrixArg {
    pub(crate) fn of<M: Matrix>(m: &M) -> Self {
        if M::HIP_KIND == BA_KIND_BYTES { MatrixArg { bytes: [m.get(0, 0), m.get(0, 1)], ptr: std::ptr::null() } } else { MatrixArg { bytes: [0, 0], ptr: m.hip_ptr() } }
    }
    pub(crate) fn ptr(&self) -> *const c_void { if self.ptr.is_null() { self.bytes.as_ptr() as *const c_void } else { self.ptr } }
}

#[link(name = "block_aligner_hip")]
extern "C" {
    // ---- per-pair handle: Block::<mode>::new / align / align_profile / res / trace().cigar[_eq] / trace().blocks()
    //      (scan_block.rs:798-878, 942-968, 1235-1244, 1469-1480, 1676-1691)
    pub fn block_new_generic(mode: u32, query_len: usize, reference_len: usize, max_size: usize) -> *mut c_void;
    pub fn block_align_padded_generic(b: *mut c_void, kind: i32, q_s: *const u8, q_len: usize, r_s: *const u8, r_len: usize,
                                      matrix: *const c_void, g: Gaps, s: SizeRange, x: i32);
    pub fn block_align_profile_padded_generic(b: *mut c_void, q_s: *const u8, q_len: usize, profile: *const c_void, s: SizeRange, x: i32);
    pub fn block_res_generic(b: *mut c_void) -> AlignResult;
    pub fn block_cigar_generic(b: *mut c_void, query_idx: usize, reference_idx: usize, cigar: *mut c_void);
    pub fn block_cigar_eq_generic(b: *mut c_void, q: *const c_void, r: *const c_void, query_idx: usize, reference_idx: usize, cigar: *mut c_void);
    pub fn block_trace_blocks_generic(b: *mut c_void, out: *mut CRectangle, capacity: usize) -> usize;
    pub fn block_free_generic(b: *mut c_void);
    // ---- library-side Cigar (cigar.rs:42-95 as c/block_aligner.h exposes it)
    pub fn block_new_cigar(query_len: usize, reference_len: usize) -> *mut c_void;
    pub fn block_len_cigar(c: *const c_void) -> usize;
    pub fn block_get_cigar(c: *const c_void, i: usize) -> COpLen;
    pub fn block_free_cigar(c: *mut c_void);
    // ---- library-side AAProfile (scores.rs:452-715 as c/block_aligner.h:79-120 exposes it)
    pub fn block_new_aaprofile(str_len: usize, block_size: usize, gap_extend: i8) -> *mut c_void;
    pub fn block_len_aaprofile(p: *const c_void) -> usize;
    pub fn block_clear_aaprofile(p: *mut c_void, str_len: usize, block_size: usize);
    pub fn block_set_aaprofile(p: *mut c_void, i: usize, b: u8, score: i8);
    pub fn block_set_all_aaprofile(p: *mut c_void, order: *const u8, order_len: usize, scores: *const i8, scores_len: usize, left_shift: usize, right_shift: usize);
    pub fn block_set_all_rev_aaprofile(p: *mut c_void, order: *const u8, order_len: usize, scores: *const i8, scores_len: usize, left_shift: usize, right_shift: usize);
    pub fn block_set_gap_open_C_aaprofile(p: *mut c_void, i: usize, gap: i8);
    pub fn block_set_gap_close_C_aaprofile(p: *mut c_void, i: usize, gap: i8);
    pub fn block_set_gap_open_R_aaprofile(p: *mut c_void, i: usize, gap: i8);
    pub fn block_set_all_gap_open_C_aaprofile(p: *mut c_void, gap: i8);
    pub fn block_set_all_gap_close_C_aaprofile(p: *mut c_void, gap: i8);
    pub fn block_set_all_gap_open_R_aaprofile(p: *mut c_void, gap: i8);
    pub fn block_get_aaprofile(p: *const c_void, i: usize, b: u8) -> i8;
    pub fn block_get_gap_extend_aaprofile(p: *const c_void) -> i8;
    pub fn block_free_aaprofile(p: *mut c_void);
    pub fn ba_aaprofile_set_raw(p: *mut c_void, pos_aa: *const i8, gap_open_C: *const i8, gap_close_C: *const i8, gap_open_R: *const i8, positions: usize) -> i32;
    // ---- batch launcher: one persistent launch over many pairs (the default route, see HipBatch)
    pub fn ba_batch_create(kind: i32, matrix: *const c_void, gaps: Gaps, size: SizeRange, x_drop: i32, mode: u32,
                           pool: *const u8, q_off: *const u64, q_len: *const u32, r_off: *const u64, r_len: *const u32, n_pairs: usize) -> *mut c_void;
    pub fn ba_batch_create_profile(profiles: *const *const c_void, size: SizeRange, x_drop: i32, mode: u32,
                                   pool: *const u8, q_off: *const u64, q_len: *const u32, n_pairs: usize) -> *mut c_void;
    pub fn ba_batch_reload(batch: *mut c_void, pool: *const u8, q_off: *const u64, q_len: *const u32, r_off: *const u64, r_len: *const u32, n_pairs: usize) -> i32;
    pub fn ba_batch_run(batch: *mut c_void, kernel_ms: *mut f32) -> i32;
    pub fn ba_batch_launch(batch: *mut c_void) -> i32;
    pub fn ba_batch_wait(batch: *mut c_void, kernel_ms: *mut f32) -> i32;
    pub fn ba_batch_results(batch: *mut c_void, score: *mut i32, query_idx: *mut u32, reference_idx: *mut u32, cells: *mut u64, cigar_len: *mut u32, status: *mut u32) -> i32;
    pub fn ba_batch_cigars(batch: *mut c_void, runs: *mut u32, capacity: u64) -> i32;
    pub fn ba_batch_compact_cigars(batch: *mut c_void, pinned_out: *mut u32, pinned_capacity: u64) -> i32;
    pub fn ba_host_alloc(bytes: u64) -> *mut c_void;
    pub fn ba_host_free(p: *mut c_void);
    pub fn ba_batch_surviving_cells(batch: *mut c_void, cells: *mut u64) -> i32;
    pub fn ba_batch_destroy(batch: *mut c_void);
    // Block::align_exp / align_profile_exp over a batch (scan_block.rs:884-902, 974-992): reached_min[p] = 0 where the crate returns None
    pub fn block_batch_align_exp(kind: i32, matrix: *const c_void, gaps: Gaps, size: SizeRange, x_drop: i32, target_score: i32, mode: u32,
                                 pool: *const u8, q_off: *const u64, q_len: *const u32, r_off: *const u64, r_len: *const u32,
                                 n_pairs: usize, results: *mut AlignResult, reached_min: *mut usize) -> i32;
    pub fn block_batch_align_profile_exp(profiles: *const *const c_void, size: SizeRange, x_drop: i32, target_score: i32, mode: u32,
                                         pool: *const u8, q_off: *const u64, q_len: *const u32, n_pairs: usize, results: *mut AlignResult, reached_min: *mut usize) -> i32;
    // one batch over several GPUs of a node (cost-balanced contiguous slices, results in the caller's order)
    pub fn ba_multibatch_create(kind: i32, matrix: *const c_void, gaps: Gaps, size: SizeRange, x_drop: i32, mode: u32,
                                pool: *const u8, q_off: *const u64, q_len: *const u32, r_off: *const u64, r_len: *const u32,
                                n_pairs: usize, devices: *const i32, n_devices: i32) -> *mut c_void;
    pub fn ba_multibatch_run(batch: *mut c_void, kernel_ms: *mut f32) -> i32;
    pub fn ba_multibatch_results(batch: *mut c_void, score: *mut i32, query_idx: *mut u32, reference_idx: *mut u32, cells: *mut u64, cigar_len: *mut u32, status: *mut u32) -> i32;
    pub fn ba_multibatch_cigars(batch: *mut c_void, runs: *mut u32, capacity: u64) -> i32;
    pub fn ba_multibatch_destroy(batch: *mut c_void);
    pub fn ba_device_count() -> i32;
    pub fn ba_set_device(device: i32) -> i32;      // per calling thread
    // every pair with its own block range (examples/nanopore_bench_global.rs:144-171: percent_len per pair); results in the caller's order
    pub fn ba_sized_batch_create(kind: i32, matrix: *const c_void, gaps: Gaps, size_per_pair: *const SizeRange, x_drop: i32, mode: u32,
                                 pool: *const u8, q_off: *const u64, q_len: *const u32, r_off: *const u64, r_len: *const u32, n_pairs: usize) -> *mut c_void;
    pub fn ba_sized_batch_create_percent(kind: i32, matrix: *const c_void, gaps: Gaps, min_percent: f32, max_percent: f32, x_drop: i32, mode: u32,
                                         pool: *const u8, q_off: *const u64, q_len: *const u32, r_off: *const u64, r_len: *const u32, n_pairs: usize) -> *mut c_void;
    pub fn ba_sized_batch_run(batch: *mut c_void, kernel_ms: *mut f32) -> i32;
    pub fn ba_sized_batch_results(batch: *mut c_void, score: *mut i32, query_idx: *mut u32, reference_idx: *mut u32, cells: *mut u64, cigar_len: *mut u32, status: *mut u32) -> i32;
    pub fn ba_sized_batch_cigars(batch: *mut c_void, runs: *mut u32, capacity: u64) -> i32;
    pub fn ba_sized_batch_destroy(batch: *mut c_void);
    // upper bound of a ba_batch_wait / ba_batch_run on the host (milliseconds; 0 = none): past it the call fails instead of blocking
    pub fn ba_set_wait_limit_ms(ms: u64);
    pub fn block_percent_len(len: usize, p: f32) -> usize;
    pub fn ba_last_error() -> *const c_char;
}

pub fn last_error() -> String { unsafe { CStr::from_ptr(ba_last_error()).to_string_lossy().into_owned() } }

pub(crate) fn op_of(code: u8) -> Operation {   // cigar.rs:10-33 / c/block_aligner.h:17-57
    match code { 1 => Operation::M, 2 => Operation::Eq, 3 => Operation::X, 4 => Operation::I, 5 => Operation::D, _ => Operation::Sentinel }
}

/// Many pairs, one launch: what to use instead of a loop over `Block::align` (a loop of single-pair launches costs a launch and a
/// device round trip per pair: ~50 us per call for a pair that fits one block, ~0.8 ms for a 900-residue pair -- slower than the CPU it
/// replaces; the batch keeps the whole device busy: INTEGRATION.md). Sequences are raw bytes in one pool; results come back in
/// the caller's order.
pub struct HipBatch { h: *mut c_void, n: usize, trace: bool }

impl HipBatch {
    pub fn new<M: Matrix>(matrix: &M, gaps: Gaps, size: std::ops::RangeInclusive<usize>, x_drop: i32, mode: u32,
                             pool: &[u8], q: &[(u64, u32)], r: &[(u64, u32)]) -> Result<Self, String> {
        assert_eq!(q.len(), r.len());
        let (q_off, q_len): (Vec<u64>, Vec<u32>) = q.iter().cloned().unzip();
        let (r_off, r_len): (Vec<u64>, Vec<u32>) = r.iter().cloned().unzip();
        let marg = MatrixArg::of(matrix);
        let h = unsafe { ba_batch_create(M::HIP_KIND, marg.ptr(), gaps, SizeRange { min: *size.start(), max: *size.end() }, x_drop, mode,
                                         pool.as_ptr(), q_off.as_ptr(), q_len.as_ptr(), r_off.as_ptr(), r_len.as_ptr(), q.len()) };
        if h.is_null() { Err(last_error()) } else { Ok(HipBatch { h, n: q.len(), trace: mode & BA_TRACE != 0 }) }
    }
    /// One pass over the batch; returns the kernel time in milliseconds.
    pub fn run(&mut self) -> Result<f32, String> {
        let mut ms = 0f32;
        if unsafe { ba_batch_run(self.h, &mut ms) } != 0 { Err(last_error()) } else { Ok(ms) }
    }
    /// Asynchronous halves of `run` (two `HipBatch` objects alive: reload and launch one while the other runs).
    pub fn launch(&mut self) -> Result<(), String> { if unsafe { ba_batch_launch(self.h) } != 0 { Err(last_error()) } else { Ok(()) } }
    pub fn wait(&mut self) -> Result<f32, String> {
        let mut ms = 0f32;
        if unsafe { ba_batch_wait(self.h, &mut ms) } != 0 { Err(last_error()) } else { Ok(ms) }
    }
    /// Between `launch` and `wait`: gather the CIGAR runs on the device behind the alignment kernels (`cigars` is then one copy).
    pub fn gather_cigars(&mut self) -> Result<(), String> {
        if unsafe { ba_batch_compact_cigars(self.h, std::ptr::null_mut(), 0) } != 0 { Err(last_error()) } else { Ok(()) }
    }
    pub fn results(&self) -> Result<Vec<AlignResult>, String> {
        let (mut s, mut qi, mut ri, mut st) = (vec![0i32; self.n], vec![0u32; self.n], vec![0u32; self.n], vec![0u32; self.n]);
        if unsafe { ba_batch_results(self.h, s.as_mut_ptr(), qi.as_mut_ptr(), ri.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut(), st.as_mut_ptr()) } != 0 { return Err(last_error()); }
        if let Some(p) = st.iter().position(|&x| x != 0) { return Err(format!("pair {} failed on the device (status {:#x})", p, st[p])); }
        Ok((0..self.n).map(|p| AlignResult { score: s[p], query_idx: qi[p] as usize, reference_idx: ri[p] as usize }).collect())
    }
    /// The CIGAR of every pair as runs, in alignment order (what `Cigar::to_vec` returns, cigar.rs:138-145).
    pub fn cigars(&self) -> Result<Vec<Vec<OpLen>>, String> {
        assert!(self.trace);
        let mut len = vec![0u32; self.n];
        if unsafe { ba_batch_results(self.h, std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(), len.as_mut_ptr(), std::ptr::null_mut()) } != 0 { return Err(last_error()); }
        let total: u64 = len.iter().map(|&x| x as u64).sum();
        let mut runs = vec![0u32; total as usize];
        if total > 0 && unsafe { ba_batch_cigars(self.h, runs.as_mut_ptr(), total) } != 0 { return Err(last_error()); }
        let mut out = Vec::with_capacity(self.n);
        let mut at = 0usize;
        for &l in &len {
            out.push(runs[at..at + l as usize].iter().map(|&x| OpLen { op: op_of((x & 15) as u8), len: (x >> 4) as usize }).collect());
            at += l as usize;
        }
        Ok(out)
    }
}
impl Drop for HipBatch { fn drop(&mut self) { unsafe { ba_batch_destroy(self.h) } } }

/// Many pairs, each with its own block range `percent_len(len, min_percent) ..= percent_len(len, max_percent)` of its longer sequence
/// (lib.rs:109-111, as examples/nanopore_bench_global.rs:144-171 aligns its reads): the library bins the pairs by range, runs every bin as an
/// ordinary batch and returns the results in the caller's order.
pub struct HipSizedBatch { h: *mut c_void, n: usize }

impl HipSizedBatch {
    pub fn with_percent_len<M: Matrix>(matrix: &M, gaps: Gaps, min_percent: f32, max_percent: f32, x_drop: i32, mode: u32,
                                       pool: &[u8], q: &[(u64, u32)], r: &[(u64, u32)]) -> Result<Self, String> {
        assert_eq!(q.len(), r.len());
        let (q_off, q_len): (Vec<u64>, Vec<u32>) = q.iter().cloned().unzip();
        let (r_off, r_len): (Vec<u64>, Vec<u32>) = r.iter().cloned().unzip();
        let marg = MatrixArg::of(matrix);
        let h = unsafe { ba_sized_batch_create_percent(M::HIP_KIND, marg.ptr(), gaps, min_percent, max_percent, x_drop, mode,
                                                       pool.as_ptr(), q_off.as_ptr(), q_len.as_ptr(), r_off.as_ptr(), r_len.as_ptr(), q.len()) };
        if h.is_null() { Err(last_error()) } else { Ok(HipSizedBatch { h, n: q.len() }) }
    }
    pub fn run(&mut self) -> Result<f32, String> {
        let mut ms = 0f32;
        if unsafe { ba_sized_batch_run(self.h, &mut ms) } != 0 { Err(last_error()) } else { Ok(ms) }
    }
    pub fn results(&self) -> Result<Vec<AlignResult>, String> {
        let (mut s, mut qi, mut ri, mut st) = (vec![0i32; self.n], vec![0u32; self.n], vec![0u32; self.n], vec![0u32; self.n]);
        if unsafe { ba_sized_batch_results(self.h, s.as_mut_ptr(), qi.as_mut_ptr(), ri.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut(), st.as_mut_ptr()) } != 0 { return Err(last_error()); }
        if let Some(p) = st.iter().position(|&x| x != 0) { return Err(format!("pair {} failed on the device (status {:#x})", p, st[p])); }
        Ok((0..self.n).map(|p| AlignResult { score: s[p], query_idx: qi[p] as usize, reference_idx: ri[p] as usize }).collect())
    }
}
impl Drop for HipSizedBatch { fn drop(&mut self) { unsafe { ba_sized_batch_destroy(self.h) } } }
