// UNCOMPILED SOURCE (no Rust toolchain in the build image). Under #[cfg(feature = "simd_hip")] src/scan_block.rs (scan_block.rs.patch)
// compiles its CPU implementation out and takes Block and Trace from this file: the reference's signatures, method for method (line
// numbers of the reference in the comments), forwarding to the C ABI (src/hip.rs). PaddedBytes, AlignResult, Rectangle, Cigar, the
// matrices and AAProfile stay the crate's own host structures; only the DP (align_core, place_block*, the trace) lives on the device.
//
// A loop of Block::align calls is one device launch per pair (see hip::HipBatch for the route the device is built for).

use std::ops::RangeInclusive;
use std::os::raw::c_void;
use crate::cigar::{Cigar, OpLen};
use crate::hip::{self, MatrixArg, SizeRange};
use crate::scores::{Gaps, Matrix, Profile};
use super::{AlignResult, PaddedBytes, Rectangle};

/// scan_block.rs:89-92. The const parameters select the mode bits of the device handle.
pub struct Block<const TRACE: bool, const X_DROP: bool = false, const LOCAL_START: bool = false, const FREE_QUERY_START_GAPS: bool = false, const FREE_QUERY_END_GAPS: bool = false> {
    res: AlignResult,
    trace: Trace,             // (the handle lives here: Block::trace() hands out a reference, as the reference does)
}

impl<const TRACE: bool, const X_DROP: bool, const LOCAL_START: bool, const FREE_QUERY_START_GAPS: bool, const FREE_QUERY_END_GAPS: bool>
    Block<{ TRACE }, { X_DROP }, { LOCAL_START }, { FREE_QUERY_START_GAPS }, { FREE_QUERY_END_GAPS }> {
    const MODE: u32 = (TRACE as u32) | (X_DROP as u32) << 1 | (LOCAL_START as u32) << 2 | (FREE_QUERY_START_GAPS as u32) << 3 | (FREE_QUERY_END_GAPS as u32) << 4;

    /// scan_block.rs:798-805: allocates everything an alignment within these bounds will need (on the device).
    pub fn new(query_len: usize, reference_len: usize, max_size: usize) -> Self {
        assert!(max_size.is_power_of_two(), "Block size must be a power of two!");
        let handle = unsafe { hip::block_new_generic(Self::MODE, query_len, reference_len, max_size) };
        assert!(!handle.is_null());
        Self { res: AlignResult { score: 0, query_idx: 0, reference_idx: 0 }, trace: Trace { handle } }
    }

    /// scan_block.rs:847-878. The preconditions the reference asserts are checked by the library (same messages; panic = abort).
    pub fn align<M: Matrix>(&mut self, query: &PaddedBytes, reference: &PaddedBytes, matrix: &M, gaps: Gaps, size: RangeInclusive<usize>, x_drop: i32) {
        let marg = MatrixArg::of(matrix);
        unsafe {
            hip::block_align_padded_generic(self.trace.handle, M::HIP_KIND, query.as_ptr(0), query.len(), reference.as_ptr(0), reference.len(),
                                            marg.ptr(), gaps, SizeRange { min: *size.start(), max: *size.end() }, x_drop);
            self.res = hip::block_res_generic(self.trace.handle);
        }
    }

    /// scan_block.rs:884-902: doubling the minimum block size until the target score is reached.
    pub fn align_exp<M: Matrix>(&mut self, query: &PaddedBytes, reference: &PaddedBytes, matrix: &M, gaps: Gaps, size: RangeInclusive<usize>, x_drop: i32, target_score: i32) -> Option<usize> {
        let mut min_size = if *size.start() < 16 { 16 } else { *size.start() };   // L (scan_block.rs:853)
        let max_size = if *size.end() < 16 { 16 } else { *size.end() };
        while min_size <= max_size {
            self.align(query, reference, matrix, gaps, min_size..=max_size, x_drop);
            if self.res.score >= target_score { return Some(min_size); }
            min_size *= 2;   // (usize: cannot overflow before exceeding max_size < 2^16)
        }
        None
    }

    /// scan_block.rs:942-968. The crate's own profile object: its arrays (Profile::hip_raw, scores.rs.patch) are handed to a library-side
    /// profile for the duration of the call (the per-position gap costs are i8 values held as i16, scores.rs:581-613).
    pub fn align_profile<P: Profile>(&mut self, query: &PaddedBytes, profile: &P, size: RangeInclusive<usize>, x_drop: i32) {
        let (pos_aa, go_c, gc_c, go_r, positions, str_len, gap_extend) = profile.hip_raw();
        unsafe {
            let to8 = |p: *const i16| (0..positions).map(|k| *p.add(k) as i8).collect::<Vec<i8>>();
            let (a, b, c) = (to8(go_c), to8(gc_c), to8(go_r));
            let h = hip::block_new_aaprofile(str_len, positions - str_len - 1, gap_extend);   // (positions = str_len + block_size + 1, scores.rs:476-478)
            assert!(!h.is_null());
            assert_eq!(hip::ba_aaprofile_set_raw(h, pos_aa, a.as_ptr(), b.as_ptr(), c.as_ptr(), positions), 0, "{}", hip::last_error());
            hip::block_align_profile_padded_generic(self.trace.handle, query.as_ptr(0), query.len(), h as *const c_void,
                                                    SizeRange { min: *size.start(), max: *size.end() }, x_drop);
            self.res = hip::block_res_generic(self.trace.handle);
            hip::block_free_aaprofile(h);
        }
    }

    /// scan_block.rs:974-992.
    pub fn align_profile_exp<P: Profile>(&mut self, query: &PaddedBytes, profile: &P, size: RangeInclusive<usize>, x_drop: i32, target_score: i32) -> Option<usize> {
        let mut min_size = if *size.start() < 16 { 16 } else { *size.start() };
        let max_size = if *size.end() < 16 { 16 } else { *size.end() };
        while min_size <= max_size {
            self.align_profile(query, profile, min_size..=max_size, x_drop);
            if self.res.score >= target_score { return Some(min_size); }
            min_size *= 2;
        }
        None
    }

    /// scan_block.rs:1235-1237.
    #[inline] pub fn res(&self) -> AlignResult { self.res }

    /// scan_block.rs:1241-1244.
    #[inline] pub fn trace(&self) -> &Trace { assert!(TRACE); &self.trace }
}

/// scan_block.rs:1344-1359: the trace of the last alignment of a Block. It lives on the device; this is the handle to it.
pub struct Trace { handle: *mut c_void }

impl Trace {
    fn fill(&self, cigar: &mut Cigar, f: impl FnOnce(*mut c_void)) {
        unsafe {
            let c = hip::block_new_cigar(usize::MAX / 4, usize::MAX / 4);   // (host object: the capacity is only compared, nothing is allocated for it)
            f(c);
            let n = hip::block_len_cigar(c);
            let runs: Vec<OpLen> = (0..n).map(|k| { let o = hip::block_get_cigar(c, k); OpLen { op: hip::op_of(o.op), len: o.len } }).collect();
            hip::block_free_cigar(c);
            cigar.set_runs(&runs);   // cigar.rs.patch: replaces the contents by these runs (alignment order)
        }
    }
    /// scan_block.rs:1469-1471: the CIGAR (M / I / D) of the alignment that ends at (i, j).
    pub fn cigar(&self, i: usize, j: usize, cigar: &mut Cigar) {
        let h = self.handle;
        self.fill(cigar, |c| unsafe { hip::block_cigar_generic(h, i, j, c) });
    }
    /// scan_block.rs:1478-1480: with = / X instead of M (the sequences of the aligned pair are resident on the device).
    pub fn cigar_eq(&self, _query: &PaddedBytes, _reference: &PaddedBytes, i: usize, j: usize, cigar: &mut Cigar) {
        let h = self.handle;
        self.fill(cigar, |c| unsafe { hip::block_cigar_eq_generic(h, std::ptr::null(), std::ptr::null(), i, j, c) });
    }
    /// scan_block.rs:1676-1691: the rectangles computed for the alignment, in fill order.
    pub fn blocks(&self) -> Vec<Rectangle> {
        unsafe {
            let n = hip::block_trace_blocks_generic(self.handle, std::ptr::null_mut(), 0);
            let mut v = vec![hip::CRectangle { row: 0, col: 0, width: 0, height: 0 }; n];
            hip::block_trace_blocks_generic(self.handle, v.as_mut_ptr(), n);
            v.iter().map(|r| Rectangle { row: r.row, col: r.col, width: r.width, height: r.height }).collect()
        }
    }
}
impl Drop for Trace { fn drop(&mut self) { unsafe { hip::block_free_generic(self.handle) } } }
