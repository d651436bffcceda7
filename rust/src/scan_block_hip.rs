// UNVERIFIED SOURCE (no Rust toolchain in the build image). src/scan_block.rs under #[cfg(feature = "simd_hip")]: the public
// types of /root/reference/src/scan_block.rs -- Block, Trace, Rectangle, PaddedBytes, AlignResult -- with the reference's
// signatures, method for method (line numbers of the reference in the comments), forwarding to the C ABI (src/hip.rs).
// PaddedBytes and Cigar stay the crate's own host structures; only the DP (align_core, place_block*, the trace) lives on the device.
//
// A loop of Block::align calls is one device launch per pair (see hip::HipBatch for the route the device is built for).
#![cfg(feature = "simd_hip")]

use std::ops::RangeInclusive;
use std::os::raw::c_void;
use crate::cigar::{Cigar, OpLen};
use crate::hip::{self, HipMatrix, SizeRange};
use crate::scores::{Gaps, Matrix};
pub use crate::hip::{AlignResult, Rectangle};          // scan_block.rs:1889-1893, 1696-1701: same fields

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
    pub fn align<M: Matrix + HipMatrix>(&mut self, query: &PaddedBytes, reference: &PaddedBytes, matrix: &M, gaps: Gaps, size: RangeInclusive<usize>, x_drop: i32) {
        unsafe {
            hip::block_align_padded_generic(self.trace.handle, M::HIP_KIND, query.as_ptr(0), query.len(), reference.as_ptr(0), reference.len(),
                                            matrix.hip_ptr(), gaps, SizeRange { min: *size.start(), max: *size.end() }, x_drop);
            self.res = hip::block_res_generic(self.trace.handle);
        }
    }

    /// scan_block.rs:884-902: doubling the minimum block size until the target score is reached.
    pub fn align_exp<M: Matrix + HipMatrix>(&mut self, query: &PaddedBytes, reference: &PaddedBytes, matrix: &M, gaps: Gaps, size: RangeInclusive<usize>, x_drop: i32, target_score: i32) -> Option<usize> {
        let mut min_size = if *size.start() < 16 { 16 } else { *size.start() };   // L (scan_block.rs:853)
        let max_size = if *size.end() < 16 { 16 } else { *size.end() };
        while min_size <= max_size {
            self.align(query, reference, matrix, gaps, min_size..=max_size, x_drop);
            if self.res.score >= target_score { return Some(min_size); }
            min_size *= 2;   // (usize: cannot overflow before exceeding max_size < 2^16)
        }
        None
    }

    /// scan_block.rs:942-968.
    pub fn align_profile(&mut self, query: &PaddedBytes, profile: &AAProfileHip, size: RangeInclusive<usize>, x_drop: i32) {
        unsafe {
            hip::block_align_profile_padded_generic(self.trace.handle, query.as_ptr(0), query.len(), profile.handle as *const c_void,
                                                    SizeRange { min: *size.start(), max: *size.end() }, x_drop);
            self.res = hip::block_res_generic(self.trace.handle);
        }
    }

    /// scan_block.rs:974-992.
    pub fn align_profile_exp(&mut self, query: &PaddedBytes, profile: &AAProfileHip, size: RangeInclusive<usize>, x_drop: i32, target_score: i32) -> Option<usize> {
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
            let mut v = vec![Rectangle { row: 0, col: 0, width: 0, height: 0 }; n];
            hip::block_trace_blocks_generic(self.handle, v.as_mut_ptr(), n);
            v
        }
    }
}
impl Drop for Trace { fn drop(&mut self) { unsafe { hip::block_free_generic(self.handle) } } }

/// scan_block.rs:1790-1793: unchanged host structure -- [NULL] + converted bytes + NULL x block_size.
#[derive(Clone, PartialEq, Debug)]
pub struct PaddedBytes { s: Vec<u8>, len: usize }

impl PaddedBytes {
    /// scan_block.rs:1798-1803
    pub fn new<M: Matrix>(len: usize, block_size: usize) -> Self {
        Self { s: vec![M::convert_char(M::NULL); 1 + len + block_size], len }
    }
    /// scan_block.rs:1806-1812
    pub fn set_bytes<M: Matrix>(&mut self, b: &[u8], block_size: usize) {
        self.s[0] = M::convert_char(M::NULL);
        self.s[1..1 + b.len()].copy_from_slice(b);
        self.s[1..1 + b.len()].iter_mut().for_each(|c| *c = M::convert_char(*c));
        self.s[1 + b.len()..1 + b.len() + block_size].fill(M::convert_char(M::NULL));
        self.len = b.len();
    }
    /// scan_block.rs:1815-1822
    pub fn set_bytes_rev<M: Matrix>(&mut self, b: &[u8], block_size: usize) {
        self.s[0] = M::convert_char(M::NULL);
        self.s[1..1 + b.len()].copy_from_slice(b);
        self.s[1..1 + b.len()].reverse();
        self.s[1..1 + b.len()].iter_mut().for_each(|c| *c = M::convert_char(*c));
        self.s[1 + b.len()..1 + b.len() + block_size].fill(M::convert_char(M::NULL));
        self.len = b.len();
    }
    /// scan_block.rs:1829-1836
    pub fn from_bytes<M: Matrix>(b: &[u8], block_size: usize) -> Self {
        let mut v = b.to_owned();
        let len = v.len();
        v.insert(0, M::NULL);
        v.resize(v.len() + block_size, M::NULL);
        v.iter_mut().for_each(|c| *c = M::convert_char(*c));
        Self { s: v, len }
    }
    /// scan_block.rs:1843-1845
    pub fn from_str<M: Matrix>(s: &str, block_size: usize) -> Self { Self::from_bytes::<M>(s.as_bytes(), block_size) }
    /// scan_block.rs:1852-1859
    pub fn from_string<M: Matrix>(s: String, block_size: usize) -> Self { Self::from_bytes::<M>(s.as_bytes(), block_size) }
    /// scan_block.rs:1863-1878 (get / set / as_ptr as the reference has them, crate-internal)
    #[inline] pub unsafe fn get(&self, i: usize) -> u8 { *self.s.as_ptr().add(i) }
    #[inline] pub unsafe fn set(&mut self, i: usize, c: u8) { *self.s.as_mut_ptr().add(i) = c; }
    #[inline] pub unsafe fn as_ptr(&self, i: usize) -> *const u8 { self.s.as_ptr().add(i) }
    /// scan_block.rs:1881-1883
    #[inline] pub fn len(&self) -> usize { self.len }
}

/// scores.rs:452-468 behind the library's AAProfile object (the per-position tables are repacked for the device by the library);
/// method for method the `Profile` surface of scores.rs:470-715 that callers use.
pub struct AAProfileHip { pub(crate) handle: *mut c_void }
impl AAProfileHip {
    pub fn new(str_len: usize, block_size: usize, gap_extend: i8) -> Self { Self { handle: unsafe { hip::block_new_aaprofile(str_len, block_size, gap_extend) } } }   // scores.rs:476-501
    pub fn from_bytes(b: &[u8], block_size: usize, match_score: i8, mismatch_score: i8, gap_open_C: i8, gap_close_C: i8, gap_open_R: i8, gap_extend: i8) -> Self {   // scores.rs:503-530
        let mut p = Self::new(b.len(), block_size, gap_extend);
        for i in 0..b.len() { for c in b'A'..=b'Z' { p.set(i + 1, c, if c == b[i].to_ascii_uppercase() { match_score } else { mismatch_score }); } }
        p.set_all_gap_open_C(gap_open_C); p.set_all_gap_close_C(gap_close_C); p.set_all_gap_open_R(gap_open_R);
        p
    }
    pub fn len(&self) -> usize { unsafe { hip::block_len_aaprofile(self.handle) } }                                       // scores.rs:539-541
    pub fn clear(&mut self, str_len: usize, block_size: usize) { unsafe { hip::block_clear_aaprofile(self.handle, str_len, block_size) } }   // scores.rs:543-560
    pub fn set(&mut self, i: usize, b: u8, score: i8) { unsafe { hip::block_set_aaprofile(self.handle, i, b, score) } }    // scores.rs:562-571
    pub fn set_all(&mut self, order: &[u8], scores: &[i8], left_shift: usize, right_shift: usize) {                        // scores.rs:573-575
        unsafe { hip::block_set_all_aaprofile(self.handle, order.as_ptr(), order.len(), scores.as_ptr(), scores.len(), left_shift, right_shift) }
    }
    pub fn set_all_rev(&mut self, order: &[u8], scores: &[i8], left_shift: usize, right_shift: usize) {                    // scores.rs:577-579
        unsafe { hip::block_set_all_rev_aaprofile(self.handle, order.as_ptr(), order.len(), scores.as_ptr(), scores.len(), left_shift, right_shift) }
    }
    pub fn set_gap_open_C(&mut self, i: usize, gap: i8) { unsafe { hip::block_set_gap_open_C_aaprofile(self.handle, i, gap) } }     // scores.rs:581-590
    pub fn set_gap_close_C(&mut self, i: usize, gap: i8) { unsafe { hip::block_set_gap_close_C_aaprofile(self.handle, i, gap) } }
    pub fn set_gap_open_R(&mut self, i: usize, gap: i8) { unsafe { hip::block_set_gap_open_R_aaprofile(self.handle, i, gap) } }
    pub fn set_all_gap_open_C(&mut self, gap: i8) { unsafe { hip::block_set_all_gap_open_C_aaprofile(self.handle, gap) } }           // scores.rs:592-613
    pub fn set_all_gap_close_C(&mut self, gap: i8) { unsafe { hip::block_set_all_gap_close_C_aaprofile(self.handle, gap) } }
    pub fn set_all_gap_open_R(&mut self, gap: i8) { unsafe { hip::block_set_all_gap_open_R_aaprofile(self.handle, gap) } }
    pub fn get(&self, i: usize, b: u8) -> i8 { unsafe { hip::block_get_aaprofile(self.handle, i, b) } }                   // scores.rs:615-622
    pub fn get_gap_extend(&self) -> i8 { unsafe { hip::block_get_gap_extend_aaprofile(self.handle) } }                    // scores.rs:640-642
}
impl Drop for AAProfileHip { fn drop(&mut self) { unsafe { hip::block_free_aaprofile(self.handle) } } }
