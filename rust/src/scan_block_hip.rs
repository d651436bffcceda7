// UNVERIFIED SOURCE (no Rust toolchain in the build image). The `simd_hip` bodies of Block's public methods: Block keeps its
// const-generic parameters and signatures (/root/reference/src/scan_block.rs:89, 798-992) and forwards to the C ABI.
// src/scan_block.rs, under #[cfg(feature = "simd_hip")]: Block keeps its type parameters and public methods.
// new(): handle = block_new_generic(TRACE as u32 | (X_DROP as u32) << 1 | (LOCAL_START as u32) << 2
//                                   | (FREE_QUERY_START_GAPS as u32) << 3 | (FREE_QUERY_END_GAPS as u32) << 4, ...)
impl<const TRACE: bool, const X_DROP: bool, const LOCAL_START: bool, const FREE_QUERY_START_GAPS: bool, const FREE_QUERY_END_GAPS: bool>
    Block<TRACE, X_DROP, LOCAL_START, FREE_QUERY_START_GAPS, FREE_QUERY_END_GAPS> {
    pub fn align_profile(&mut self, q: &PaddedBytes, p: &AAProfile, size: RangeInclusive<usize>, x_drop: i32) {
        unsafe { hip::block_align_profile_generic(self.handle, q.hip_handle(), p.hip_handle(),
                                                  hip::SizeRange { min: *size.start(), max: *size.end() }, x_drop) }
        self.res = unsafe { hip::block_res_generic(self.handle) };
    }
    pub fn align<M: Matrix>(&mut self, q: &PaddedBytes, r: &PaddedBytes, m: &M, gaps: Gaps, size: RangeInclusive<usize>, x_drop: i32) {
        unsafe { hip::block_align_generic(self.handle, M::HIP_KIND, q.hip_handle(), r.hip_handle(), m as *const M as *const _,
                                          gaps, hip::SizeRange { min: *size.start(), max: *size.end() }, x_drop) }
        self.res = unsafe { hip::block_res_generic(self.handle) };
    }
}
