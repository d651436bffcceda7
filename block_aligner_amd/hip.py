"""ctypes binding of libblock_aligner_hip.so (include/block_aligner_hip.h) and a host-side mirror of the reference's
Rust API for this path: `PaddedBytes`, `Block` (const-generic modes as constructor flags), `Cigar`, `AlignResult`,
`percent_len` (src/scan_block.rs, src/cigar.rs, src/lib.rs), plus `BatchAligner` over the batch launcher.

There is no CPU fallback: importing works anywhere (so the symbol table can be checked without a GPU), but every
alignment call goes to the HIP kernels and fails loudly if the library or a gfx950 device is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import scores as S

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libblock_aligner_hip.so")

TRACE, X_DROP, LOCAL_START, FREE_QUERY_START_GAPS, FREE_QUERY_END_GAPS, CIGAR_EQ = 1, 2, 4, 8, 16, 32
OP_CHARS = " M=XID"


class GapsC(C.Structure):
    _fields_ = [("open", C.c_int8), ("extend", C.c_int8)]


class SizeRangeC(C.Structure):
    _fields_ = [("min", C.c_size_t), ("max", C.c_size_t)]


class RectangleC(C.Structure):
    _fields_ = [("row", C.c_size_t), ("col", C.c_size_t), ("width", C.c_size_t), ("height", C.c_size_t)]


class AlignResultC(C.Structure):
    _fields_ = [("score", C.c_int32), ("query_idx", C.c_size_t), ("reference_idx", C.c_size_t)]


class OpLenC(C.Structure):
    _fields_ = [("op", C.c_uint8), ("len", C.c_size_t)]


@dataclass(frozen=True)
class AlignResult:
    score: int
    query_idx: int
    reference_idx: int


# The same kernels behind a host side that reads the BA_* development switches (environment variables): for the tests that force
# a code path on small inputs and for the measurement scripts under tools/. The release library reads no environment variables.
DEV_LIB_PATH = os.path.join(_HERE, "lib", "libblock_aligner_hip_dev.so")
_lib = None
_loaded = {}


def use_library(path: str) -> None:
    """Route every later call through another build of the library (objects must not outlive the switch)."""
    global LIB_PATH, _lib
    LIB_PATH = path
    _lib = _loaded.get(path)


def lib() -> C.CDLL:
    """Load the HIP library; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None and LIB_PATH in _loaded:
        _lib = _loaded[LIB_PATH]
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                               "block_aligner_amd has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp, sz, u32, i32, i8, u8p = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int32, C.c_int8, C.c_char_p
        L.ba_last_error.restype = C.c_char_p
        L.ba_host_alloc.restype = vp
        L.ba_host_alloc.argtypes = [C.c_uint64]
        L.ba_host_free.argtypes = [vp]
        L.ba_set_device.argtypes = [C.c_int]
        L.block_percent_len.restype = sz
        L.block_percent_len.argtypes = [sz, C.c_float]
        for k in ("aa", "nuc", "bytes"):
            getattr(L, f"block_new_padded_{k}").restype = vp
            getattr(L, f"block_new_padded_{k}").argtypes = [sz, sz]
            getattr(L, f"block_set_bytes_padded_{k}").argtypes = [vp, u8p, sz, sz]
            getattr(L, f"block_free_padded_{k}").argtypes = [vp]
        for k in ("aa", "nuc"):
            getattr(L, f"block_set_bytes_rev_padded_{k}").argtypes = [vp, u8p, sz, sz]
        L.block_new_cigar.restype = vp
        L.block_new_cigar.argtypes = [sz, sz]
        L.block_get_cigar.restype = OpLenC
        L.block_get_cigar.argtypes = [vp, sz]
        L.block_len_cigar.restype = sz
        L.block_len_cigar.argtypes = [vp]
        L.block_free_cigar.argtypes = [vp]
        L.block_new_generic.restype = vp
        L.block_new_generic.argtypes = [u32, sz, sz, sz]
        L.block_align_generic.argtypes = [vp, C.c_int, vp, vp, vp, GapsC, SizeRangeC, i32]
        L.block_res_generic.restype = AlignResultC
        L.block_res_generic.argtypes = [vp]
        L.block_cigar_generic.argtypes = [vp, sz, sz, vp]
        L.block_cigar_eq_generic.argtypes = [vp, vp, vp, sz, sz, vp]
        L.block_free_generic.argtypes = [vp]
        L.block_align_profile_generic.argtypes = [vp, vp, vp, SizeRangeC, i32]
        L.block_trace_blocks_generic.restype = sz
        L.block_trace_blocks_generic.argtypes = [vp, vp, sz]
        L.block_batch_align_exp.argtypes = [C.c_int, vp, GapsC, SizeRangeC, i32, i32, u32, vp, vp, vp, vp, vp, sz, vp, vp]
        L.block_batch_align_profile_exp.argtypes = [vp, SizeRangeC, i32, i32, u32, vp, vp, vp, sz, vp, vp]
        L.block_new_aaprofile.restype = vp
        L.block_new_aaprofile.argtypes = [sz, sz, i8]
        L.block_free_aaprofile.argtypes = [vp]
        L.ba_aaprofile_set_raw.argtypes = [vp, vp, vp, vp, vp, sz]
        L.ba_batch_create_profile.restype = vp
        L.ba_batch_create_profile.argtypes = [vp, SizeRangeC, i32, u32, vp, vp, vp, sz]
        L.ba_batch_create.restype = vp
        L.ba_batch_create.argtypes = [C.c_int, vp, GapsC, SizeRangeC, i32, u32, vp, vp, vp, vp, vp, sz]
        L.ba_batch_reload.argtypes = [vp, vp, vp, vp, vp, vp, sz]
        L.ba_batch_reload_profile.argtypes = [vp, vp, vp, vp, vp, sz]
        L.ba_batch_run.argtypes = [vp, C.POINTER(C.c_float)]
        L.ba_batch_launch.argtypes = [vp]
        L.ba_batch_wait.argtypes = [vp, C.POINTER(C.c_float)]
        L.ba_set_wait_limit_ms.argtypes = [C.c_uint64]; L.ba_set_wait_limit_ms.restype = None
        L.ba_batch_results.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.ba_batch_cigars.argtypes = [vp, vp, C.c_uint64]
        L.ba_batch_compact_cigars.argtypes = [vp, vp, C.c_uint64]
        L.ba_batch_surviving_cells.argtypes = [vp, vp]
        L.ba_batch_retried.argtypes = [vp]
        L.ba_device_memory.argtypes = [vp, vp]
        L.ba_batch_info.argtypes = [vp, vp]
        L.ba_batch_kernel.argtypes = [vp]
        L.ba_batch_geometry.argtypes = [vp]
        L.ba_batch_spec_cells.argtypes = [vp, vp]
        L.ba_build_id.restype = C.c_char_p
        L.ba_batch_destroy.argtypes = [vp]
        L.ba_multibatch_create.restype = vp
        L.ba_multibatch_create.argtypes = [C.c_int, vp, GapsC, SizeRangeC, i32, u32, vp, vp, vp, vp, vp, sz, vp, C.c_int]
        L.ba_multibatch_run.argtypes = [vp, C.POINTER(C.c_float)]
        L.ba_multibatch_results.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.ba_multibatch_cigars.argtypes = [vp, vp, C.c_uint64]
        L.ba_multibatch_parts.argtypes = [vp, vp, C.c_int]
        L.ba_sized_batch_create.restype = vp
        L.ba_sized_batch_create.argtypes = [C.c_int, vp, GapsC, vp, i32, u32, vp, vp, vp, vp, vp, sz]
        L.ba_sized_batch_create_percent.restype = vp
        L.ba_sized_batch_create_percent.argtypes = [C.c_int, vp, GapsC, C.c_float, C.c_float, i32, u32, vp, vp, vp, vp, vp, sz]
        L.ba_sized_batch_run.argtypes = [vp, C.POINTER(C.c_float)]
        L.ba_sized_batch_results.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.ba_sized_batch_cigars.argtypes = [vp, vp, C.c_uint64]
        L.ba_sized_batch_classes.argtypes = [vp, vp, vp, vp, vp, C.c_int]
        L.ba_sized_batch_destroy.argtypes = [vp]
        L.ba_multibatch_kernel_ms.argtypes = [vp, vp, C.c_int]
        L.ba_multibatch_destroy.argtypes = [vp]
        L.ba_shard_slices.argtypes = [vp, vp, sz, C.c_int, vp]
        _lib = L
        _loaded[LIB_PATH] = L
    return _lib


_pinned = {}   # base address -> (bytes, owning library) of the live ba_host_alloc buffers


def pinned_array(n: int, dtype=np.uint32) -> np.ndarray:
    """A numpy array over page-locked host memory (ba_host_alloc): pass it as `out=` to BatchAligner.compact_cigars / cigars. Free it with
    free_pinned(array) when done (nothing else frees it); arrays from anywhere else are refused where the device writes through them."""
    nbytes = int(n) * np.dtype(dtype).itemsize
    p = lib().ba_host_alloc(nbytes)
    if not p:
        raise RuntimeError(last_error())
    _pinned[p] = (nbytes, lib())
    return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p)).view(dtype)


def is_pinned(a: np.ndarray) -> bool:
    """True if `a` lies inside a live pinned_array() buffer."""
    addr = a.ctypes.data
    return any(base <= addr and addr + a.nbytes <= base + nb for base, (nb, _) in _pinned.items())


def free_pinned(a: np.ndarray) -> None:
    """ba_host_free for an array made by pinned_array (the array must not be used afterwards)."""
    ent = _pinned.pop(a.ctypes.data, None)
    if ent is None:
        raise ValueError("not the start of a pinned_array() buffer")
    ent[1].ba_host_free(a.ctypes.data)


def last_error() -> str:
    return lib().ba_last_error().decode()


def device_count() -> int:
    return lib().ba_device_count()


def set_device(d: int) -> None:
    if lib().ba_set_device(d):
        raise RuntimeError(last_error())


def device_memory():
    """(free, total) bytes of the selected device."""
    f, t = C.c_uint64(), C.c_uint64()
    if lib().ba_device_memory(C.byref(f), C.byref(t)):
        raise RuntimeError(last_error())
    return f.value, t.value


def percent_len(length: int, p: float) -> int:
    """lib.rs:109-111"""
    return lib().block_percent_len(length, p)


def _kind_name(matrix_cls) -> str:
    return {0: "aa", 1: "nuc", 2: "bytes"}[matrix_cls.KIND]


def _size(size) -> SizeRangeC:
    if isinstance(size, range):  # Rust `a..=b` written as range(a, b + 1)
        return SizeRangeC(size.start, size.stop - 1)
    return SizeRangeC(int(size[0]), int(size[1]))


class PaddedBytes:
    """scan_block.rs:1790-1884. `matrix_cls` plays the role of the `<M: Matrix>` type parameter."""

    def __init__(self, length: int, block_size: int, matrix_cls=S.AAMatrix):
        self.kind = matrix_cls.KIND
        self._k = _kind_name(matrix_cls)
        self._h = getattr(lib(), f"block_new_padded_{self._k}")(length, block_size)
        self._len = length
        self._cap = length

    new = classmethod(lambda cls, length, block_size, matrix_cls=S.AAMatrix: cls(length, block_size, matrix_cls))

    @classmethod
    def from_bytes(cls, b: bytes, block_size: int, matrix_cls=S.AAMatrix) -> "PaddedBytes":
        p = cls(len(b), block_size, matrix_cls)
        p.set_bytes(b, block_size)
        return p

    from_str = classmethod(lambda cls, s, block_size, matrix_cls=S.AAMatrix: cls.from_bytes(s.encode(), block_size, matrix_cls))

    def set_bytes(self, b: bytes, block_size: int) -> None:
        getattr(lib(), f"block_set_bytes_padded_{self._k}")(self._h, bytes(b), len(b), block_size)
        self._len = len(b)

    def set_bytes_rev(self, b: bytes, block_size: int) -> None:
        getattr(lib(), f"block_set_bytes_rev_padded_{self._k}")(self._h, bytes(b), len(b), block_size)
        self._len = len(b)

    def len(self) -> int:
        return self._len

    __len__ = len

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            getattr(_lib, f"block_free_padded_{self._k}")(self._h)
            self._h = None


class Cigar:
    """cigar.rs:42-163"""

    def __init__(self, query_len: int, reference_len: int):
        self._h = lib().block_new_cigar(query_len, reference_len)

    new = classmethod(lambda cls, q, r: cls(q, r))

    def len(self) -> int:
        return lib().block_len_cigar(self._h)

    __len__ = len

    def get(self, i: int):
        o = lib().block_get_cigar(self._h, i)
        return (o.op, o.len)

    def to_vec(self):
        return [self.get(i) for i in range(self.len())]

    def __str__(self) -> str:
        return "".join(f"{n}{OP_CHARS[op]}" for op, n in self.to_vec() if op)

    to_string = __str__

    def format(self, q: bytes, r: bytes):
        a, b, i, j = [], [], 0, 0
        for op, n in self.to_vec():
            for _ in range(n):
                if op in (1, 2, 3):
                    a.append(chr(q[i])); b.append(chr(r[j])); i += 1; j += 1
                elif op == 4:
                    a.append(chr(q[i])); b.append("-"); i += 1
                elif op == 5:
                    a.append("-"); b.append(chr(r[j])); j += 1
        return "".join(a), "".join(b)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.block_free_cigar(self._h)
            self._h = None


class _NativeProfile:
    """The library's AAProfile object (ffi.rs:60-195) filled from the numpy mirror in scores.AAProfile."""

    def __init__(self, p: S.AAProfile):
        L = lib()
        self._h = L.block_new_aaprofile(p.str_len, p.curr_len - p.str_len - 1, p.gap_extend)
        pos = np.ascontiguousarray(p.pos_aa[: p.curr_len], dtype=np.int8)
        g = [np.ascontiguousarray(a[: p.curr_len], dtype=np.int8) for a in (p.pos_gap_open_C, p.pos_gap_close_C, p.pos_gap_open_R)]
        if L.ba_aaprofile_set_raw(self._h, pos.ctypes.data, g[0].ctypes.data, g[1].ctypes.data, g[2].ctypes.data, p.curr_len):
            raise RuntimeError(last_error())

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.block_free_aaprofile(self._h)
            self._h = None


class _Trace:
    def __init__(self, block: "Block"):
        self._b = block

    def cigar(self, i: int, j: int, cigar: Cigar) -> None:
        lib().block_cigar_generic(self._b._h, i, j, cigar._h)

    def cigar_eq(self, q: PaddedBytes, r: PaddedBytes, i: int, j: int, cigar: Cigar) -> None:
        lib().block_cigar_eq_generic(self._b._h, q._h, r._h, i, j, cigar._h)

    def blocks(self):
        """Trace::blocks() (scan_block.rs:1676-1691): [(row, col, width, height), ...] in fill order."""
        n = lib().block_trace_blocks_generic(self._b._h, None, 0)
        arr = (RectangleC * max(n, 1))()
        lib().block_trace_blocks_generic(self._b._h, arr, n)
        return [(r.row, r.col, r.width, r.height) for r in arr[:n]]


class Block:
    """Block::<TRACE, X_DROP, LOCAL_START, FREE_QUERY_START_GAPS, FREE_QUERY_END_GAPS> (scan_block.rs:89, 798-1244)."""

    def __init__(self, query_len: int, reference_len: int, max_size: int, trace: bool = False, x_drop: bool = False,
                 local_start: bool = False, free_query_start_gaps: bool = False, free_query_end_gaps: bool = False):
        self.mode = (TRACE * trace) | (X_DROP * x_drop) | (LOCAL_START * local_start) | \
                    (FREE_QUERY_START_GAPS * free_query_start_gaps) | (FREE_QUERY_END_GAPS * free_query_end_gaps)
        self._h = lib().block_new_generic(self.mode, query_len, reference_len, max_size)

    new = classmethod(lambda cls, *a, **k: cls(*a, **k))

    def align(self, query: PaddedBytes, reference: PaddedBytes, matrix, gaps, size, x_drop: int = 0) -> None:
        raw = matrix.raw()
        g = GapsC(gaps.open, gaps.extend) if isinstance(gaps, S.Gaps) else GapsC(gaps[0], gaps[1])
        lib().block_align_generic(self._h, matrix.KIND, query._h, reference._h, raw.ctypes.data, g, _size(size), x_drop)

    def align_exp(self, query, reference, matrix, gaps, size, x_drop: int, target_score: int):
        """scan_block.rs:884-902: double the min block size until the score reaches the target."""
        s = _size(size)
        mn, mx = max(s.min, 16), max(s.max, 16)
        while mn <= mx:
            self.align(query, reference, matrix, gaps, (mn, mx), x_drop)
            if self.res().score >= target_score:
                return mn
            mn *= 2
        return None

    def align_profile(self, query: PaddedBytes, profile: S.AAProfile, size, x_drop: int = 0) -> None:
        """scan_block.rs:942-968"""
        native = _NativeProfile(profile)
        lib().block_align_profile_generic(self._h, query._h, native._h, _size(size), x_drop)

    def align_profile_exp(self, query, profile, size, x_drop: int, target_score: int):
        """scan_block.rs:974-992"""
        s = _size(size)
        mn, mx = max(s.min, 16), max(s.max, 16)
        while mn <= mx:
            self.align_profile(query, profile, (mn, mx), x_drop)
            if self.res().score >= target_score:
                return mn
            mn *= 2
        return None

    def res(self) -> AlignResult:
        r = lib().block_res_generic(self._h)
        return AlignResult(r.score, r.query_idx, r.reference_idx)

    def trace(self) -> _Trace:
        assert self.mode & TRACE, "Block was created without TRACE"
        return _Trace(self)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.block_free_generic(self._h)
            self._h = None


class BatchAligner:
    """Batch launcher: many independent pairs, one persistent kernel launch, one wavefront per pair.

    pool: uint8 array of raw sequence bytes; pair p = pool[q_off[p]:+q_len[p]] (query) vs pool[r_off[p]:+r_len[p]].
    """

    def __init__(self, matrix, gaps, size, x_drop: int, mode: int, pool, q_off, q_len, r_off, r_len):
        L = lib()
        self.n = len(q_len)
        self.mode = mode
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64); r_off = np.ascontiguousarray(r_off, dtype=np.uint64)
        q_len = np.ascontiguousarray(q_len, dtype=np.uint32); r_len = np.ascontiguousarray(r_len, dtype=np.uint32)
        raw = matrix.raw()
        g = GapsC(gaps.open, gaps.extend) if isinstance(gaps, S.Gaps) else GapsC(gaps[0], gaps[1])
        self._h = L.ba_batch_create(matrix.KIND, raw.ctypes.data, g, _size(size), x_drop, mode, pool.ctypes.data,
                                    q_off.ctypes.data, q_len.ctypes.data, r_off.ctypes.data, r_len.ctypes.data, self.n)
        if not self._h:
            raise RuntimeError(last_error())

    def reload(self, pool, q_off, q_len, r_off, r_len) -> None:
        """Replace the pairs and keep the device buffers (ba_batch_reload): the new set must fit the original one's sizes."""
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64); r_off = np.ascontiguousarray(r_off, dtype=np.uint64)
        q_len = np.ascontiguousarray(q_len, dtype=np.uint32); r_len = np.ascontiguousarray(r_len, dtype=np.uint32)
        if lib().ba_batch_reload(self._h, pool.ctypes.data, q_off.ctypes.data, q_len.ctypes.data, r_off.ctypes.data, r_len.ctypes.data, len(q_len)):
            raise RuntimeError(last_error())
        self.n = len(q_len)

    def run(self) -> float:
        """Launch and wait; returns the kernel's HIP-event time in milliseconds."""
        ms = C.c_float()
        if lib().ba_batch_run(self._h, C.byref(ms)):
            raise RuntimeError(last_error())
        return ms.value

    def launch(self) -> None:
        """Enqueue one pass on the batch's stream and return (ba_batch_launch); wait() collects it."""
        if lib().ba_batch_launch(self._h):
            raise RuntimeError(last_error())

    def compact_cigars(self, pinned_out=None) -> None:
        """Between launch() and wait(): gather the CIGAR runs on the device behind the kernels -- straight into `pinned_out` (an array
        from pinned_array(); pass the same array as cigars(out=...)) or into a device buffer (cigars() is then one copy)."""
        if pinned_out is not None and not is_pinned(pinned_out):
            raise ValueError("compact_cigars writes through this buffer from the device: it must come from pinned_array()")
        if lib().ba_batch_compact_cigars(self._h, pinned_out.ctypes.data if pinned_out is not None else None, pinned_out.size if pinned_out is not None else 0):
            raise RuntimeError(last_error())

    def wait(self) -> float:
        ms = C.c_float()
        if lib().ba_batch_wait(self._h, C.byref(ms)):
            raise RuntimeError(last_error())
        return ms.value

    def results(self):
        n = self.n
        out = dict(score=np.zeros(n, np.int32), query_idx=np.zeros(n, np.uint32), reference_idx=np.zeros(n, np.uint32),
                   cells=np.zeros(n, np.uint64), cigar_len=np.zeros(n, np.uint32), status=np.zeros(n, np.uint32))
        if lib().ba_batch_results(self._h, *(out[k].ctypes.data for k in ("score", "query_idx", "reference_idx", "cells", "cigar_len", "status"))):
            raise RuntimeError(last_error())
        return out

    def cigars(self, cigar_len=None, out=None):
        """-> (runs, offsets): runs[offsets[p]:offsets[p+1]] are pair p's packed (len << 4 | op) runs. out: a uint32 array to reuse
        (a fresh 600 MB array costs more in page faults than the copy into it)."""
        if cigar_len is None:
            cigar_len = self.results()["cigar_len"]
        off = np.zeros(self.n + 1, np.uint64)
        np.cumsum(cigar_len, out=off[1:])
        total = int(off[-1])
        runs = out[:total] if out is not None and out.size >= total else np.empty(total, np.uint32)
        if lib().ba_batch_cigars(self._h, runs.ctypes.data, runs.size):
            raise RuntimeError(last_error())
        return runs, off

    def surviving_cells(self):
        """TRACE batches: per pair, sum of width x height over Trace::blocks() (scan_block.rs:1676-1691)."""
        out = np.zeros(self.n, np.uint64)
        if lib().ba_batch_surviving_cells(self._h, out.ctypes.data):
            raise RuntimeError(last_error())
        return out

    def retried(self) -> int:
        """Pairs the last run re-ran with full-size trace slots (ba_batch_retried)."""
        return lib().ba_batch_retried(self._h)

    KERNELS = ("k_align", "k_multi", "k_quad", "k_small")

    def spec_cells(self) -> int:
        """Cells of the last run's speculative, untraced rectangles (X-drop + TRACE): a part of results()["cells"]."""
        o = C.c_uint64()
        if lib().ba_batch_spec_cells(self._h, C.byref(o)):
            raise RuntimeError(last_error())
        return int(o.value)

    def info(self):
        o = np.zeros(4, np.uint64)
        lib().ba_batch_info(self._h, o.ctypes.data)
        return dict(grid=int(o[0]), lds_bytes_per_wave=int(o[1]), trace_arena_bytes=int(o[2]), pool_bytes=int(o[3]),
                    kernel=self.KERNELS[lib().ba_batch_kernel(self._h)], geometry=lib().ba_batch_geometry(self._h))

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ba_batch_destroy(self._h)
            self._h = None

    __del__ = close


def shard_slices(q_len, r_len, parts: int) -> np.ndarray:
    """ba_shard_slices: boundaries of `parts` contiguous slices of near-equal summed |q| + |r| (needs no device)."""
    q_len = np.ascontiguousarray(q_len, dtype=np.uint32); r_len = np.ascontiguousarray(r_len, dtype=np.uint32)
    bounds = np.zeros(parts + 1, np.uint64)
    if lib().ba_shard_slices(q_len.ctypes.data, r_len.ctypes.data, len(q_len), parts, bounds.ctypes.data):
        raise RuntimeError(last_error())
    return bounds


class MultiBatchAligner:
    """One batch over several GPUs (ba_multibatch_*): contiguous cost-balanced slices, one per entry of `devices`."""

    def __init__(self, matrix, gaps, size, x_drop: int, mode: int, pool, q_off, q_len, r_off, r_len, devices):
        L = lib()
        self.n = len(q_len)
        self.mode = mode
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64); r_off = np.ascontiguousarray(r_off, dtype=np.uint64)
        q_len = np.ascontiguousarray(q_len, dtype=np.uint32); r_len = np.ascontiguousarray(r_len, dtype=np.uint32)
        raw = matrix.raw()
        g = GapsC(gaps.open, gaps.extend) if isinstance(gaps, S.Gaps) else GapsC(gaps[0], gaps[1])
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        self._h = L.ba_multibatch_create(matrix.KIND, raw.ctypes.data, g, _size(size), x_drop, mode, pool.ctypes.data, q_off.ctypes.data,
                                         q_len.ctypes.data, r_off.ctypes.data, r_len.ctypes.data, self.n, dev.ctypes.data, len(dev))
        if not self._h:
            raise RuntimeError(last_error())

    def run(self) -> float:
        ms = C.c_float()
        if lib().ba_multibatch_run(self._h, C.byref(ms)):
            raise RuntimeError(last_error())
        return ms.value

    def results(self):
        n = self.n
        out = dict(score=np.zeros(n, np.int32), query_idx=np.zeros(n, np.uint32), reference_idx=np.zeros(n, np.uint32),
                   cells=np.zeros(n, np.uint64), cigar_len=np.zeros(n, np.uint32), status=np.zeros(n, np.uint32))
        if lib().ba_multibatch_results(self._h, *(out[k].ctypes.data for k in ("score", "query_idx", "reference_idx", "cells", "cigar_len", "status"))):
            raise RuntimeError(last_error())
        return out

    def cigars(self, cigar_len=None):
        if cigar_len is None:
            cigar_len = self.results()["cigar_len"]
        off = np.zeros(self.n + 1, np.uint64)
        np.cumsum(cigar_len, out=off[1:])
        runs = np.zeros(int(off[-1]), np.uint32)
        if lib().ba_multibatch_cigars(self._h, runs.ctypes.data, runs.size):
            raise RuntimeError(last_error())
        return runs, off

    def kernel_ms(self):
        """Kernel time of every slice in the last run() (ms; HIP events on the slice's own stream)."""
        t = np.zeros(64, np.float32)
        k = lib().ba_multibatch_kernel_ms(self._h, t.ctypes.data, 64)
        return t[:k].copy()

    def parts(self):
        b = np.zeros(65, np.uint64)
        k = lib().ba_multibatch_parts(self._h, b.ctypes.data, 65)
        return b[: k + 1].copy()

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ba_multibatch_destroy(self._h)
            self._h = None

    __del__ = close


class SizedBatchAligner:
    """Every pair with its own block range (ba_sized_batch_*): `sizes` = an (n, 2) array of (min, max), or percent = (min_percent, max_percent) of
    the longer sequence's length per pair, as examples/nanopore_bench_global.rs:144-171 calls percent_len."""

    def __init__(self, matrix, gaps, x_drop: int, mode: int, pool, q_off, q_len, r_off, r_len, sizes=None, percent=None):
        L = lib()
        self.n = len(q_len)
        self.mode = mode
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64); r_off = np.ascontiguousarray(r_off, dtype=np.uint64)
        q_len = np.ascontiguousarray(q_len, dtype=np.uint32); r_len = np.ascontiguousarray(r_len, dtype=np.uint32)
        raw = matrix.raw()
        g = GapsC(gaps.open, gaps.extend) if isinstance(gaps, S.Gaps) else GapsC(gaps[0], gaps[1])
        if (sizes is None) == (percent is None):
            raise ValueError("give either sizes or percent")
        if sizes is not None:
            sz = np.ascontiguousarray(sizes, dtype=np.uint64).reshape(self.n, 2)   # SizeRange {uintptr min, max}
            self._h = L.ba_sized_batch_create(matrix.KIND, raw.ctypes.data, g, sz.ctypes.data, x_drop, mode, pool.ctypes.data, q_off.ctypes.data,
                                              q_len.ctypes.data, r_off.ctypes.data, r_len.ctypes.data, self.n)
        else:
            self._h = L.ba_sized_batch_create_percent(matrix.KIND, raw.ctypes.data, g, percent[0], percent[1], x_drop, mode, pool.ctypes.data, q_off.ctypes.data,
                                                      q_len.ctypes.data, r_off.ctypes.data, r_len.ctypes.data, self.n)
        if not self._h:
            raise RuntimeError(last_error())

    def run(self) -> float:
        ms = C.c_float()
        if lib().ba_sized_batch_run(self._h, C.byref(ms)):
            raise RuntimeError(last_error())
        return ms.value

    def results(self):
        n = self.n
        out = dict(score=np.zeros(n, np.int32), query_idx=np.zeros(n, np.uint32), reference_idx=np.zeros(n, np.uint32),
                   cells=np.zeros(n, np.uint64), cigar_len=np.zeros(n, np.uint32), status=np.zeros(n, np.uint32))
        if lib().ba_sized_batch_results(self._h, *(out[k].ctypes.data for k in ("score", "query_idx", "reference_idx", "cells", "cigar_len", "status"))):
            raise RuntimeError(last_error())
        return out

    def cigars(self, cigar_len=None):
        if cigar_len is None:
            cigar_len = self.results()["cigar_len"]
        off = np.zeros(self.n + 1, np.uint64)
        np.cumsum(cigar_len, out=off[1:])
        runs = np.zeros(int(off[-1]), np.uint32)
        if lib().ba_sized_batch_cigars(self._h, runs.ctypes.data, runs.size):
            raise RuntimeError(last_error())
        return runs, off

    def classes(self):
        """[(min, max, pairs, fill kernel, kernel ms of the last run)] per bin."""
        cap = 64
        while True:   # (the call returns the number of bins, however many fit: a second call with room for all of them if there are more)
            r = np.zeros((cap, 2), np.uint64); c = np.zeros(cap, np.uint64); k = np.zeros(cap, np.int32); t = np.zeros(cap, np.float32)
            nb = lib().ba_sized_batch_classes(self._h, r.ctypes.data, c.ctypes.data, k.ctypes.data, t.ctypes.data, cap)
            if nb <= cap:
                break
            cap = nb
        return [(int(r[i, 0]), int(r[i, 1]), int(c[i]), int(k[i]), float(t[i])) for i in range(nb)]

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.ba_sized_batch_destroy(self._h)
            self._h = None

    __del__ = close


class ProfileBatchAligner(BatchAligner):
    """Sequence-to-profile batch: pair p aligns the amino-acid query pool[q_off[p]:+q_len[p]] to profiles[p]
    (Block::align_profile over many pairs, examples/pssm_bench.rs:86-103)."""

    def __init__(self, profiles, size, x_drop: int, mode: int, pool, q_off, q_len):
        L = lib()
        self.n = len(q_len)
        self.mode = mode
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
        q_len = np.ascontiguousarray(q_len, dtype=np.uint32)
        natives = [_NativeProfile(p) for p in profiles]
        arr = (C.c_void_p * self.n)(*[x._h for x in natives])
        self._h = L.ba_batch_create_profile(arr, _size(size), x_drop, mode, pool.ctypes.data, q_off.ctypes.data, q_len.ctypes.data, self.n)
        if not self._h:
            raise RuntimeError(last_error())


def batch_align_exp(matrix, gaps, size, x_drop: int, target_score: int, mode: int, pool, q_off, q_len, r_off, r_len):
    """Block::align_exp over a batch (scan_block.rs:884-902) -> (score, query_idx, reference_idx, reached_min) arrays;
    reached_min[p] = 0 where the reference returns None."""
    n = len(q_len)
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    q_off = np.ascontiguousarray(q_off, dtype=np.uint64); r_off = np.ascontiguousarray(r_off, dtype=np.uint64)
    q_len = np.ascontiguousarray(q_len, dtype=np.uint32); r_len = np.ascontiguousarray(r_len, dtype=np.uint32)
    raw = matrix.raw()
    g = GapsC(gaps.open, gaps.extend) if isinstance(gaps, S.Gaps) else GapsC(gaps[0], gaps[1])
    res = (AlignResultC * n)()
    reached = np.zeros(n, np.uint64)
    if lib().block_batch_align_exp(matrix.KIND, raw.ctypes.data, g, _size(size), x_drop, target_score, mode, pool.ctypes.data,
                                   q_off.ctypes.data, q_len.ctypes.data, r_off.ctypes.data, r_len.ctypes.data, n, res, reached.ctypes.data):
        raise RuntimeError(last_error())
    return _unpack_results(res, n) + (reached,)


def batch_align_profile_exp(profiles, size, x_drop: int, target_score: int, mode: int, pool, q_off, q_len):
    """Block::align_profile_exp over a batch (scan_block.rs:974-992)."""
    n = len(q_len)
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    q_off = np.ascontiguousarray(q_off, dtype=np.uint64); q_len = np.ascontiguousarray(q_len, dtype=np.uint32)
    natives = [_NativeProfile(p) for p in profiles]
    arr = (C.c_void_p * n)(*[x._h for x in natives])
    res = (AlignResultC * n)()
    reached = np.zeros(n, np.uint64)
    if lib().block_batch_align_profile_exp(arr, _size(size), x_drop, target_score, mode, pool.ctypes.data, q_off.ctypes.data,
                                           q_len.ctypes.data, n, res, reached.ctypes.data):
        raise RuntimeError(last_error())
    return _unpack_results(res, n) + (reached,)


def _unpack_results(res, n):
    a = np.frombuffer(res, dtype=np.dtype([("score", np.int32), ("_pad", np.int32), ("qi", np.uint64), ("ri", np.uint64)]), count=n)
    return a["score"].copy(), a["qi"].copy(), a["ri"].copy()


def runs_to_string(runs) -> str:
    return "".join(f"{int(x) >> 4}{OP_CHARS[int(x) & 15]}" for x in runs)
