"""ORACLE — TEST INFRASTRUCTURE ONLY. ctypes loader for oracle/_build/libba_oracle_{avx2,scalar}.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (block_aligner_amd) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FLAG_BITS = {"trace": 1, "x_drop": 2, "local_start": 4, "free_query_start_gaps": 8, "free_query_end_gaps": 16, "cigar_eq": 32}


def build(force: bool = False) -> None:
    """Compile both oracle libraries (g++ only; no GPU, no reference sources needed)."""
    need = force or any(not os.path.exists(os.path.join(HERE, "_build", f"libba_oracle_{b}.so")) for b in ("avx2", "scalar"))
    if need:
        subprocess.check_call(["make", "-C", HERE, "-j2"] + (["-B"] if force else []), stdout=subprocess.DEVNULL)


def flags_of(mode) -> int:
    f = 0
    for m in mode:
        f |= FLAG_BITS[m]
    return f


class Oracle:
    def __init__(self, backend: str = "avx2"):
        path = os.path.join(HERE, "_build", f"libba_oracle_{backend}.so")
        if not os.path.exists(path):
            build()
        self.lib = lib = C.CDLL(path)
        lib.ba_oracle_backend.restype = C.c_char_p
        lib.ba_oracle_last_error.restype = C.c_char_p
        lib.ba_oracle_percent_len.restype = C.c_size_t
        lib.ba_oracle_percent_len.argtypes = [C.c_size_t, C.c_float]
        lib.ba_oracle_lane_op.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ba_oracle_align.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int8, C.c_int8,
                                        C.c_size_t, C.c_size_t, C.c_int32, C.c_uint32, C.POINTER(C.c_int32),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t, C.c_void_p]
        lib.ba_oracle_align_blocks.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int8, C.c_int8,
                                               C.c_size_t, C.c_size_t, C.c_int32, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        lib.ba_oracle_align_exp.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int8, C.c_int8,
                                            C.c_size_t, C.c_size_t, C.c_int32, C.c_int32, C.c_uint32, C.POINTER(C.c_int32),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.ba_oracle_align_profile.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int8, C.c_size_t, C.c_size_t, C.c_int32, C.c_uint32, C.POINTER(C.c_int32),
                                                C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_char_p, C.c_size_t, C.c_void_p]
        lib.ba_oracle_batch_align.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                              C.c_int8, C.c_int8, C.c_size_t, C.c_size_t, C.c_int32, C.c_uint32, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        lib.ba_oracle_batch_align_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_void_p, C.c_int8, C.c_size_t, C.c_size_t, C.c_int32, C.c_uint32, C.c_int,
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
        self.backend = lib.ba_oracle_backend().decode()

    def _err(self):
        return RuntimeError(self.lib.ba_oracle_last_error().decode())

    def percent_len(self, n: int, p: float) -> int:
        return self.lib.ba_oracle_percent_len(n, p)

    def lane_op(self, op: int, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.int16)
        b = np.zeros(16, np.int16) if b is None else np.ascontiguousarray(b, dtype=np.int16)
        out = np.zeros(16, np.int16)
        rc = self.lib.ba_oracle_lane_op(op, a.ctypes.data, b.ctypes.data, out.ctypes.data)
        assert rc == 0
        return out

    def align(self, matrix, q: bytes, r: bytes, gaps, size, x_drop=0, mode=(), cigar_eq=False):
        """-> dict(score, query_idx, reference_idx, cigar, cells, steps, end_block_size, surviving_cells)."""
        raw = matrix.raw()
        flags = flags_of(mode) | (FLAG_BITS["cigar_eq"] if cigar_eq else 0)
        score, qi, ri = C.c_int32(), C.c_uint64(), C.c_uint64()
        cap = 4 * (len(q) + len(r)) + 64
        buf = C.create_string_buffer(cap)
        stats = np.zeros(4, np.uint64)
        rc = self.lib.ba_oracle_align(matrix.KIND, raw.ctypes.data, q, len(q), r, len(r), gaps[0], gaps[1], size[0], size[1],
                                      x_drop, flags, C.byref(score), C.byref(qi), C.byref(ri), buf, cap, stats.ctypes.data)
        if rc:
            raise self._err()
        return dict(score=score.value, query_idx=qi.value, reference_idx=ri.value, cigar=buf.value.decode(),
                    cells=int(stats[0]), steps=int(stats[1]), end_block_size=int(stats[2]), surviving_cells=int(stats[3]))

    def align_blocks(self, matrix, q: bytes, r: bytes, gaps, size, x_drop=0, mode=()):
        """Trace::blocks() of the alignment -> [(row, col, width, height), ...] in fill order (scan_block.rs:1676-1691)."""
        raw = matrix.raw()
        cap = (len(q) + len(r)) // 4 + 64
        rects = np.zeros(4 * cap, np.uint64)
        count = C.c_uint64()
        rc = self.lib.ba_oracle_align_blocks(matrix.KIND, raw.ctypes.data, q, len(q), r, len(r), gaps[0], gaps[1], size[0], size[1],
                                             x_drop, flags_of(mode), rects.ctypes.data, cap, C.byref(count))
        if rc:
            raise self._err()
        assert count.value <= cap
        return [tuple(int(v) for v in rects[4 * k: 4 * k + 4]) for k in range(count.value)]

    def align_exp(self, matrix, q, r, gaps, size, x_drop, target, mode=()):
        raw = matrix.raw()
        score, qi, ri, reached = C.c_int32(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        rc = self.lib.ba_oracle_align_exp(matrix.KIND, raw.ctypes.data, q, len(q), r, len(r), gaps[0], gaps[1], size[0], size[1],
                                          x_drop, target, flags_of(mode), C.byref(score), C.byref(qi), C.byref(ri), C.byref(reached))
        if rc:
            raise self._err()
        return dict(score=score.value, query_idx=qi.value, reference_idx=ri.value, reached=reached.value or None)

    def align_profile(self, q: bytes, profile, size, x_drop=0, mode=()):
        n = profile.str_len + 1
        pos_aa = np.ascontiguousarray(profile.pos_aa[:n], dtype=np.int8)
        goc = np.ascontiguousarray(profile.pos_gap_open_C[:n], dtype=np.int8)
        gcc = np.ascontiguousarray(profile.pos_gap_close_C[:n], dtype=np.int8)
        gor = np.ascontiguousarray(profile.pos_gap_open_R[:n], dtype=np.int8)
        score, qi, ri = C.c_int32(), C.c_uint64(), C.c_uint64()
        cap = 4 * (len(q) + n) + 64
        buf = C.create_string_buffer(cap)
        stats = np.zeros(4, np.uint64)
        rc = self.lib.ba_oracle_align_profile(q, len(q), profile.str_len, pos_aa.ctypes.data, goc.ctypes.data, gcc.ctypes.data,
                                              gor.ctypes.data, profile.gap_extend, size[0], size[1], x_drop, flags_of(mode),
                                              C.byref(score), C.byref(qi), C.byref(ri), buf, cap, stats.ctypes.data)
        if rc:
            raise self._err()
        return dict(score=score.value, query_idx=qi.value, reference_idx=ri.value, cigar=buf.value.decode(),
                    cells=int(stats[0]), steps=int(stats[1]), end_block_size=int(stats[2]), surviving_cells=int(stats[3]))

    def batch_align(self, matrix, pool, q_off, q_len, r_off, r_len, gaps, size, x_drop=0, mode=(), cigar_eq=False,
                    threads=1, want_cigar=None):
        """Batch over a packed byte pool. -> dict(scores, query_idx, reference_idx, cig_ops, cig_off, cig_len, cells, seconds)."""
        raw = matrix.raw()
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64); r_off = np.ascontiguousarray(r_off, dtype=np.uint64)
        q_len = np.ascontiguousarray(q_len, dtype=np.uint32); r_len = np.ascontiguousarray(r_len, dtype=np.uint32)
        n = len(q_len)
        flags = flags_of(mode) | (FLAG_BITS["cigar_eq"] if cigar_eq else 0)
        trace = "trace" in mode if want_cigar is None else want_cigar
        scores = np.zeros(n, np.int32); qi = np.zeros(n, np.uint32); ri = np.zeros(n, np.uint32)
        if trace:
            cap = q_len.astype(np.uint64) + r_len.astype(np.uint64) + 1
            cig_off = np.zeros(n, np.uint64)
            np.cumsum(cap[:-1], out=cig_off[1:])
            cig_ops = np.zeros(int(cap.sum()), np.uint32)
            cig_len = np.zeros(n, np.uint32)
            cp, co, cl = cig_ops.ctypes.data, cig_off.ctypes.data, cig_len.ctypes.data
        else:
            cig_ops = cig_off = cig_len = None
            cp = co = cl = None
        cells, secs = C.c_uint64(), C.c_double()
        rc = self.lib.ba_oracle_batch_align(matrix.KIND, raw.ctypes.data, pool.ctypes.data, q_off.ctypes.data, q_len.ctypes.data,
                                            r_off.ctypes.data, r_len.ctypes.data, n, gaps[0], gaps[1], size[0], size[1], x_drop,
                                            flags, threads, scores.ctypes.data, qi.ctypes.data, ri.ctypes.data, cp, co, cl,
                                            C.byref(cells), C.byref(secs))
        if rc:
            raise self._err()
        return dict(scores=scores, query_idx=qi, reference_idx=ri, cig_ops=cig_ops, cig_off=cig_off, cig_len=cig_len,
                    cells=cells.value, seconds=secs.value)

    def batch_align_profile(self, pool, q_off, q_len, profiles, size, x_drop=0, mode=(), threads=1):
        """Threaded batch of sequence-to-profile alignments. -> dict(scores, query_idx, reference_idx, cells (per pair), cig_ops,
        cig_off, cig_len, seconds)."""
        pool = np.ascontiguousarray(pool, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64); q_len = np.ascontiguousarray(q_len, dtype=np.uint32)
        n = len(q_len)
        p_len = np.array([p.str_len for p in profiles], np.uint32)
        rows = p_len.astype(np.uint64) + 1
        p_off = np.zeros(n, np.uint64)
        np.cumsum(rows[:-1], out=p_off[1:])
        pos_aa = np.concatenate([np.ascontiguousarray(p.pos_aa[: p.str_len + 1], dtype=np.int8) for p in profiles])
        goc = np.concatenate([np.asarray(p.pos_gap_open_C[: p.str_len + 1], dtype=np.int8) for p in profiles])
        gcc = np.concatenate([np.asarray(p.pos_gap_close_C[: p.str_len + 1], dtype=np.int8) for p in profiles])
        gor = np.concatenate([np.asarray(p.pos_gap_open_R[: p.str_len + 1], dtype=np.int8) for p in profiles])
        ge = profiles[0].gap_extend
        scores = np.zeros(n, np.int32); qi = np.zeros(n, np.uint32); ri = np.zeros(n, np.uint32); cells = np.zeros(n, np.uint64)
        trace = "trace" in mode
        if trace:
            cap = q_len.astype(np.uint64) + p_len.astype(np.uint64) + 1
            cig_off = np.zeros(n, np.uint64)
            np.cumsum(cap[:-1], out=cig_off[1:])
            cig_ops = np.zeros(int(cap.sum()), np.uint32); cig_len = np.zeros(n, np.uint32)
            cp, co, cl = cig_ops.ctypes.data, cig_off.ctypes.data, cig_len.ctypes.data
        else:
            cig_ops = cig_off = cig_len = None
            cp = co = cl = None
        secs = C.c_double()
        rc = self.lib.ba_oracle_batch_align_profile(pool.ctypes.data, q_off.ctypes.data, q_len.ctypes.data, n, pos_aa.ctypes.data, goc.ctypes.data,
                                                    gcc.ctypes.data, gor.ctypes.data, p_off.ctypes.data, p_len.ctypes.data, ge, size[0], size[1], x_drop,
                                                    flags_of(mode), threads, scores.ctypes.data, qi.ctypes.data, ri.ctypes.data, cp, co, cl,
                                                    cells.ctypes.data, C.byref(secs))
        if rc:
            raise self._err()
        return dict(scores=scores, query_idx=qi, reference_idx=ri, cells=cells, cig_ops=cig_ops, cig_off=cig_off, cig_len=cig_len, seconds=secs.value)


OPS = " M=XID"


def cigar_runs_to_string(ops: np.ndarray) -> str:
    """Packed (len << 4 | op) runs -> standard CIGAR text (cigar.rs:147-163)."""
    return "".join(f"{int(x) >> 4}{OPS[int(x) & 15]}" for x in ops)
