"""A second, independent reading of the reference's driver -- test infrastructure, never imported by the product.

Transcribed directly from /root/reference/src/scan_block.rs (align 847-878, align_core 94-595, the border moves 1003-1061, place_block
1083-1228, the Trace stack 1361-1462, cigar_core 1482-1672, blocks 1676-1691) and src/avx2.rs (the 16-lane semantics: saturating adds, the
prefix scan of 297-338 with its zero shift-in inside 8-lane halves, simd_hargmax 266-274), WITHOUT looking at oracle/ -- so that a
transcription slip in oracle/block_aligner_oracle.hpp is no longer invisible: tests/test_driver_model.py asserts oracle == model on
thousands of random pairs with forced grows and shrinks. Round 5: sequence-to-profile alignment too (Block::align_profile 942-968,
place_block_profile_right / _down 612-783 -- two different fill functions with position-specific gap_open_C / gap_close_C / gap_open_R --
over the AAProfile accessors of src/scores.rs:452-715), again written from the Rust source alone.

Deliberately different in form from the oracle: one numpy column at a time over the whole rectangle height (the 16-lane vectors are rows of
a [vectors, 16] array), the trace kept as a Python list of per-rectangle cell arrays (restore_ckpt = truncating the list), plain Python
control flow. It cannot replace the crate (see tests/test_crate_golden.py for that), it is a second witness.
"""
from __future__ import annotations

import numpy as np

L = 16          # avx2.rs:10
STEP = 8        # scan_block.rs:787
X_DROP_ITER = 2
SHRINK_SUFFIX_LEN = STEP // 4
ZERO = 1 << 14  # avx2.rs:15
MIN = 0         # avx2.rs:16
I16_MIN, I16_MAX = -32768, 32767
RIGHT, DOWN, GROW = 0, 1, 2
OP_M, OP_EQ, OP_X, OP_I, OP_D = 1, 2, 3, 4, 5
OPS = " M=XID"


def sat(x):
    """_mm256_adds_epi16 / subs_epi16 on values held in wider integers."""
    return np.clip(x, I16_MIN, I16_MAX)


def clamp(x: int) -> int:   # scan_block.rs:1703-1706
    return min(max(x, I16_MIN), I16_MAX)


def wrap16(x: int) -> int:
    """A 16-bit lane after a wrapping operation (_mm256_slli_epi16)."""
    return ((x + 32768) & 0xffff) - 32768


def sllz(v, n):
    """simd_sllz_i16!: _mm256_slli_si256 shifts each 128-bit half (8 lanes) on its own, zeros come in at lanes 0 and 8 (avx2.rs:148-160).
    v: [..., 16]."""
    out = np.zeros_like(v)
    out[..., n:8] = v[..., 0:8 - n]
    out[..., 8 + n:16] = v[..., 8:16 - n]
    return out


def prefix_scan(r, g):
    """simd_prefix_scan_i16 (avx2.rs:297-338) on [..., 16] arrays; g = gap extend."""
    consts = np.array([(k % 8 + 1) * g for k in range(L)], np.int64)   # get_prefix_scan_consts' second value (avx2.rs:283-295)
    s1 = np.maximum(r, sat(sllz(r, 1) + g))
    s2 = np.maximum(s1, sat(sllz(s1, 2) + wrap16(g << 1)))
    s4 = np.maximum(s2, sat(sllz(s2, 4) + wrap16(g << 2)))
    # correct1: lower half = lanes 0..3 of shift4 twice, upper half = lane 7 of shift4 (shufflehi + permute4x64 0b01010000)
    c = np.empty_like(s4)
    c[..., 0:4] = s4[..., 0:4]; c[..., 4:8] = s4[..., 0:4]; c[..., 8:16] = s4[..., 7:8]
    return np.maximum(s4, sat(c + consts))


class Matrix:
    """Scores as Matrix::get_scores yields them (scores.rs:116-127, 199-209, 258-267) from the raw tables of block_aligner_amd.scores."""

    def __init__(self, m):
        if m is None:           # a profile's query: AAProfile::convert_char / NULL (scores.rs:483,647-651) are the amino-acid matrix's
            self.kind = 0; self.raw = None
        else:
            self.kind = m.KIND
            self.raw = np.asarray(m.raw(), np.int64).ravel()
        self.null = {0: ord("A") + 26, 1: ord("Z"), 2: 0}[self.kind]

    def convert(self, b: bytes) -> np.ndarray:   # convert_char: scores.rs:130-134, 212-216, 270-272
        a = np.frombuffer(bytes(b), np.uint8).astype(np.int64)
        if self.kind == 2:
            return a
        a = np.where((a >= 97) & (a <= 122), a - 32, a)
        return a - 65 if self.kind == 0 else a

    def null_conv(self) -> int:
        return self.null - 65 if self.kind == 0 else self.null

    def scores(self, c: int, v: np.ndarray) -> np.ndarray:
        if self.kind == 0:      # two 16-byte tables of row c, picked by bit 4 of the byte (halfsimd_lookup2_i16)
            return self.raw[c * 32 + (v & 31)]
        if self.kind == 1:      # row (c & 7), entry (v & 15) (halfsimd_lookup1_i16 over a 16-byte row)
            return self.raw[(c & 7) * 16 + (v & 15)]
        return np.where(v == c, self.raw[0], self.raw[1])


class Rect:
    """One add_block entry with the trace words its place_block wrote (scan_block.rs:1428-1462)."""
    __slots__ = ("i", "j", "w", "h", "right", "t", "t2", "z")

    def __init__(self, i, j, w, h, right):
        self.i, self.j, self.w, self.h, self.right = i, j, w, h, right
        n = w if right else h          # columns of the fill
        m = h if right else w          # cells along the vectors
        self.t = np.zeros((n, m), np.uint8); self.t2 = np.zeros((n, m), np.uint8); self.z = np.zeros((n, m), np.uint8)


class Model:
    def __init__(self, trace=False, x_drop=False, local_start=False, free_query_start_gaps=False, free_query_end_gaps=False):
        self.TRACE, self.X_DROP, self.LOCAL, self.FQS, self.FQE = trace, x_drop, local_start, free_query_start_gaps, free_query_end_gaps

    # ------------------------------------------------------------------ Block::align (scan_block.rs:847-878)
    def align(self, q: bytes, r: bytes, matrix, gaps, size, x_drop=0):
        go, ge = gaps
        assert go < 0 and ge < 0, "Gap costs must be negative!"
        assert go < ge, "Gap open must cost more than gap extend!"
        min_size = max(size[0], L); max_size = max(size[1], L)
        assert min_size < 65535 and max_size < 65535
        assert min_size & (min_size - 1) == 0 and max_size & (max_size - 1) == 0
        if self.X_DROP:
            assert x_drop >= 0
        assert not (self.LOCAL and self.FQS) and not (self.X_DROP and self.FQE)
        assert not self.FQE or min_size > len(q), "Min block size must be larger than the query length for FREE_QUERY_END_GAPS!"
        self.m = Matrix(matrix)
        pad = np.full(max_size, self.m.null_conv(), np.int64)
        self.q = np.concatenate([[self.m.null_conv()], self.m.convert(q), pad])      # PaddedBytes (scan_block.rs:1798-1836)
        self.r = np.concatenate([[self.m.null_conv()], self.m.convert(r), pad])
        self.qlen, self.rlen = len(q), len(r)
        self.go, self.ge, self.x_drop = go, ge, x_drop
        self.min_size, self.max_size = min_size, max_size
        # Allocated::clear (scan_block.rs:1322-1339)
        z = lambda n: np.full(n, MIN, np.int64)
        self.D_col, self.C_col, self.D_row, self.R_row = z(max_size), z(max_size), z(max_size), z(max_size)
        self.ck = [z(max_size) for _ in range(4)]
        self.temp1, self.temp2 = z(L), z(L)
        self.stack: list[Rect] = []
        self.ckpt_blocks = 0
        self.cells = 0
        # place_block serves both directions, the sequences swapped by the caller (scan_block.rs:160-176, 212-228)
        self._right_fn = lambda *a: self._place(self.q, self.qlen, self.r, self.rlen, *a, True)
        self._down_fn = lambda *a: self._place(self.r, self.rlen, self.q, self.qlen, *a, False)
        self._core()
        return dict(score=self.score, query_idx=self.qi, reference_idx=self.ri, cells=self.cells)

    # ------------------------------------------------------------------ Block::align_profile (scan_block.rs:942-968)
    def align_profile(self, q: bytes, profile, size, x_drop=0):
        """profile: block_aligner_amd.scores.AAProfile (only its tables are read: pos_aa[position][residue], the three per-position gap
        arrays, gap_extend, str_len -- the reference keeps a transposed copy aa_pos for the down fill, scores.rs:455,524-529: same numbers)."""
        assert profile.get_gap_extend() < 0, "Gap extend cost must be negative!"
        min_size = max(size[0], L); max_size = max(size[1], L)
        assert min_size < 65535 and max_size < 65535
        assert min_size & (min_size - 1) == 0 and max_size & (max_size - 1) == 0
        if self.X_DROP:
            assert x_drop >= 0
        assert not (self.LOCAL and self.FQS) and not (self.X_DROP and self.FQE)
        assert not self.FQE or min_size > len(q), "Min block size must be larger than the query length for FREE_QUERY_END_GAPS!"
        self.m = Matrix(None)
        pad = np.full(max_size, self.m.null_conv(), np.int64)
        self.q = np.concatenate([[self.m.null_conv()], self.m.convert(q), pad])
        self.r = None
        self.qlen, self.rlen = len(q), profile.len()
        self.P_aa = np.asarray(profile.pos_aa, np.int64)                       # [position][residue 0..31]
        self.P_goC = np.asarray(profile.pos_gap_open_C, np.int64); self.P_clC = np.asarray(profile.pos_gap_close_C, np.int64)
        self.P_goR = np.asarray(profile.pos_gap_open_R, np.int64)
        assert self.P_aa.shape[0] >= self.rlen + max_size + 1, "the profile must be padded for the maximum block size (scores.rs:485)"
        self.ge = int(profile.get_gap_extend()); self.go = None; self.x_drop = x_drop
        self.min_size, self.max_size = min_size, max_size
        z = lambda n: np.full(n, MIN, np.int64)
        self.D_col, self.C_col, self.D_row, self.R_row = z(max_size), z(max_size), z(max_size), z(max_size)
        self.ck = [z(max_size) for _ in range(4)]
        self.temp1, self.temp2 = z(L), z(L)
        self.stack = []
        self.ckpt_blocks = 0
        self.cells = 0
        self._right_fn = lambda *a: self._place_profile(True, *a)
        self._down_fn = lambda *a: self._place_profile(False, *a)
        self._core()
        return dict(score=self.score, query_idx=self.qi, reference_idx=self.ri, cells=self.cells)

    # ------------------------------------------------------------------ border moves (scan_block.rs:1003-1061)
    @staticmethod
    def _just_offset(bs, b1, b2, off_add):
        b1[:bs] = sat(b1[:bs] + off_add); b2[:bs] = sat(b2[:bs] + off_add)

    @staticmethod
    def _prefix_max(b) -> int:
        return int(b[:STEP].max())

    @staticmethod
    def _suffix_max(b, n) -> int:
        return int(b[n - SHRINK_SUFFIX_LEN:n].max())

    @staticmethod
    def _shift_and_offset(bs, b1, b2, t1, t2, off_add) -> int:
        corner = int(sat(b1[STEP - 1] + off_add))
        b1[:bs - STEP] = sat(b1[STEP:bs] + off_add); b2[:bs - STEP] = sat(b2[STEP:bs] + off_add)
        b1[bs - STEP:bs] = t1[:STEP]; b2[bs - STEP:bs] = t2[:STEP]
        return corner

    # ------------------------------------------------------------------ place_block (scan_block.rs:1083-1228)
    def _place(self, seqv, lenv, seqc, lenc, start_i, start_j, width, height, Dc, Cc, Dr, Rr, corner, rel_zero, right):
        """seqv / lenv: the sequence along the vectors (the function's `query`), seqc / lenc: the one along the columns. Dc, Cc: views of
        the vector-axis border pair; Dr, Rr: where column j's last cell goes. Returns (D_max, D_argmax_i, D_argmax_j) as 16-lane arrays."""
        go, ge = self.go, self.ge
        D_max = np.full(L, MIN, np.int64); am_i = np.zeros(L, np.int64); am_j = np.zeros(L, np.int64)
        if width == 0 or height == 0:
            return D_max, am_i, am_j
        nv = height // L
        gap_all = np.array([(k + 1) * ge for k in range(L)], np.int64)                   # get_prefix_scan_consts' first value
        rect = self.stack[-1] if self.TRACE else None
        vb = seqv[start_i:start_i + height]
        row_of_vec = start_i + np.arange(nv) * L
        for j in range(width):
            c = int(seqc[start_j + j])
            D10 = Dc[:height].copy(); C10 = Cc[:height]
            D00 = np.empty(height, np.int64); D00[0] = corner; D00[1:] = D10[:-1]        # simd_sl_i16!(D10, D_corner, 1) along the column
            corner = MIN
            D11 = sat(D00 + self.m.scores(c, vb))
            if (not self.LOCAL and start_i == 0 and start_j + j == 0) or (self.FQS and right and start_i == 0):
                D11[0] = rel_zero
            if self.LOCAL:
                D11 = np.maximum(D11, rel_zero)
            C11_open = sat(D10 + go)
            C11 = np.maximum(sat(C10 + ge), C11_open)
            D11 = np.maximum(D11, C11)
            D11_open = sat(D11 + clamp(go - ge))
            R11 = prefix_scan(D11_open.reshape(nv, L), ge)
            last = MIN                                                                    # R01 starts at MIN
            for v in range(nv):                                                           # simd_broadcasthi_i16(R01) + gap_extend_all
                R11[v] = np.maximum(R11[v], sat(last + gap_all))
                last = int(R11[v, L - 1])
            R11 = R11.reshape(height)
            D11 = np.maximum(D11, R11)
            if self.TRACE:
                tR = (R11 == D11_open)
                tR_prev = np.empty(height, bool); tR_prev[0] = False; tR_prev[1:] = tR[:-1]   # simd_sl_i16!(temp_trace_R, prev_trace_R, 1)
                rect.t[j] = (D11 == C11).astype(np.uint8) | ((D11 == R11).astype(np.uint8) << 1)
                rect.t2[j] = (C11 == C11_open).astype(np.uint8) | (tR_prev.astype(np.uint8) << 1)
                if self.LOCAL:
                    rect.z[j] = (D11 == rel_zero)
            d2 = D11.reshape(nv, L)
            acc = np.maximum.accumulate(np.vstack([D_max[None, :], d2]), axis=0)[1:]    # D_max after each vector of the column
            track = np.ones(nv, bool) if self.X_DROP else ((row_of_vec + L > lenv) if self.FQE else np.zeros(nv, bool))
            if track.any():
                hit = (acc == d2) & track[:, None]
                for k in range(L):
                    vs = np.nonzero(hit[:, k])[0]
                    if vs.size:
                        am_i[k] = int(vs[-1]) * L; am_j[k] = j
            D_max = acc[-1].copy()
            Dc[:height] = D11; Cc[:height] = C11
            Dr[j] = D11[height - 1]; Rr[j] = R11[height - 1]
            self.cells += height
            if not self.X_DROP and not self.FQE and start_i + height > lenv and start_j + j >= lenc:
                break                                                                     # (the rest of the rectangle's trace is never read)
        return D_max, am_i, am_j

    # ------------------------------------------------------------------ place_block_profile_right / _down (scan_block.rs:612-783)
    def _place_profile(self, right, start_i, start_j, width, height, Dc, Cc, Dr, Rr, corner, rel_zero):
        """right: vectors along the query, one profile position per column (per-column scalars, scores through the position's row);
        down: vectors along the profile (per-cell gap vectors with the roles of C and R exchanged, scores of the column's residue)."""
        ge = self.ge
        lenv, lenc = (self.qlen, self.rlen) if right else (self.rlen, self.qlen)      # $query.len() / $reference.len() of the instantiation
        D_max = np.full(L, MIN, np.int64); am_i = np.zeros(L, np.int64); am_j = np.zeros(L, np.int64)
        if width == 0 or height == 0:
            return D_max, am_i, am_j
        nv = height // L
        gap_all = np.array([(k + 1) * ge for k in range(L)], np.int64)
        rect = self.stack[-1] if self.TRACE else None
        row_of_vec = start_i + np.arange(nv) * L
        rows = start_i + np.arange(height)
        clC = 0; clR = np.zeros(height, np.int64)                                          # simd_set1_i16(MIN) until assigned (640-643)
        for j in range(width):
            if right:                                                                     # 658-663
                idx = start_j + j
                goC, clC, goR = int(self.P_goC[idx]), int(self.P_clC[idx]), int(self.P_goR[idx])
                scores = self.P_aa[idx, self.q[rows] & 31]                                # get_scores_pos: the position's 32-entry row, looked up by residue
            else:                                                                         # 671-676
                goC, goR, clR = self.P_goR[rows], self.P_goC[rows], self.P_clC[rows]      # get_gap_open_down_R / _down_C / get_gap_close_down_C
                scores = self.P_aa[rows, int(self.q[start_j + j])]                        # get_scores_aa: residue c's scores at positions idx ..
            D10 = Dc[:height].copy(); C10 = Cc[:height]
            D00 = np.empty(height, np.int64); D00[0] = corner; D00[1:] = D10[:-1]
            corner = MIN
            D11 = sat(D00 + scores)
            if (not self.LOCAL and start_i == 0 and start_j + j == 0) or (self.FQS and right and start_i == 0):
                D11[0] = rel_zero
            if self.LOCAL:
                D11 = np.maximum(D11, rel_zero)
            C11_open = sat(D10 + sat(goC + ge))                                            # 692
            C11 = np.maximum(sat(C10 + ge), C11_open)
            C11_end = sat(C11 + clC) if right else C11                                     # 694
            D11 = np.maximum(D11, C11_end)
            D11_open = sat(D11 + goR)                                                      # 698
            R11 = prefix_scan(D11_open.reshape(nv, L), ge)
            last = MIN
            for v in range(nv):
                R11[v] = np.maximum(R11[v], sat(last + gap_all))
                last = int(R11[v, L - 1])
            R11 = R11.reshape(height)
            R11_end = R11 if right else sat(R11 + clR)                                     # 704
            D11 = np.maximum(D11, R11_end)
            if self.TRACE:
                tR = (R11 == D11_open)
                tR_prev = np.empty(height, bool); tR_prev[0] = False; tR_prev[1:] = tR[:-1]
                rect.t[j] = (D11 == C11_end).astype(np.uint8) | ((D11 == R11_end).astype(np.uint8) << 1)
                rect.t2[j] = (C11 == C11_open).astype(np.uint8) | (tR_prev.astype(np.uint8) << 1)
                if self.LOCAL:
                    rect.z[j] = (D11 == rel_zero)
            d2 = D11.reshape(nv, L)
            acc = np.maximum.accumulate(np.vstack([D_max[None, :], d2]), axis=0)[1:]
            track = np.ones(nv, bool) if self.X_DROP else ((row_of_vec + L > lenv) if self.FQE else np.zeros(nv, bool))
            if track.any():
                hit = (acc == d2) & track[:, None]
                for k in range(L):
                    vs = np.nonzero(hit[:, k])[0]
                    if vs.size:
                        am_i[k] = int(vs[-1]) * L; am_j[k] = j
            D_max = acc[-1].copy()
            Dc[:height] = D11; Cc[:height] = C11                                           # (C11, not C11_end; R_row takes R11, not R11_end: 763-768)
            Dr[j] = D11[height - 1]; Rr[j] = R11[height - 1]
            self.cells += height
            if not self.X_DROP and not self.FQE and start_i + height > lenv and start_j + j >= lenc:
                break
        return D_max, am_i, am_j

    def _add_block(self, i, j, w, h, right):
        self.stack.append(Rect(i, j, w, h, right))

    # ------------------------------------------------------------------ align_core (scan_block.rs:94-595)
    def _core(self):
        qlen, rlen = self.qlen, self.rlen
        best_max = 0; best_i = 0; best_j = 0
        prev_dir = GROW; d = GROW
        prev_size = 0; bs = self.min_size
        off = 0; off_max = 0
        y_drop_iter = 0; x_drop_iter = 0
        si = 0; sj = 0
        i_ck = 0; j_ck = 0; off_ck = 0
        D_corner = MIN
        Dc, Cc, Dr, Rr = self.D_col, self.C_col, self.D_row, self.R_row
        t1, t2 = self.temp1, self.temp2
        while True:
            prev_off = off
            g_max = np.full(L, MIN, np.int64); g_ai = np.zeros(L, np.int64); g_aj = np.zeros(L, np.int64)
            if d == RIGHT:
                off = off_max
                off_add = clamp(prev_off - off)
                if self.TRACE:
                    self._add_block(si, sj + bs - STEP, STEP, bs, True)
                self._just_offset(bs, Dc, Cc, off_add)
                Dm, ai, aj = self._right_fn(si, sj + bs - STEP, STEP, bs, Dc, Cc, t1, t2,
                                            int(sat(D_corner + off_add)) if prev_dir == DOWN else MIN, clamp(-off + ZERO))
                right_max = self._prefix_max(Dc)
                D_corner = self._shift_and_offset(bs, Dr, Rr, t1, t2, off_add)
                down_max = self._prefix_max(Dr)
            elif d == DOWN:
                off = off_max
                off_add = clamp(prev_off - off)
                if self.TRACE:
                    self._add_block(si + bs - STEP, sj, bs, STEP, False)
                self._just_offset(bs, Dr, Rr, off_add)
                Dm, ai, aj = self._down_fn(sj, si + bs - STEP, STEP, bs, Dr, Rr, t1, t2,
                                           int(sat(D_corner + off_add)) if prev_dir == RIGHT else MIN, clamp(-off + ZERO))
                down_max = self._prefix_max(Dr)
                D_corner = self._shift_and_offset(bs, Dc, Cc, t1, t2, off_add)
                right_max = self._prefix_max(Dc)
            else:
                D_corner = MIN
                grow_step = bs - prev_size
                if self.TRACE:
                    self._add_block(si + prev_size, sj, prev_size, grow_step, False)
                g_max, g_ai, g_aj = self._down_fn(sj, si + prev_size, grow_step, prev_size, Dr, Rr,
                                                  Dc[prev_size:], Cc[prev_size:], MIN, clamp(-off + ZERO))
                if self.TRACE:
                    self._add_block(si, sj + prev_size, grow_step, bs, True)
                Dm, ai, aj = self._right_fn(si, sj + prev_size, grow_step, bs, Dc, Cc,
                                            Dr[prev_size:], Rr[prev_size:], MIN, clamp(-off + ZERO))
                right_max = self._prefix_max(Dc); down_max = self._prefix_max(Dr)
                for a, b in zip(self.ck, (Dc, Cc, Dr, Rr)):
                    a[:bs] = b[:bs]
                self.ckpt_blocks = len(self.stack)
            prev_dir = d
            D_max_max = int(Dm[qlen % L]) if self.FQE else int(Dm.max())
            grow_max = int(g_max.max())
            mx = max(D_max_max, grow_max)
            off_max = off + mx - ZERO
            y_drop_iter += 1
            grow_no_max = d == GROW
            if off_max > best_max:
                if self.FQE:
                    idx_j = int(aj[qlen % L])
                    best_i = qlen
                    assert d != DOWN
                    best_j = sj + (bs - STEP) + idx_j if d == RIGHT else sj + prev_size + idx_j
                if self.X_DROP:
                    lane = int(np.nonzero(Dm == D_max_max)[0][0])                         # simd_hargmax_i16: the first lane at the maximum
                    idx_i, idx_j = int(ai[lane]), int(aj[lane])
                    rr, cc = idx_i + lane, (bs - STEP) + idx_j
                    if d == RIGHT:
                        best_i, best_j = si + rr, sj + cc
                    elif d == DOWN:
                        best_i, best_j = si + cc, sj + rr
                    elif D_max_max >= grow_max:
                        best_i, best_j = si + idx_i + lane, sj + prev_size + idx_j
                    else:
                        lane = int(np.nonzero(g_max == grow_max)[0][0])
                        best_i, best_j = si + prev_size + int(g_aj[lane]), sj + int(g_ai[lane]) + lane
                if bs < self.max_size:
                    i_ck, j_ck, off_ck = si, sj, off
                    for a, b in zip(self.ck, (Dc, Cc, Dr, Rr)):
                        a[:bs] = b[:bs]
                    self.ckpt_blocks = len(self.stack)
                    grow_no_max = False
                best_max = off_max
                y_drop_iter = 0
            if self.X_DROP:
                if off_max < best_max - self.x_drop:
                    if x_drop_iter < X_DROP_ITER - 1:
                        x_drop_iter += 1
                    else:
                        break
                else:
                    x_drop_iter = 0
            if si + bs > qlen and sj + bs > rlen:
                break
            if sj + bs > rlen:
                si += STEP; d = DOWN
                continue
            if si + bs > qlen:
                sj += STEP; d = RIGHT
                continue
            next_size = bs * 2
            if next_size <= self.max_size and (y_drop_iter > bs // STEP - 1 or grow_no_max):
                prev_size = bs; bs = next_size; d = GROW
                si, sj, off = i_ck, j_ck, off_ck
                for a, b in zip(self.ck, (Dc, Cc, Dr, Rr)):
                    b[:prev_size] = a[:prev_size]
                del self.stack[self.ckpt_blocks:]                                          # Trace::restore_ckpt
                y_drop_iter = 0
                continue
            if bs > self.min_size and y_drop_iter == 0:
                if max(self._suffix_max(Dr, bs), self._suffix_max(Dc, bs)) >= mx:
                    prev_dir = GROW
                    bs //= 2
                    for b in (Dc, Cc, Dr, Rr):
                        b[:bs] = b[bs:2 * bs].copy()
                    si += bs; sj += bs
                    i_ck, j_ck, off_ck = si, sj, off
                    for a, b in zip(self.ck, (Dc, Cc, Dr, Rr)):
                        a[:bs] = b[:bs]
                    right_max = self._prefix_max(Dc); down_max = self._prefix_max(Dr)
                    self.ckpt_blocks = len(self.stack)
                    y_drop_iter = 0
            if down_max > right_max:
                si += STEP; d = DOWN
            else:
                sj += STEP; d = RIGHT
        self.end_block_size = bs
        if self.X_DROP or self.FQE:
            self.score, self.qi, self.ri = best_max, best_i, best_j
        else:
            if d == DOWN:
                self.score = off + int(Dr[rlen - sj]) - ZERO
            else:
                self.score = off + int(Dc[qlen - si]) - ZERO
            self.qi, self.ri = qlen, rlen

    # ------------------------------------------------------------------ Trace::blocks (scan_block.rs:1676-1691)
    def blocks(self):
        return [(b.i, b.j, b.w, b.h) for b in self.stack]

    # ------------------------------------------------------------------ cigar_core (scan_block.rs:1482-1672)
    @staticmethod
    def _lut(right: bool, t: int, t2: int, table: int):
        """OP_LUT (scan_block.rs:1518-1568): (op, di, dj, next table); tables D = 0, C = 1, R = 2."""
        D, C, R = 0, 1, 2
        if right:
            if table == C:
                return (OP_D, 0, 1, C) if t2 in (0, 2) else (OP_D, 0, 1, D)
            if table == R:
                return (OP_I, 1, 0, R) if t2 in (0, 1) else (OP_I, 1, 0, D)
            if t == 0:
                return (OP_M, 1, 1, D)
            if t in (1, 3):
                return (OP_D, 0, 1, C) if t2 in (0, 2) else (OP_D, 0, 1, D)
            return (OP_I, 1, 0, R) if t2 in (0, 1) else (OP_I, 1, 0, D)
        if table == R:
            return (OP_I, 1, 0, R) if t2 in (0, 2) else (OP_I, 1, 0, D)
        if table == C:
            return (OP_D, 0, 1, C) if t2 in (0, 1) else (OP_D, 0, 1, D)
        if t == 0:
            return (OP_M, 1, 1, D)
        if t in (1, 3):
            return (OP_I, 1, 0, R) if t2 in (0, 2) else (OP_I, 1, 0, D)
        return (OP_D, 0, 1, C) if t2 in (0, 1) else (OP_D, 0, 1, D)

    def cigar(self, i: int, j: int, eq: bool = False) -> str:
        assert i <= self.qlen and j <= self.rlen, "Traceback cigar end position must be in bounds!"
        ops = []
        table = 0
        bidx = len(self.stack)
        done = False
        while (i > 0 or j > 0) and not done:
            while True:
                bidx -= 1
                b = self.stack[bidx]
                if i >= b.i and j >= b.j:
                    break
            while i >= b.i and j >= b.j and (i > 0 or j > 0):
                if b.right and self.FQS and i == 0:
                    done = True
                    break
                ci, cj = i - b.i, j - b.j
                col, cell = (cj, ci) if b.right else (ci, cj)
                if self.LOCAL and table == 0 and b.z[col, cell]:
                    done = True
                    break
                op, di, dj, table = self._lut(b.right, int(b.t[col, cell]), int(b.t2[col, cell]), table)
                if eq and op == OP_M:
                    op = OP_EQ if self.q[i] == self.r[j] else OP_X
                i -= di; j -= dj
                ops.append(op)
        out, k = [], 0
        ops.reverse()
        while k < len(ops):
            n = k
            while n < len(ops) and ops[n] == ops[k]:
                n += 1
            out.append(f"{n - k}{OPS[ops[k]]}")
            k = n
        return "".join(out)
