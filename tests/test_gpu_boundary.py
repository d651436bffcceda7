"""Boundary functions of the reference's C API that the other GPU tests only load: executed through the C ABI and compared with
the oracle. block_set_all_aaprofile / _rev (ffi.rs:101-120, scores.rs:532-537,677-715: a whole PSSM in one call, optionally
reversed, with the shift pair the reference applies to every score) and block_set_bytes_rev_padded_{aa,nuc} (ffi.rs:246-251,
scan_block.rs:1824-1850: a sequence stored back to front)."""
import ctypes as C

import numpy as np
import pytest

from block_aligner_amd import scores as S
from block_aligner_amd import synth
from block_aligner_amd.hip import GapsC, SizeRangeC, AlignResultC

pytestmark = pytest.mark.gpu

ORDER = b"ARNDCQEGHILKMFPSTWYVBZX"


def native_profile(L, str_len, block, ge, scores, lsh, rsh, rev, gaps):
    L.block_new_aaprofile.restype = C.c_void_p
    L.block_new_aaprofile.argtypes = [C.c_size_t, C.c_size_t, C.c_int8]
    p = L.block_new_aaprofile(str_len, block, ge)
    fn = L.block_set_all_rev_aaprofile if rev else L.block_set_all_aaprofile
    fn.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]
    fn(p, ORDER, len(ORDER), scores.ctypes.data, scores.size, lsh, rsh)
    for name, g in zip(("open_C", "close_C", "open_R"), gaps):
        f = getattr(L, f"block_set_all_gap_{name}_aaprofile")
        f.argtypes = [C.c_void_p, C.c_int8]
        f(p, g)
    return p


@pytest.mark.parametrize("rev", [False, True])
@pytest.mark.parametrize("shifts", [(0, 0), (1, 0), (2, 3)])
def test_set_all_aaprofile(hip, oracle, rev, shifts):
    L = hip.lib()
    rng = np.random.default_rng(42 + rev + 10 * shifts[0])
    str_len, block = 137, 64
    scores = rng.integers(-20, 21, size=str_len * len(ORDER)).astype(np.int8)
    gaps = (-9, -1, -7)
    p = native_profile(L, str_len, block, -1, scores, shifts[0], shifts[1], rev, gaps)
    mirror = S.AAProfile(str_len, block, -1)
    (mirror.set_all_rev if rev else mirror.set_all)(ORDER, scores, shifts[0], shifts[1])
    mirror.set_all_gap_open_C(gaps[0]); mirror.set_all_gap_close_C(gaps[1]); mirror.set_all_gap_open_R(gaps[2])
    # every cell the call wrote, read back through the ABI
    L.block_get_aaprofile.restype = C.c_int8
    L.block_get_aaprofile.argtypes = [C.c_void_p, C.c_size_t, C.c_uint8]
    for i in range(0, str_len + 1):
        for b in ORDER:
            assert L.block_get_aaprofile(p, i, b) == int(mirror.pos_aa[i, b - 65]), (i, chr(b))
    # hand-checked corner: position 1 (or the last one, reversed) holds the first row of scores, shifted as i8
    first = ((scores[: len(ORDER)].astype(np.int16) << shifts[0]).astype(np.int8) >> shifts[1]).astype(np.int8)
    row = str_len if rev else 1
    assert [L.block_get_aaprofile(p, row, b) for b in ORDER] == [int(x) for x in first]
    # and an alignment against it, with traceback, equals the oracle's on the mirrored profile
    q = synth.rand_str(rng, 120, synth.AMINO).tobytes()
    pq = hip.PaddedBytes.from_bytes(q, block, S.AAMatrix)
    L.block_new_aa_trace.restype = C.c_void_p
    L.block_new_aa_trace.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t]
    blk = L.block_new_aa_trace(len(q), str_len, block)
    L.block_align_profile_aa_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, SizeRangeC, C.c_int32]
    L.block_align_profile_aa_trace(blk, pq._h, p, SizeRangeC(32, block), 0)
    L.block_res_aa_trace.restype = AlignResultC
    L.block_res_aa_trace.argtypes = [C.c_void_p]
    r = L.block_res_aa_trace(blk)
    ref = oracle.align_profile(q, mirror, (32, block), 0, ("trace",))
    assert (r.score, r.query_idx, r.reference_idx) == (ref["score"], ref["query_idx"], ref["reference_idx"])
    cg = hip.Cigar(r.query_idx, r.reference_idx)
    L.block_cigar_aa_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    L.block_cigar_aa_trace(blk, r.query_idx, r.reference_idx, cg._h)
    assert str(cg) == ref["cigar"]
    L.block_free_aa_trace.argtypes = [C.c_void_p]
    L.block_free_aa_trace(blk)
    L.block_free_aaprofile.argtypes = [C.c_void_p]
    L.block_free_aaprofile(p)


@pytest.mark.parametrize("kind", ["aa", "nuc"])
def test_set_bytes_rev_padded(hip, oracle, kind):
    """A sequence stored back to front aligns like its reversal (scan_block.rs:1824-1831)."""
    rng = np.random.default_rng(7)
    alpha, cls, matrix, gaps = (synth.AMINO, S.AAMatrix, S.BLOSUM62, (-11, -1)) if kind == "aa" else (synth.DNA, S.NucMatrix, S.NucMatrix.new_simple(2, -3), (-5, -1))
    r = synth.rand_str(rng, 400, alpha)
    q = synth.mutate(rng, r, 40, alpha)
    qb, rb = q.tobytes(), r.tobytes()
    block = 128
    pq = hip.PaddedBytes(len(qb), block, cls); pr = hip.PaddedBytes(len(rb), block, cls)
    pq.set_bytes_rev(qb, block); pr.set_bytes_rev(rb, block)
    assert pq.len() == len(qb)
    a = hip.Block(len(qb), len(rb), block, trace=True)
    a.align(pq, pr, matrix, S.Gaps(*gaps), (32, block), 0)
    res = a.res()
    ref = oracle.align(matrix, qb[::-1], rb[::-1], gaps, (32, block), 0, ("trace",), cigar_eq=True)
    assert (res.score, res.query_idx, res.reference_idx) == (ref["score"], ref["query_idx"], ref["reference_idx"])
    cg = hip.Cigar(res.query_idx, res.reference_idx)
    a.trace().cigar_eq(pq, pr, res.query_idx, res.reference_idx, cg)
    assert str(cg) == ref["cigar"]
    # mixed case input is folded like the forward setter's
    pq2 = hip.PaddedBytes(len(qb), block, cls)
    pq2.set_bytes_rev(qb.lower(), block)
    a.align(pq2, pr, matrix, S.Gaps(*gaps), (32, block), 0)
    assert a.res().score == ref["score"]


def test_align_padded_images_by_pointer(hip, oracle):
    """block_align_padded_generic (the Rust shim's entry, rust/src/scan_block_hip.rs): the caller's own padded images --
    [NULL] + converted bytes + NULL x block_size, scan_block.rs:1790-1812 -- by pointer; same results as the handle objects."""
    L = hip.lib()
    rng = np.random.default_rng(3)
    block = 256
    r = synth.rand_str(rng, 700, synth.DNA)
    q = synth.mutate(rng, r, 70, synth.DNA)
    m = S.NucMatrix.new_simple(2, -3)

    def image(b):   # NucMatrix: convert_char = upper case, NULL = 'Z' (scores.rs:139-216)
        return np.concatenate([[ord("Z")], np.frombuffer(bytes(b).upper(), np.uint8), np.full(block, ord("Z"))]).astype(np.uint8)

    qi, ri = image(q.tobytes()), image(r.tobytes())
    L.block_new_generic.restype = C.c_void_p
    blk = L.block_new_generic(hip.TRACE | hip.X_DROP, len(q), len(r), block)
    L.block_align_padded_generic.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, GapsC, SizeRangeC, C.c_int32]
    raw = m.raw()
    L.block_align_padded_generic(blk, 1, qi.ctypes.data, len(q), ri.ctypes.data, len(r), raw.ctypes.data, GapsC(-5, -1), SizeRangeC(32, block), 50)
    res = L.block_res_generic(blk)
    ref = oracle.align(m, q.tobytes(), r.tobytes(), (-5, -1), (32, block), 50, ("trace", "x_drop"), cigar_eq=True)
    assert (res.score, res.query_idx, res.reference_idx) == (ref["score"], ref["query_idx"], ref["reference_idx"])
    cg = hip.Cigar(res.query_idx, res.reference_idx)
    L.block_cigar_eq_generic(blk, None, None, res.query_idx, res.reference_idx, cg._h)
    assert str(cg) == ref["cigar"]
    L.block_free_generic(blk)
    # and a profile through the same kind of entry
    aq = synth.rand_str(rng, 90, synth.AMINO).tobytes()
    prof = S.AAProfile.from_bytes(synth.rand_str(rng, 110, synth.AMINO).tobytes(), 64, 2, -1, -3, 0, -3, -1)
    aimg = np.concatenate([[26], np.frombuffer(aq, np.uint8) - 65, np.full(64, 26)]).astype(np.uint8)   # AAMatrix: c - 'A', NULL = '[' - 'A' = 26
    native = hip._NativeProfile(prof)
    blk = L.block_new_generic(hip.TRACE, len(aq), prof.str_len, 64)
    L.block_align_profile_padded_generic.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, SizeRangeC, C.c_int32]
    L.block_align_profile_padded_generic(blk, aimg.ctypes.data, len(aq), native._h, SizeRangeC(32, 64), 0)
    res = L.block_res_generic(blk)
    ref = oracle.align_profile(aq, prof, (32, 64), 0, ("trace",))
    assert (res.score, res.query_idx, res.reference_idx) == (ref["score"], ref["query_idx"], ref["reference_idx"])
    L.block_free_generic(blk)


@pytest.mark.parametrize("size", [(32, 256), (128, 4096)])
def test_handle_cigars_from_the_end_and_from_interior_cells(hip, oracle, size):
    """block_cigar_* walks one path with a whole wave (ba_driver.hpp walk_wave): rectangle records 64 at a time, trace words staged
    in LDS -- or, for rectangles larger than the LDS region (a grow to 4096 cells), read from global memory. The CIGAR from the
    end cell equals the oracle's; from any other cell of the computed region it must still be a valid path to the origin
    (Trace::cigar takes any cell, scan_block.rs:1469-1480) that consumes exactly (i, j)."""
    rng = np.random.default_rng(size[1])
    ps = synth.make_pairs(1, (5000, 5001), (300, 301), 200, synth.DNA, seed=size[1], indels=2, indel_len=(600, 900))
    qb, rb = ps.query(0), ps.reference(0)
    m = S.NucMatrix.new_simple(2, -3)
    pq = hip.PaddedBytes.from_bytes(qb, size[1], S.NucMatrix); pr = hip.PaddedBytes.from_bytes(rb, size[1], S.NucMatrix)
    a = hip.Block(len(qb), len(rb), size[1], trace=True)
    a.align(pq, pr, m, S.Gaps(-5, -1), size, 0)
    res = a.res()
    ref = oracle.align(m, qb, rb, (-5, -1), size, 0, ("trace",), cigar_eq=True)
    assert (res.score, res.query_idx, res.reference_idx) == (ref["score"], ref["query_idx"], ref["reference_idx"])
    cg = hip.Cigar(res.query_idx, res.reference_idx)
    a.trace().cigar_eq(pq, pr, res.query_idx, res.reference_idx, cg)
    assert str(cg) == ref["cigar"]
    if size[1] == 4096:
        assert max(h for _, _, w, h in a.trace().blocks()) >= 2048      # the long indels really grew the block
    # interior cells of the path itself: every prefix of the alignment is again a path from that cell
    step = {"M": (1, 1), "=": (1, 1), "X": (1, 1), "I": (1, 0), "D": (0, 1)}

    def consumed(c):
        i = j = 0
        ends = []
        for op, ln in c.to_vec():
            if not op:
                continue
            di, dj = step[hip.OP_CHARS[op]]
            i += di * ln; j += dj * ln
            ends.append((i, j))
        return ends

    ends = consumed(cg)
    assert ends[-1] == (res.query_idx, res.reference_idx)
    for (ci, cj) in ends[len(ends) // 3:: max(1, len(ends) // 7)]:
        part = hip.Cigar(ci, cj)
        a.trace().cigar_eq(pq, pr, ci, cj, part)
        assert consumed(part)[-1] == (ci, cj)
        assert str(cg).startswith(str(part))      # the path below a cell of the path is the path's own prefix


@pytest.mark.parametrize("size", [(32, 32), (128, 128), (32, 1024), (2048, 2048)])
def test_whole_wave_walk_takes_diagonal_runs(hip, oracle, size):
    """Round 5: in state D the lanes of the whole-wave walk look at one cell each of the diagonal below the current one and a run of plain
    match / mismatch cells is taken in one step, cut into = / X runs from two lane masks (ba_driver.hpp walk_wave). Shapes that stress it:
    identical sequences (runs as long as a rectangle is wide: 64 lanes at (2048, 2048)), a mismatch every k-th base (= / X runs of every
    length 1 .. 17), single-base gaps between long runs, runs that end at the matrix edge; with and without = / X; walked from the end and
    from interior cells."""
    rng = np.random.default_rng(size[0] + size[1])
    base = synth.rand_str(rng, 1500, synth.DNA)
    cases = [(base, base.copy())]
    for k in (2, 3, 5, 17):
        other = base.copy()
        other[::k] = (np.searchsorted(synth.DNA, other[::k]) + 1) % 4
        other[::k] = synth.DNA[other[::k]]
        cases.append((base, other))
    gapped = np.concatenate([base[:400], base[401:900], synth.DNA[:1], base[900:]])   # one deletion, one insertion, long runs between
    cases.append((base, gapped))
    cases.append((base[:700], np.concatenate([base[:700], synth.rand_str(rng, 300, synth.DNA)])))   # the path ends on the matrix edge
    m = S.NucMatrix.new_simple(2, -3)
    for q, r in cases:
        qb, rb = q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()
        pq = hip.PaddedBytes.from_bytes(qb, size[1], S.NucMatrix); pr = hip.PaddedBytes.from_bytes(rb, size[1], S.NucMatrix)
        a = hip.Block(len(qb), len(rb), size[1], trace=True)
        a.align(pq, pr, m, S.Gaps(-5, -1), size, 0)
        res = a.res()
        for eq in (True, False):
            ref = oracle.align(m, qb, rb, (-5, -1), size, 0, ("trace",), cigar_eq=eq)
            assert (res.score, res.query_idx, res.reference_idx) == (ref["score"], ref["query_idx"], ref["reference_idx"])
            cg = hip.Cigar(res.query_idx, res.reference_idx)
            if eq:
                a.trace().cigar_eq(pq, pr, res.query_idx, res.reference_idx, cg)
            else:
                a.trace().cigar(res.query_idx, res.reference_idx, cg)
            assert str(cg) == ref["cigar"], (size, eq, str(cg)[:120], ref["cigar"][:120])
        # interior cells on the main diagonal: whatever the walk finds from there consumes exactly (i, i)
        for i in (1, 63, 64, 65, 511, 600):
            part = hip.Cigar(i, i)
            a.trace().cigar_eq(pq, pr, i, i, part)
            ci = cj = 0
            for op, ln in part.to_vec():
                if op:
                    ci += ln * (hip.OP_CHARS[op] in "M=XI"); cj += ln * (hip.OP_CHARS[op] in "M=XD")
            assert (ci, cj) == (i, i)
