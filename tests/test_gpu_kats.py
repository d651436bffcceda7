"""GPU: the reference's own known answers, driven through the C ABI exactly as the reference's tests drive the
Rust API (Block::new / align / res / trace().cigar[_eq]); scan_block.rs:1908-2120, lib.rs:8-35."""
import pytest

from block_aligner_amd import scores as S
from tests.common import check_expect, kat_matrix, kat_profile

pytestmark = pytest.mark.gpu

MCLS = {"aa": S.AAMatrix, "nuc": S.NucMatrix, "bytes": S.ByteMatrix, "profile": S.AAMatrix}


def run_kat(H, k):
    cls = MCLS[k["kind"]]
    pad = k["size"][1]
    q = H.PaddedBytes.from_bytes(k["q"].encode(), pad, cls)
    mode = set(k["mode"])
    a = H.Block(k["alloc"][0], k["alloc"][1], k["alloc"][2], **{m: True for m in mode})
    if k["kind"] == "profile":   # scan_block.rs:2122-2168
        r = None
        a.align_profile(q, kat_profile(k), tuple(k["size"]), k["x_drop"])
    else:
        r = H.PaddedBytes.from_bytes(k["r"].encode(), pad, cls)
        a.align(q, r, kat_matrix(k), S.Gaps(*k["gaps"]), tuple(k["size"]), k["x_drop"])
    res = a.res()
    out = dict(score=res.score, query_idx=res.query_idx, reference_idx=res.reference_idx)
    cig = cig_eq = None
    e = k["expect"]
    if "cigar" in e or "cigar_eq" in e:
        c = H.Cigar(res.query_idx, res.reference_idx)
        if "cigar" in e:
            a.trace().cigar(res.query_idx, res.reference_idx, c)
            cig = str(c)
        if "cigar_eq" in e:
            a.trace().cigar_eq(q, r, res.query_idx, res.reference_idx, c)
            cig_eq = str(c)
    check_expect(k["name"], e, out, cigar=cig, cigar_eq=cig_eq)


def test_reference_kats_seq_seq(hip, kats):
    n = 0
    for k in kats["align"] + kats["inferred"]:
        if k["kind"] == "profile" or not set(k["mode"]) <= {"trace", "x_drop"}:
            continue
        run_kat(hip, k)
        n += 1
    assert n >= 34


def test_reference_kats_profile(hip, kats):
    """scan_block.rs:2122-2168 (test_profile): sequence-to-profile alignment, with and without traceback."""
    ks = [k for k in kats["align"] if k["kind"] == "profile"]
    assert len(ks) == 6
    for k in ks:
        run_kat(hip, k)


def test_reference_kats_special_modes(hip, kats):
    """scan_block.rs:2171-2230: LOCAL_START, FREE_QUERY_START_GAPS and FREE_QUERY_END_GAPS."""
    ks = [k for k in kats["align"] if set(k["mode"]) & {"local_start", "free_query_start_gaps", "free_query_end_gaps"}]
    assert len(ks) == 6
    for k in ks:
        run_kat(hip, k)


def test_c_example_flow(hip):
    """c/example.c:6-78 (example1 + example2) through the AA-specific entry points of the reference header."""
    import ctypes as C
    L = hip.lib()
    L.block_new_padded_aa.restype = C.c_void_p
    L.block_new_aa_trace.restype = C.c_void_p
    L.block_new_aa_trace.argtypes = [C.c_size_t] * 3
    L.block_align_aa_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, hip.GapsC, hip.SizeRangeC, C.c_int32]
    L.block_res_aa_trace.restype = hip.AlignResultC
    L.block_res_aa_trace.argtypes = [C.c_void_p]
    L.block_cigar_aa_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    L.block_free_aa_trace.argtypes = [C.c_void_p]
    a_str, b_str = b"AAAAAAAA", b"AARAAAA"
    a = L.block_new_padded_aa(len(a_str), 32)
    b = L.block_new_padded_aa(len(b_str), 32)
    L.block_set_bytes_padded_aa(a, a_str, len(a_str), 32)
    L.block_set_bytes_padded_aa(b, b_str, len(b_str), 32)
    blk = L.block_new_aa_trace(len(a_str), len(b_str), 32)
    blosum62 = C.c_void_p.in_dll(L, "BLOSUM62")
    L.block_align_aa_trace(blk, a, b, C.addressof(blosum62), hip.GapsC(-11, -1), hip.SizeRangeC(32, 32), 0)
    res = L.block_res_aa_trace(blk)
    assert (res.score, res.query_idx, res.reference_idx) == (12, 8, 7)   # inferred (see golden file)
    cig = L.block_new_cigar(res.query_idx, res.reference_idx)
    L.block_cigar_aa_trace(blk, res.query_idx, res.reference_idx, cig)
    n = L.block_len_cigar(cig)
    ops = [L.block_get_cigar(cig, i) for i in range(n)]
    assert sum(o.len for o in ops if o.op in (1, 4)) == 8 and sum(o.len for o in ops if o.op in (1, 5)) == 7
    L.block_free_cigar(cig); L.block_free_aa_trace(blk); L.block_free_padded_aa(a); L.block_free_padded_aa(b)


def test_handle_state_is_allocated_once(hip):
    """Block::new allocates, align never does (scan_block.rs:798-805, 1280-1340): 1000 sequential align + cigar calls on one
    handle -- the serial loop every reference example runs (examples/nanopore_bench.rs:83-93) -- leave the device's free
    memory exactly where it was after the second call, and give the same answer every time."""
    import time
    import numpy as np
    from block_aligner_amd import scores as S, synth
    rng = np.random.default_rng(3)
    r = synth.rand_str(rng, 900, synth.AMINO)
    q = synth.mutate(rng, r, 120, synth.AMINO)
    qb, rb = q.astype(np.uint8).tobytes(), r.astype(np.uint8).tobytes()
    blk = hip.Block(len(qb), len(rb), 256, trace=True, x_drop=True)
    pq = hip.PaddedBytes.from_bytes(qb, 256, S.AAMatrix); pr = hip.PaddedBytes.from_bytes(rb, 256, S.AAMatrix)
    cig = hip.Cigar(len(qb), len(rb))
    first = None
    free2 = None
    t0 = None
    for it in range(1000):
        if it == 2:
            free2 = hip.device_memory()[0]
            t0 = time.perf_counter()
        blk.align(pq, pr, S.BLOSUM62, S.Gaps(-11, -1), (32, 256), 50)
        res = blk.res()
        blk.trace().cigar_eq(pq, pr, res.query_idx, res.reference_idx, cig)
        got = (res.score, res.query_idx, res.reference_idx, str(cig) if it % 100 == 0 else None)
        if first is None:
            first = got
        assert got[:3] == first[:3] and (got[3] is None or got[3] == first[3])
    per_call = (time.perf_counter() - t0) / 998 * 1e6
    assert hip.device_memory()[0] == free2
    # the floor of one call: a pair that fits one 32-cell block (upload, one launch, one 64-byte read-back)
    tq = hip.PaddedBytes.from_bytes(b"MKVLAARNDCEQ", 256, S.AAMatrix); tr = hip.PaddedBytes.from_bytes(b"MKVLARNDCEQ", 256, S.AAMatrix)
    blk.align(tq, tr, S.BLOSUM62, S.Gaps(-11, -1), (32, 256), 50)
    t1 = time.perf_counter()
    for _ in range(300):
        blk.align(tq, tr, S.BLOSUM62, S.Gaps(-11, -1), (32, 256), 50)
    floor = (time.perf_counter() - t1) / 300 * 1e6
    assert hip.device_memory()[0] == free2
    print(f"\nhandle API: {per_call:.1f} us per align + cigar_eq call pair (900 aa, block 32..256); {floor:.1f} us per align of a 12-residue pair")
    assert per_call < 5000
