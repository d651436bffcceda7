"""Device-side known-answer test of the lane primitives.

The reference pins its 16-lane prefix scan with two vectors (`src/avx2.rs:469-489`); the carry from one 16-cell vector of a column to the next is
`scan_block.rs:1144-1150` (`R11 = max(prefix_scan(D11_open), broadcasthi(R01) + gap_extend_all)`, `R01 = MIN = 0` above a column's first vector).
On the GPU a column is not cut into 16-lane vectors: `k_multi` gives a lane eight cells and scans sixteen lanes (`wave_prefix_max16`, the `G` / `w0`
constants), `k_small` scans the four lanes of a quad, `k_align` / `k_quad` / the solo drivers give a lane two cells and scan 64 / 32 / 16 lanes --
each with the reference's zero-shift-in artefact folded into per-lane constants. `ba_dev_lane_scan` (development library) runs exactly those device
functions on caller-supplied columns; expected values come from the oracle's `simd_prefix_scan_i16` restatement (AVX2 intrinsics and the scalar
lane model), vector by vector, with the carry applied as the reference does.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FORMS = {0: (128, 512, "k_multi: 16 lanes x 8 cells"), 1: (32, 512, "k_small: 4 lanes x 8 cells"), 2: (128, 128, "64 lanes x 2 cells"),
         3: (64, 64, "32 lanes x 2 cells"), 4: (32, 32, "16 lanes x 2 cells")}   # form -> (cells per column, cells per wave, what)


def device_scan(hip, form, x, g):
    x = np.ascontiguousarray(x, np.int16)
    out = np.zeros_like(x)
    f = hip.lib().ba_dev_lane_scan
    f.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    f.restype = C.c_int
    assert f(form, x.ctypes.data, x.size, g, out.ctypes.data) == 0, hip.last_error()
    return out


def sat16(v):
    return np.clip(v, -32768, 32767)


def expected_columns(o, x, height, g):
    """R11 of whole columns: the reference's per-vector scan (oracle lane op 0) + its carry between the vectors of a column."""
    x = np.asarray(x, np.int16).reshape(-1, height // 16, 16)
    gap_all = sat16(np.arange(1, 17, dtype=np.int64) * g)     # get_prefix_scan_consts' first value (avx2.rs:297-310)
    out = np.zeros(x.shape, np.int64)
    for c in range(x.shape[0]):
        carry = 0                                              # R01 = MIN above the column
        for v in range(x.shape[1]):
            s = o.lane_op(0, x[c, v], [g] * 16).astype(np.int64)
            out[c, v] = np.maximum(s, sat16(carry + gap_all))
            carry = out[c, v, 15]
    return out.astype(np.int16).reshape(-1)


def columns_for(form, first_vectors):
    """Whole waves of columns whose FIRST vector is one of `first_vectors`, the rest of the column far below (so that the first vector's answer is
    the reference's own, and the rest still exercises the carry)."""
    height, per_wave, _ = FORMS[form]
    cols = []
    for v in first_vectors:
        col = np.full(height, -30000, np.int16)
        col[:16] = v
        cols.append(col)
    while (len(cols) * height) % per_wave:
        cols.append(np.zeros(height, np.int16))
    return np.concatenate(cols)


@pytest.mark.parametrize("form", sorted(FORMS))
def test_reference_prefix_scan_vectors(devlib, oracle, kats, form):
    """avx2.rs:476-486: the reference's two answers, as the first vector of a device column (inputs >= 0, so the MIN carry above the column changes
    nothing)."""
    height = FORMS[form][0]
    for k in kats["lane"]:
        if k["op"] != "prefix_scan":
            continue
        x = columns_for(form, [k["input"]])
        got = device_scan(devlib, form, x, k["gap"])
        assert list(got[:16]) == k["expect"], (FORMS[form][2], k["name"], list(got[:16]))
        assert np.array_equal(got, expected_columns(oracle, x, height, k["gap"])), (FORMS[form][2], k["name"])


@pytest.mark.parametrize("form", sorted(FORMS))
def test_lane_scan_against_the_oracle(devlib, oracle, oracle_scalar, form):
    """10^5 and more 16-lane vectors per form: uniform over the whole int16 range, all negative, near both saturation bounds, small values around
    the MIN = 0 sentinel (where the zero-shift-in artefact of avx2.rs:315-338 decides), for gap_extend -1 .. -128."""
    height, per_wave, what = FORMS[form]
    rng = np.random.default_rng(1234 + form)
    n_cells = per_wave * max(1, (16 * 8192) // per_wave)      # 8192 vectors per distribution and gap
    total = 0
    for g in (-1, -2, -3, -7, -16, -100, -128):
        for name, gen in (("uniform", lambda n: rng.integers(-32768, 32768, n)), ("negative", lambda n: rng.integers(-32768, 0, n)),
                          ("low", lambda n: rng.integers(-32768, -32000, n)), ("high", lambda n: rng.integers(32000, 32768, n)),
                          ("around the sentinel", lambda n: rng.integers(-40, 41, n)),
                          ("sparse", lambda n: np.where(rng.random(n) < 0.05, rng.integers(-32768, 32768, n), -32768))):
            x = gen(n_cells).astype(np.int16)
            got = device_scan(devlib, form, x, g)
            exp = expected_columns(oracle, x, height, g)
            bad = np.flatnonzero(got != exp)
            assert bad.size == 0, (what, g, name, int(bad[0]), x[bad[0] - bad[0] % 16: bad[0] - bad[0] % 16 + 16], got[bad[0]], exp[bad[0]])
            total += n_cells // 16
    # the AVX2 restatement and the scalar lane model agree on a sample of the same vectors (the full agreement is tests/test_oracle.py's)
    x = rng.integers(-32768, 32768, 16 * 64).astype(np.int16)
    assert np.array_equal(expected_columns(oracle, x, 16, -5), expected_columns(oracle_scalar, x, 16, -5))
    assert total >= 100000
