"""`ByLane` semantics on the CPU oracle: lane i of the per-lane-coefficient entry
equals a single-lane run of the shared-coefficient entry (pinned by the
reference KATs) with configuration i and state i (compose.rs:375-389), and
replicated coefficients reproduce `Lanes` (compose.rs:478-494)."""
import zlib

import numpy as np
import pytest

from tests import _bylane_cases as B
from tests._backends import OracleBackend


@pytest.fixture(scope="module")
def ob(oracle_lib):
    return OracleBackend()


@pytest.mark.parametrize("op,dtype,words,clamp", B.OPS, ids=[o[0] for o in B.OPS])
@pytest.mark.parametrize("layout", [B.FM, B.LM])
def test_bylane_equals_one_shared_run_per_lane(ob, op, dtype, words, clamp, layout):
    rng = np.random.default_rng(zlib.crc32(f"bl-{op}-{layout}".encode()))
    for lanes, frames, n in [(1, 7, 1), (5, 33, 2), (9, 20, 3), (3, 50, 5)]:
        frac = int(rng.integers(0, 32)) if dtype == np.int32 else None
        coef = B.coef_planes(rng, dtype, n, lanes, clamp, frac or 0)
        x = B.samples(rng, dtype, lanes * frames)
        st = B.init_state(rng, dtype, words * n, lanes)
        st0 = st.copy()
        rc, y = ob.bylane(op, coef, frac, n, st, x, lanes, frames, layout)
        assert rc == 0
        xm = x.reshape(frames, lanes).T if layout == B.FM else x.reshape(lanes, frames)
        ym = y.reshape(frames, lanes).T if layout == B.FM else y.reshape(lanes, frames)
        for l in range(lanes):
            cfg = B.shared_cfg(op, dtype, clamp, coef[:, :, l], frac)
            s1 = np.ascontiguousarray(st0[:, l:l + 1])
            rc, y1 = ob.stream(op, cfg, n, s1, np.ascontiguousarray(xm[l]), 1, frames, B.LM)
            assert rc == 0
            assert np.array_equal(B.bits(y1), B.bits(np.ascontiguousarray(ym[l]))), (op, l)
            assert np.array_equal(s1[:, 0], st[:, l])


@pytest.mark.parametrize("op,dtype,words,clamp", B.OPS, ids=[o[0] for o in B.OPS])
def test_replicated_coefficients_reproduce_lanes(ob, op, dtype, words, clamp):
    rng = np.random.default_rng(zlib.crc32(f"rep-{op}".encode()))
    lanes, frames, n = 6, 40, 2
    frac = 29 if dtype == np.int32 else None
    coef = np.repeat(B.coef_planes(rng, dtype, n, 1, clamp, frac or 0), lanes, axis=2)
    x = B.samples(rng, dtype, lanes * frames)
    sa = B.init_state(rng, dtype, words * n, lanes)
    sb = sa.copy()
    rc, ya = ob.bylane(op, coef, frac, n, sa, x, lanes, frames, B.FM)
    rc2, yb = ob.stream(op, B.shared_cfg(op, dtype, clamp, coef[:, :, 0], frac), n, sb, x, lanes, frames, B.FM)
    assert rc == 0 and rc2 == 0
    assert np.array_equal(B.bits(ya), B.bits(yb)) and np.array_equal(sa, sb)


def test_bylane_argument_errors(ob):
    coef = np.zeros((1, 5, 2), np.int32)
    x = np.zeros(8, np.int32)
    st = np.zeros((4, 2), np.uint32)
    assert ob.bylane("biquad_i32_df1", coef, 32, 1, st, x, 2, 4, B.FM)[0] < 0   # frac out of 0..31
    assert ob.bylane("biquad_i32_df1", coef, 30, 1, st, x, 2, 4, 2)[0] < 0      # layout
    rc, y = ob.bylane("biquad_i32_df1", coef, 30, 0, st, x + 3, 2, 4, B.FM)     # empty slice copies
    assert rc == 0 and (y == 3).all()
