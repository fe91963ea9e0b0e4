"""Per-lane coefficients (`ByLane<[C; N]>`, dsp-process/src/compose.rs:363-390) on the
HIP path through the C ABI vs the CPU oracle: bit-exact for i32, 0 ULP (allowed: 1) for
f32/f64, including the written-back state, stream continuation and in-place operation."""
import zlib

import numpy as np
import pytest

from tests import _bylane_cases as B
from tests import _harness as H
from tests._backends import GpuBackend, OracleBackend

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1, 1), (63, 23, 2), (64, 24, 1), (65, 47, 3), (257, 64, 1), (100, 65, 4), (3, 1000, 2), (1028, 77, 1),
          (512, 300, 2), (4096, 50, 1), (256, 1001, 5)]


@pytest.fixture(scope="module")
def bes(gpu):
    return OracleBackend(), GpuBackend()


@pytest.mark.parametrize("op,dtype,words,clamp", B.OPS, ids=[o[0] for o in B.OPS])
@pytest.mark.parametrize("layout", [B.FM, B.LM])
def test_bylane_parity(bes, op, dtype, words, clamp, layout):
    ob, gb = bes
    rng = np.random.default_rng(zlib.crc32(f"gbl-{op}-{layout}".encode()))
    for lanes, frames, n in SHAPES:
        frac = int(rng.integers(0, 32)) if dtype == np.int32 else None
        coef = B.coef_planes(rng, dtype, n, lanes, clamp, frac or 0)
        x = B.samples(rng, dtype, lanes * frames)
        init = B.init_state(rng, dtype, words * n, lanes)
        for inplace in (False, True):
            so, sg = init.copy(), init.copy()
            rco, yo = ob.bylane(op, coef, frac, n, so, x.copy(), lanes, frames, layout, inplace=inplace)
            rcg, yg = gb.bylane(op, coef, frac, n, sg, x.copy(), lanes, frames, layout, inplace=inplace)
            assert rco == 0 and rcg == 0, H.engine().err()
            assert np.array_equal(B.bits(yo), B.bits(yg)), (op, lanes, frames, n)
            assert np.array_equal(so, sg)
            x2 = x[::-1].copy()  # continue the stream from the written-back state
            rco, yo = ob.bylane(op, coef, frac, n, so, x2, lanes, frames, layout)
            rcg, yg = gb.bylane(op, coef, frac, n, sg, x2, lanes, frames, layout)
            assert rco == 0 and rcg == 0
            assert np.array_equal(B.bits(yo), B.bits(yg)) and np.array_equal(so, sg)


def test_bylane_lds_path_shape_and_errors(bes):
    """2048 lanes x 256 frames takes the LDS-DMA frame-major kernel (lanes % 256 == 0)."""
    ob, gb = bes
    rng = np.random.default_rng(5)
    lanes, frames, n = 2048, 256, 2
    coef = B.coef_planes(rng, np.int32, n, lanes, False, 30)
    x = B.samples(rng, np.int32, lanes * frames)
    so, sg = np.zeros((8, lanes), np.uint32), np.zeros((8, lanes), np.uint32)
    _, yo = ob.bylane("biquad_i32_df1", coef, 30, n, so, x, lanes, frames, B.FM)
    rc, yg = gb.bylane("biquad_i32_df1", coef, 30, n, sg, x, lanes, frames, B.FM)
    assert rc == 0 and np.array_equal(yo, yg) and np.array_equal(so, sg)
    assert gb.bylane("biquad_i32_df1", coef, 32, n, sg, x, lanes, frames, B.FM)[0] == -1
    assert "frac" in H.engine().err()
    assert gb.bylane("biquad_i32_df1", coef, 30, 65, sg, x, lanes, frames, B.FM)[0] == -1
