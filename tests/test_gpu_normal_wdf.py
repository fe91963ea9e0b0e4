"""`Normal` and `Wdf` sections on the HIP path through the C ABI vs the CPU oracle: i32 bit-exact,
f32/f64 0 ULP (allowed: 1), written-back state, continuation, in place, both layouts, chains longer
than one fused launch."""
import ctypes as C

import numpy as np
import pytest

from tests import _harness as H
from tests import _nw_cases as W
from tests._backends import GpuBackend, OracleBackend

pytestmark = pytest.mark.gpu
SHAPES = [(1, 1), (63, 23), (65, 47), (257, 64), (100, 65), (3, 1000), (1028, 77), (512, 300), (2048, 64)]


@pytest.fixture(scope="module")
def bes(gpu):
    return OracleBackend(), GpuBackend()


def bits(a):
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


@pytest.mark.parametrize("dtype", [np.int32, np.float32, np.float64], ids=["i32", "f32", "f64"])
@pytest.mark.parametrize("layout", [W.FM, W.LM])
def test_normal_parity(bes, dtype, layout):
    ob, gb = bes
    rng = np.random.default_rng(11 + layout)
    op = {np.int32: "normal_i32_df1", np.float32: "normal_f32_df1", np.float64: "normal_f64_df1"}[dtype]
    for (lanes, frames), n in zip(SHAPES, [1, 2, 1, 3, 5, 1, 1, 2, 4]):
        frac = int(rng.integers(0, 32)) if dtype == np.int32 else None
        cfg = W.normal_cfg(W.normal_rows(rng, n, dtype, frac), dtype)
        words = (8 if dtype == np.float64 else 4) * n
        if dtype == np.int32:
            x = rng.integers(W.I32_MIN, W.I32_MAX, size=lanes * frames, dtype=np.int64, endpoint=True).astype(np.int32)
            init = rng.integers(0, 1 << 32, size=(words, lanes), dtype=np.uint64).astype(np.uint32)
        else:
            x = rng.standard_normal(lanes * frames).astype(dtype)
            init = np.zeros((words, lanes), np.uint32)
        for inplace in (False, True):
            so, sg = init.copy(), init.copy()
            rco, yo = ob.stream(op, cfg, n, so, x.copy(), lanes, frames, layout, inplace=inplace)
            rcg, yg = gb.stream(op, cfg, n, sg, x.copy(), lanes, frames, layout, inplace=inplace)
            assert rco == 0 and rcg == 0, H.engine().err()
            assert np.array_equal(bits(yo), bits(yg)) and np.array_equal(so, sg)
            rco, yo = ob.stream(op, cfg, n, so, x[::-1].copy(), lanes, frames, layout)
            rcg, yg = gb.stream(op, cfg, n, sg, x[::-1].copy(), lanes, frames, layout)
            assert np.array_equal(bits(yo), bits(yg)) and np.array_equal(so, sg)


@pytest.mark.parametrize("layout", [W.FM, W.LM])
def test_wdf_parity(bes, layout):
    ob, gb = bes
    rng = np.random.default_rng(21 + layout)
    for (lanes, frames), n_sections in zip(SHAPES, [1, 2, 3, 4, 5, 9, 1, 6, 2]):
        secs = W.random_wdf(rng, n_sections)
        cfg = W.wdf_array(secs)
        words = gb.helper("wdf_state_words", C.cast(cfg, C.c_void_p), n_sections)
        assert words == ob.helper("wdf_state_words", C.cast(cfg, C.c_void_p), n_sections)
        init = rng.integers(0, 1 << 32, size=(words, lanes), dtype=np.uint64).astype(np.uint32)
        x = rng.integers(W.I32_MIN, W.I32_MAX, size=lanes * frames, dtype=np.int64, endpoint=True).astype(np.int32)
        for inplace in (False, True):
            so, sg = init.copy(), init.copy()
            rco, yo = ob.stream("wdf_i32", cfg, n_sections, so, x.copy(), lanes, frames, layout, inplace=inplace)
            rcg, yg = gb.stream("wdf_i32", cfg, n_sections, sg, x.copy(), lanes, frames, layout, inplace=inplace)
            assert rco == 0 and rcg == 0, H.engine().err()
            assert np.array_equal(yo, yg) and np.array_equal(so, sg), (lanes, frames, n_sections)
            rco, yo = ob.stream("wdf_i32", cfg, n_sections, so, x[::-1].copy(), lanes, frames, layout)
            rcg, yg = gb.stream("wdf_i32", cfg, n_sections, sg, x[::-1].copy(), lanes, frames, layout)
            assert np.array_equal(yo, yg) and np.array_equal(so, sg)


def test_wdf_bench_architectures_and_helpers(bes):
    """The sections of the reference's embedded bench (tests/embedded/src/bin/biquad.rs:126-164) in one chain."""
    ob, gb = bes
    secs = []
    for m, g in W.WDF_BENCH:
        rco, so_ = W.wdf_section(ob, m, g)
        rcg, sg_ = W.wdf_section(gb, m, g)
        assert rco == 0 and rcg == 0 and list(so_.a) == list(sg_.a)
        secs.append(sg_)
    cfg = W.wdf_array(secs)
    words = gb.helper("wdf_state_words", C.cast(cfg, C.c_void_p), len(secs))
    rng = np.random.default_rng(3)
    lanes, frames = 300, 200
    x = (rng.standard_normal(lanes * frames) * (1 << 24)).astype(np.int32)
    so, sg = np.zeros((words, lanes), np.uint32), np.zeros((words, lanes), np.uint32)
    _, yo = ob.stream("wdf_i32", cfg, len(secs), so, x, lanes, frames, W.FM)
    rc, yg = gb.stream("wdf_i32", cfg, len(secs), sg, x, lanes, frames, W.FM)
    assert rc == 0 and np.array_equal(yo, yg) and np.array_equal(so, sg)
    assert W.wdf_section(gb, 0xA, [0.3])[0] == -12 and "out of range" in H.engine().err()
    out = (C.c_double * 5)()
    assert gb.helper("normal_from_sos", (C.c_double * 6)(1, 0, 0, 1, -3.0, 1.0), out) == -1
    sos = ob.o.lowpass_sos(0.1)
    want = (C.c_double * 5)()
    assert gb.helper("normal_from_sos", (C.c_double * 6)(*sos), out) == 0 and ob.helper("normal_from_sos", (C.c_double * 6)(*sos), want) == 0
    assert list(out) == list(want)
