"""`iir::normal::Normal` and `iir::wdf::Wdf` on the CPU oracle.  The reference has no test with values for
either (parity unpinned: src/iir/normal.rs and src/iir/wdf.rs carry no test module), so the pins are
(i) the independent Python restatement (oracle/spec.py), bit for bit, and (ii) properties that follow
from the cited lines: a Normal section built by `Normal::from(&ba)` rotates and scales its state by the
pole of `ba` every sample; `Wdf<1, 0x1>` is a unit delay; a quantised Wdf allpass preserves the
signal energy to within rounding."""
import ctypes as C
import math

import numpy as np
import pytest

from idsp_amd import _abi
from oracle import spec
from tests import _harness as H
from tests import _nw_cases as W
from tests._backends import OracleBackend


@pytest.fixture(scope="module")
def ob(oracle_lib):
    return OracleBackend()


@pytest.mark.parametrize("dtype", [np.int32, np.float32, np.float64], ids=["i32", "f32", "f64"])
@pytest.mark.parametrize("layout", [W.FM, W.LM])
def test_normal_oracle_equals_spec(ob, dtype, layout):
    rng = np.random.default_rng(5 + layout)
    op = {np.int32: "normal_i32_df1", np.float32: "normal_f32_df1", np.float64: "normal_f64_df1"}[dtype]
    for n in (1, 2, 3):
        frac = int(rng.integers(20, 31)) if dtype == np.int32 else None
        rows = W.normal_rows(rng, n, dtype, frac)
        cfg = W.normal_cfg(rows, dtype)
        lanes, frames = 3, 40
        words = (8 if dtype == np.float64 else 4) * n
        st = np.zeros((words, lanes), np.uint32)
        if dtype == np.int32:
            x = rng.integers(-(1 << 28), 1 << 28, size=lanes * frames, dtype=np.int64).astype(np.int32)
        else:
            x = rng.standard_normal(lanes * frames).astype(dtype)
        rc, y = ob.stream(op, cfg, n, st, x, lanes, frames, layout)
        assert rc == 0
        xm = x.reshape(frames, lanes).T if layout == W.FM else x.reshape(lanes, frames)
        ym = y.reshape(frames, lanes).T if layout == W.FM else y.reshape(lanes, frames)
        for l in range(lanes):
            states = [spec.DirectForm1() for _ in range(n)]
            for f in range(frames):
                v = int(xm[l, f]) if dtype == np.int32 else dtype(xm[l, f])
                for k in range(n):  # sample-major == stage-major for causal sections
                    v = spec.normal_i32(rows[k][0], frac, states[k], v) if dtype == np.int32 else \
                        spec.normal_float(rows[k], states[k], v, dtype)
                if dtype == np.int32:
                    assert v == int(ym[l, f])
                else:
                    assert dtype(v).tobytes() == dtype(ym[l, f]).tobytes()


def test_normal_from_sos_pole_rotation(ob):
    """`Normal::from(&ba)` (normal.rs:62-76) places the conjugate pole pair of `ba` at p.re +- j p.im; with
    zero input the state (y0, y1) is rotated and scaled by p every sample (normal.rs:44-52), so its
    magnitude decays by exactly |p| = sqrt(a2/a0) per step; real poles are refused like the assert."""
    rng = np.random.default_rng(2)
    for _ in range(10):
        sos = ob.o.lowpass_sos(float(rng.uniform(0.01, 0.4)), q=float(rng.uniform(0.6, 5.0)))
        out = (C.c_double * 5)()
        assert ob.helper("normal_from_sos", (C.c_double * 6)(*sos), out) == 0
        assert list(out) == spec.normal_from_sos(sos)
        assert abs(math.hypot(out[3], out[4]) - math.sqrt(sos[5] / sos[3])) < 1e-12
        st = np.zeros((8, 1), np.uint32)
        st[4:8, 0] = np.array([0.75, -0.25]).view(np.uint32)  # y0 = 0.75, y1 = -0.25
        mags = []
        for _ in range(20):
            ob.stream("normal_f64_df1", H.biquad_f64([list(out)]), 1, st, np.zeros(1), 1, 1, W.LM)
            y0, y1 = st[4:8, 0].copy().view(np.float64)
            mags.append(math.hypot(y0, y1))
        ratios = np.array(mags[1:]) / np.array(mags[:-1])
        assert np.allclose(ratios, math.hypot(out[3], out[4]), rtol=1e-12)
    assert ob.helper("normal_from_sos", (C.c_double * 6)(1, 0, 0, 1, -3.0, 1.0), (C.c_double * 5)()) < 0  # real poles


@pytest.mark.parametrize("layout", [W.FM, W.LM])
def test_wdf_oracle_equals_spec(ob, layout):
    rng = np.random.default_rng(31 + layout)
    for n_sections in (1, 2, 5, 9):
        secs = W.random_wdf(rng, n_sections)
        cfg = W.wdf_array(secs)
        words = ob.helper("wdf_state_words", C.cast(cfg, C.c_void_p), n_sections)
        assert words == sum(s.n for s in secs)
        lanes, frames = 3, 50
        st = rng.integers(0, 1 << 32, size=(words, lanes), dtype=np.uint64).astype(np.uint32)
        st0 = st.copy()
        x = rng.integers(W.I32_MIN, W.I32_MAX, size=lanes * frames, dtype=np.int64, endpoint=True).astype(np.int32)
        rc, y = ob.stream("wdf_i32", cfg, n_sections, st, x, lanes, frames, layout)
        assert rc == 0
        xm = x.reshape(frames, lanes).T if layout == W.FM else x.reshape(lanes, frames)
        ym = y.reshape(frames, lanes).T if layout == W.FM else y.reshape(lanes, frames)
        for l in range(lanes):
            zs, w = [], 0
            for s in secs:
                zs.append([spec.i32(int(v)) for v in st0[w:w + s.n, l]])
                w += s.n
            for f in range(frames):
                v = int(xm[l, f])
                for s, z in zip(secs, zs):
                    v = spec.wdf_process(s.n, s.m, list(s.a), z, v)
                assert v == int(ym[l, f])
            assert [spec.u32(v) for z in zs for v in z] == st[:, l].tolist()


def test_wdf_quantize_and_properties(ob):
    # `Tpa::quantize` ranges (wdf.rs:50-62) for every architecture of the reference's bench
    for m, g in W.WDF_BENCH:
        rc, sec = W.wdf_section(ob, m, g)
        assert rc == 0, (hex(m), g)
        want = [spec.tpa_quantize((m >> (4 * i)) & 0xF, gi) for i, gi in enumerate(g)]
        assert list(sec.a)[:len(g)] == want and None not in want
    assert W.wdf_section(ob, 0xA, [0.3])[0] == _abi.IDSP_EOUTOFRANGE   # A needs 1 > g > 1/2
    assert W.wdf_section(ob, 0xD, [-0.3])[0] == _abi.IDSP_EOUTOFRANGE  # D needs -1 < g < -1/2
    # Wdf<1, 0x1>::default() is a unit delay (adaptor X swaps the ports)
    rc, d = W.wdf_section(ob, 0x1, [0.0])
    x = np.arange(1, 9, dtype=np.int32)
    _, y = ob.stream("wdf_i32", W.wdf_array([d]), 1, np.zeros((1, 1), np.uint32), x, 1, 8, W.LM)
    assert y.tolist() == [0, 1, 2, 3, 4, 5, 6, 7]
    # an allpass keeps the energy of a (band-limited, moderate level) signal
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(20000) * (1 << 20)).astype(np.int32)
    secs = [W.wdf_section(ob, m, g)[1] for m, g in W.WDF_BENCH[:3]]
    words = sum(s.n for s in secs)
    _, y = ob.stream("wdf_i32", W.wdf_array(secs), len(secs), np.zeros((words, 1), np.uint32), x, 1, x.size, W.LM)
    ex, ey = float((x.astype(np.float64) ** 2).sum()), float((y.astype(np.float64) ** 2).sum())
    assert abs(ey / ex - 1.0) < 1e-3
    # argument errors
    bad = _abi.Wdf()
    bad.n = 9
    assert ob.stream("wdf_i32", W.wdf_array([bad]), 1, np.zeros((9, 1), np.uint32), x[:4], 1, 4, W.LM)[0] < 0
