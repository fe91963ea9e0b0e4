"""`Lockin<C>` with biquad arms and with an external LO (src/lockin.rs:16-39) on the CPU oracle: against the spec model
(oracle/spec.py, a second restatement of the cited lines — the reference asserts no values for `Lockin`), and the
reference's own example test examples/ddc_lockin.rs:100-111 (`recovers_dc_iq`) replayed through the f32 entry."""
import ctypes as C

import numpy as np
import pytest

from oracle import spec
from tests import _harness as H
from tests import _lockin_generic_cases as G

FM, LM = H.FM, H.LM


def _idx(f, l, lanes, frames, layout):
    return f * lanes + l if layout == FM else l * frames + f


@pytest.mark.parametrize("layout", [FM, LM])
@pytest.mark.parametrize("n", [1, 2, 4])
def test_phase_form_with_biquad_arms_matches_the_spec_model(n, layout):
    o = H.oracle()
    rng = np.random.default_rng(100 + n)
    lanes, frames = 3, 40
    arr, rows = G.sections_i32(n, rng)
    words = o.fn["lockin_biquad_state_words"](n, 1)
    assert words == 2 + 8 * n
    st = np.zeros((words, lanes), np.uint32)
    st[1] = rng.integers(0, 1 << 32, lanes, dtype=np.uint64).astype(np.uint32)
    st[0] = rng.integers(0, 1 << 32, lanes, dtype=np.uint64).astype(np.uint32)
    x = rng.integers(-(1 << 28), 1 << 28, lanes * frames, dtype=np.int32)
    y = np.empty(lanes * frames * 2, np.int32)
    st0 = st.copy()
    assert o.stream("lockin_i32_biquad_process", arr, n, st, x, y, lanes, frames, layout) == 0
    for l in range(lanes):
        accu = spec.Accu(int(np.int32(st0[0, l])), int(np.int32(st0[1, l])))
        states = [[spec.DirectForm1() for _ in range(n)] for _ in range(2)]
        for f in range(frames):
            i = _idx(f, l, lanes, frames, layout)
            re, im = spec.lockin_phase(lambda s, v: spec.biquad_chain_i32(rows, s, v), states, int(x[i]), accu.next())
            assert (re, im) == (int(y[2 * i]), int(y[2 * i + 1])), (l, f)
        assert int(np.int32(st[0, l])) == accu.state
        for q in range(2):
            for k in range(n):
                got = [int(np.int32(st[2 + (q * n + k) * 4 + w, l])) for w in range(4)]
                assert got == states[q][k].x + states[q][k].y


@pytest.mark.parametrize("layout", [FM, LM])
def test_external_lo_forms_match_the_spec_model(layout):
    o = H.oracle()
    rng = np.random.default_rng(7)
    lanes, frames = 2, 48
    x = rng.integers(-(1 << 28), 1 << 28, lanes * frames, dtype=np.int32)
    lo = rng.integers(-(1 << 31), (1 << 31) - 1, lanes * frames * 2, dtype=np.int64).astype(np.int32)
    # lowpass arms
    ks = [[1 << 22, -(1 << 27)], [1 << 21, -(1 << 26)]]
    cfg = H.lockin_cfg(ks)
    words = o.fn["lockin_state_words"](C.byref(cfg)) - 2
    st = np.zeros((words, lanes), np.uint32)
    y = np.empty(lanes * frames * 2, np.int32)
    assert G.call_lo(o, "lockin_i32_lo_process", cfg, None, st, x, lo, y, lanes, frames, layout, False) == 0
    for l in range(lanes):
        states = [[[0, 0] for _ in ks] for _ in range(2)]
        for f in range(frames):
            i = _idx(f, l, lanes, frames, layout)
            got = spec.lockin_lo(lambda s, v: spec.lowpass_cascade(ks, s, v), states, int(x[i]), (int(lo[2 * i]), int(lo[2 * i + 1])))
            assert got == (int(y[2 * i]), int(y[2 * i + 1])), (l, f)
    # biquad arms, i32
    n = 2
    arr, rows = G.sections_i32(n, rng)
    st = np.zeros((o.fn["lockin_biquad_state_words"](n, 0), lanes), np.uint32)
    assert G.call_lo(o, "lockin_i32_biquad_lo_process", arr, n, st, x, lo, y, lanes, frames, layout, False) == 0
    for l in range(lanes):
        states = [[spec.DirectForm1() for _ in range(n)] for _ in range(2)]
        for f in range(frames):
            i = _idx(f, l, lanes, frames, layout)
            got = spec.lockin_lo(lambda s, v: spec.biquad_chain_i32(rows, s, v), states, int(x[i]), (int(lo[2 * i]), int(lo[2 * i + 1])))
            assert got == (int(y[2 * i]), int(y[2 * i + 1])), (l, f)
    # biquad arms, f32
    arrf, rowsf = G.sections_f32(n, rng)
    xf = rng.standard_normal(lanes * frames).astype(np.float32)
    lof = rng.standard_normal(lanes * frames * 2).astype(np.float32)
    yf = np.empty(lanes * frames * 2, np.float32)
    st = np.zeros((o.fn["lockin_biquad_state_words"](n, 0), lanes), np.uint32)
    assert G.call_lo(o, "lockin_f32_biquad_lo_process", arrf, n, st, xf, lof, yf, lanes, frames, layout, False) == 0
    for l in range(lanes):
        states = [[spec.DirectForm1(np.float32(0)) for _ in range(n)] for _ in range(2)]
        for f in range(frames):
            i = _idx(f, l, lanes, frames, layout)
            re, im = spec.lockin_lo(lambda s, v: spec.biquad_chain_f32(rowsf, s, v), states, xf[i], (lof[2 * i], lof[2 * i + 1]))
            assert np.float32(re).tobytes() == yf[2 * i].tobytes() and np.float32(im).tobytes() == yf[2 * i + 1].tobytes(), (l, f)


def test_ddc_lockin_example_recovers_dc_iq_on_the_oracle():
    """examples/ddc_lockin.rs:100-111: mean I/Q of the last quarter within 3e-3 of 0.5 (cos phi, sin phi), rms error < 6e-3."""
    o = H.oracle()
    x, lo, expected = G.ddc_fixture()
    arr, _ = G.sections_f32(1, None, f0=0.002)
    st = np.zeros((8, 1), np.uint32)
    y = np.empty(x.size * 2, np.float32)
    assert G.call_lo(o, "lockin_f32_biquad_lo_process", arr, 1, st, x, lo, y, 1, x.size, LM, False) == 0
    tail = y.reshape(-1, 2)[12288:].astype(np.float64)
    assert abs(tail[:, 0].mean() - expected[0]) < 3e-3 and abs(tail[:, 1].mean() - expected[1]) < 3e-3
    assert np.sqrt(((tail - np.array(expected)) ** 2).sum(axis=1).mean()) < 6e-3


def test_invalid_arguments_are_reported():
    o = H.oracle()
    arr, _ = G.sections_i32(1, np.random.default_rng(0))
    one = np.zeros(16, np.int32)
    st = np.zeros((16, 1), np.uint32)
    from idsp_amd import _abi

    assert o.stream("lockin_i32_biquad_process", arr, 5, st, one, one, 1, 1, FM) == _abi.IDSP_EINVAL
    assert o.stream("lockin_i32_biquad_process", arr, 0, st, one, one, 1, 1, FM) == _abi.IDSP_EINVAL
    arr[0].frac = 32
    assert o.stream("lockin_i32_biquad_process", arr, 1, st, one, one, 1, 1, FM) == _abi.IDSP_EINVAL
    assert o.fn["lockin_biquad_state_words"](5, 1) == 0 and o.fn["lockin_biquad_state_words"](3, 0) == 24
