"""f64 half-band cascades and symmetric FIR on HIP against the CPU oracle: 0 ULP outputs and state, both layouts, ragged
shapes through several chunks, consecutive calls on one state (include/idsp_hip.h `idsp_hbf_{dec,int}_f64`,
`idsp_fir_sym_f64_process`)."""
import ctypes as C

import numpy as np
import pytest
import torch

from idsp_amd import _abi
from tests import _harness as H

pytestmark = pytest.mark.gpu
FM, LM = H.FM, H.LM
DEV = "cuda:0"


def dev(a):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(DEV)


def pair(name, cfg, words, nin_of, nout_of, shapes, layout, rng):
    o, e = H.oracle(), H.engine()
    for lanes, frames in shapes:
        st0 = rng.standard_normal((words // 2, lanes)).view(np.uint32).reshape(words // 2, lanes, 2).transpose(0, 2, 1).reshape(words, lanes).copy()
        so, sg = st0.copy(), dev(st0)
        for rep in range(2):
            x = rng.standard_normal(lanes * nin_of(frames))
            yo = np.empty(lanes * nout_of(frames))
            yg = torch.full((yo.size,), float("nan"), dtype=torch.float64, device=DEV)
            assert o.cfgcall(name, cfg, so, x, yo, lanes, frames, layout) == 0
            assert e.cfgcall(name, cfg, sg, dev(x), yg, lanes, frames, layout) == 0, e.err()
            torch.cuda.synchronize()
            assert np.array_equal(yg.cpu().numpy().view(np.uint64), yo.view(np.uint64)), (name, lanes, frames, layout, rep)
            assert np.array_equal(sg.cpu().numpy().view(np.uint32), so), (name, lanes, frames, layout, rep)


@pytest.mark.parametrize("layout", [FM, LM])
@pytest.mark.parametrize("tap_set,stages", [(0, 1), (0, 2), (0, 4), (1, 3), (1, 5)])
@pytest.mark.parametrize("kind", ["dec", "int"])
def test_hbf_f64_parity(kind, tap_set, stages, layout):
    rng = np.random.default_rng(100 * stages + 10 * tap_set + layout)
    cfg = _abi.HbfCascadeF64()
    assert H.oracle().fn[f"hbf_{kind}_cascade_f64"](tap_set, stages, C.byref(cfg)) == 0
    words = H.oracle().fn[f"hbf_{kind}_state_words_f64"](C.byref(cfg))
    R, ch = 1 << stages, 2048 >> stages
    hi, lo = (lambda f: f * R), (lambda f: f)
    shapes = [(1, 1), (3, 5), (2, ch - 1), (2, ch), (3, ch + 1), (5, 2 * ch + 7), (17, 40), (300, 9)]
    pair(f"hbf_{kind}_f64", cfg, words, hi if kind == "dec" else lo, lo if kind == "dec" else hi, shapes, layout, rng)


@pytest.mark.parametrize("layout", [FM, LM])
@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_fir_sym_f64_parity(kind, layout):
    rng = np.random.default_rng(300 + 10 * kind + layout)
    for m in (1, 4, 23, 32):
        cfg = _abi.FirSymF64()
        cfg.kind, cfg.m = kind, m
        for k, v in enumerate(rng.standard_normal(m) * 0.3):
            cfg.taps[k] = v
        words = H.oracle().fn["fir_sym_state_words_f64"](C.byref(cfg))
        pair("fir_sym_f64_process", cfg, words, lambda f: f, lambda f: f, [(1, 1), (3, 255), (2, 2048), (5, 2049), (17, 5000)], layout, rng)


def test_f64_custom_taps_and_errors():
    e = H.engine()
    cfg = _abi.HbfCascadeF64()
    cfg.stages = 6
    one = torch.zeros(8, dtype=torch.float64, device=DEV)
    st = torch.zeros((64, 1), dtype=torch.int32, device=DEV)
    assert e.cfgcall("hbf_dec_f64", cfg, st, one, one, 1, 1, LM) == _abi.IDSP_EINVAL
    assert e.fn["hbf_dec_state_words_f64"](C.byref(cfg)) == 0
    cfg.stages, cfg.m[0] = 1, 7  # a tap count no built-in cascade has
    for k in range(7):
        cfg.taps[0][k] = 0.1 * (k + 1)
    words = e.fn["hbf_dec_state_words_f64"](C.byref(cfg))
    assert words == 2 * 19
    pair("hbf_dec_f64", cfg, words, lambda f: 2 * f, lambda f: f, [(4, 100)], FM, np.random.default_rng(1))
