"""Replay tests/golden/vectors.npz (made by tests/golden/make_vectors.py from the spec
model) on a back end: every committed input must give the committed output bit for bit."""
import ctypes as C
import os

import numpy as np

from idsp_amd import _abi
from tests import _harness as H

LM = H.LM
V = np.load(os.path.join(os.path.dirname(__file__), "golden", "vectors.npz"))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def run(be):
    lanes, frames = V["f32_x"].shape
    u, lo, hi = (int(t) for t in V["i32_clamp"])
    for frac in (30, 13):
        for name in ("lp", "raw"):
            key = f"i32_f{frac}_{name}"
            ba = [int(t) for t in V[key + "_ba"]]
            x = V[key + "_x"]
            plain, clamp = H.biquad_i32([(ba, frac)]), H.biquad_clamp_i32([(ba, frac, u, lo, hi)])
            for op, words, cfg, out in (("biquad_i32_df1", 4, plain, "df1"), ("biquad_i32_df1_clamp", 4, clamp, "df1c"),
                                        ("biquad_i32_dither", 5, plain, "dit"), ("biquad_i32_dither_clamp", 5, clamp, "ditc"),
                                        ("biquad_i32_wide", 6, plain, "wide"), ("biquad_i32_wide_clamp", 6, clamp, "widec")):
                rc, y = be.stream(op, cfg, 1, np.zeros((words, lanes), np.uint32), x.reshape(-1), lanes, frames, LM)
                assert rc == 0 and np.array_equal(y.reshape(lanes, frames), V[f"{key}_{out}"]), (key, op)
    uf, lof, hif = (float(t) for t in V["f32_clamp"])
    ba = V["f32_ba"].tolist()
    for op, words, cfg, out in (("biquad_f32_df1", 4, H.biquad_f32([ba]), "df1"),
                                ("biquad_f32_df1_clamp", 4, H.biquad_clamp_f32([(ba, uf, lof, hif)]), "df1c"),
                                ("biquad_f32_df2t", 2, H.biquad_f32([ba]), "df2t"),
                                ("biquad_f32_df2t_clamp", 2, H.biquad_clamp_f32([(ba, uf, lof, hif)]), "df2tc")):
        rc, y = be.stream(op, cfg, 1, np.zeros((words, lanes), np.uint32), V["f32_x"].reshape(-1), lanes, frames, LM)
        assert rc == 0 and np.array_equal(bits(y.reshape(lanes, frames)), bits(V["f32_" + out])), op
    for ts in (0, 1):
        for stages in (1, 4, 5):
            for kind in ("dec", "int"):
                cfg = _abi.HbfCascadeF32()
                assert be.helper(f"hbf_{kind}_cascade", ts, stages, C.byref(cfg)) == 0
                x, want = V[f"hbf{kind}_{ts}_{stages}_x"], V[f"hbf{kind}_{ts}_{stages}_y"]
                fr = want.size if kind == "dec" else x.size
                st = np.zeros((be.helper(f"hbf_{kind}_state_words", C.byref(cfg)), 1), np.uint32)
                rc, y = be.cfgcall(f"hbf_{kind}_f32", cfg, st, x, (want.size,), np.float32, 1, fr, LM)
                assert rc == 0 and np.array_equal(bits(y), bits(want)), (kind, ts, stages)
    taps = V["fir_taps"]
    for kind in range(4):
        cfg = _abi.FirSymF32()
        cfg.kind, cfg.m = kind, taps.size
        for k, t in enumerate(taps):
            cfg.taps[k] = t
        st = np.zeros((be.helper("fir_sym_state_words", C.byref(cfg)), 1), np.uint32)
        rc, y = be.cfgcall("fir_sym_f32_process", cfg, st, V["fir_x"], (V["fir_x"].size,), np.float32, 1, V["fir_x"].size, LM)
        assert rc == 0 and np.array_equal(bits(y), bits(V[f"fir_{kind}_y"])), kind
    rc, cs = be.cossin(V["cossin_phase"])
    assert rc == 0 and np.array_equal(cs, V["cossin_out"])
    for order in (1, 2):
        ks = V[f"lowpass{order}_k"].tolist()
        cfg = H.lockin_cfg(ks)
        x = V[f"lowpass{order}_x"]
        rc, y = be.cfgcall("lowpass_i32", cfg, np.zeros((2 * order * len(ks), 1), np.uint32), x, (x.size,), np.int32, 1, x.size, LM)
        assert rc == 0 and np.array_equal(y, V[f"lowpass{order}_y"])
        st = np.zeros((2 + 4 * order * len(ks), 1), np.uint32)
        st[:2, 0] = V[f"lockin{order}_accu"].astype(np.uint32)
        rc, y = be.cfgcall("lockin_i32_process", cfg, st, x, (2 * x.size,), np.int32, 1, x.size, LM)
        assert rc == 0 and np.array_equal(y.reshape(-1, 2), V[f"lockin{order}_y"])
