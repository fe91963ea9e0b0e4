"""Parity proper: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs.  Integer paths must be bit-identical; f32 paths are held to
0 ULP here (north_star allows 1 ULP — `F32_ULP_TOL` states the bar used).
Every case also compares the written-back state, continues the stream with a
second call (chunked == whole, the reference's streaming semantics) and covers
ragged sizes around the kernels' internal tile sizes (24/48-frame register
window, 64x64 lane-major tiles, 4096-sample HBF chunks)."""
import ctypes as C
import zlib

import numpy as np
import pytest

from idsp_amd import _abi
from tests import _harness as H
from tests._backends import GpuBackend, OracleBackend

pytestmark = pytest.mark.gpu

FM, LM = H.FM, H.LM
F32_ULP_TOL = 0  # bar used by these tests (allowed by north_star: 1)
I32_MIN, I32_MAX = -(1 << 31), (1 << 31) - 1


@pytest.fixture(scope="module")
def bes(gpu):
    return OracleBackend(), GpuBackend()


def adversarial_i32(rng, shape):
    x = rng.integers(I32_MIN, I32_MAX, size=shape, dtype=np.int64, endpoint=True).astype(np.int32)
    flat = x.reshape(-1)
    if flat.size >= 8:
        k = max(1, flat.size // 8)
        idx = rng.choice(flat.size, size=k, replace=False)
        flat[idx] = rng.choice(np.array([I32_MIN, I32_MAX, 0, 1, -1], dtype=np.int32), size=k)
    return x


def adversarial_f32(rng, shape):
    x = rng.standard_normal(size=shape).astype(np.float32)
    flat = x.reshape(-1)
    if flat.size >= 8:
        k = max(1, flat.size // 10)
        idx = rng.choice(flat.size, size=k, replace=False)
        flat[idx] = rng.choice(np.array([0.0, -0.0, 1e-41, -3e-42, 1e30, -1e30, 1.0], dtype=np.float32), size=k)
    return x


def random_i32_sections(rng, n, clamp):
    rows = []
    for _ in range(n):
        frac = int(rng.integers(0, 32))
        kind = rng.integers(0, 3)
        if kind == 0:  # a stable lowpass quantised to Q(frac)
            o = H.oracle()
            sos = o.lowpass_sos(float(rng.uniform(0.001, 0.4)), gain=float(rng.uniform(0.1, 1.0)))
            q = _abi.BiquadI32()
            o.fn["biquad_i32_from_sos"]((C.c_double * 6)(*sos), frac, C.byref(q))
            ba = list(q.ba)
        elif kind == 1:  # arbitrary bits incl. saturated values: exercises wrapping
            ba = adversarial_i32(rng, 5).tolist()
        else:
            ba = rng.integers(-(1 << 20), 1 << 20, size=5).tolist()
        if clamp:
            lo, hi = sorted(rng.integers(I32_MIN, I32_MAX, size=2, dtype=np.int64).tolist())
            if rng.integers(0, 4) == 0:
                lo, hi = I32_MIN, I32_MAX
            rows.append((ba, frac, int(rng.integers(-1000, 1000)), int(lo), int(hi)))
        else:
            rows.append((ba, frac))
    return rows


I32_OPS = [
    ("biquad_i32_df1", 4, False), ("biquad_i32_df1_clamp", 4, True),
    ("biquad_i32_dither", 5, False), ("biquad_i32_dither_clamp", 5, True),
    ("biquad_i32_wide", 6, False), ("biquad_i32_wide_clamp", 6, True),
]
SHAPES = [(1, 1), (1, 100), (63, 23), (64, 24), (65, 47), (130, 48), (64, 49), (257, 64), (100, 65), (70, 130), (3, 1000),
          (1028, 77), (4096, 50), (516, 9), (512, 300), (256, 1001), (768, 41), (1024, 8), (256, 7), (2048, 121)]


def run_both(bes, op, cfg, n, words, x, lanes, frames, layout, rng, is_float=False):
    """One call + continuation + in-place on both back ends; returns nothing, asserts parity."""
    ob, gb = bes
    init = rng.integers(0, 1 << 32, size=(words, lanes), dtype=np.uint64).astype(np.uint32)
    if is_float:  # random finite floats as initial state
        init = rng.standard_normal(size=(words, lanes)).astype(np.float32).view(np.uint32)
    for inplace in (False, True):
        so, sg = init.copy(), init.copy()
        xo, xg = x.copy(), x.copy()
        rco, yo = ob.stream(op, cfg, n, so, xo, lanes, frames, layout, inplace=inplace)
        rcg, yg = gb.stream(op, cfg, n, sg, xg, lanes, frames, layout, inplace=inplace)
        assert rco == 0 and rcg == 0, (rco, rcg, H.engine().err())
        if is_float:
            assert H.ulp_diff_f32(yo, yg).max(initial=0) <= F32_ULP_TOL
            assert H.ulp_diff_f32(so.view(np.float32), sg.view(np.float32)).max(initial=0) <= F32_ULP_TOL
        else:
            assert np.array_equal(yo, yg)
            assert np.array_equal(so, sg)
        # continue the stream from the written-back state
        x2 = x[::-1].copy()
        rco, yo = ob.stream(op, cfg, n, so, x2, lanes, frames, layout)
        rcg, yg = gb.stream(op, cfg, n, sg, x2, lanes, frames, layout)
        assert rco == 0 and rcg == 0
        if is_float:
            assert H.ulp_diff_f32(yo, yg).max(initial=0) <= F32_ULP_TOL
        else:
            assert np.array_equal(yo, yg) and np.array_equal(so, sg)


@pytest.mark.parametrize("op,words,clamp", I32_OPS)
@pytest.mark.parametrize("layout", [FM, LM])
def test_biquad_i32_parity(bes, op, words, clamp, layout):
    rng = np.random.default_rng(zlib.crc32(f"{op}-{layout}".encode()))
    for (lanes, frames), n in zip(SHAPES, [1, 2, 1, 3, 4, 1, 5, 1, 9, 2, 1, 1, 2, 1, 1, 3, 1, 2, 1, 1]):
        rows = random_i32_sections(rng, n, clamp)
        cfg = H.biquad_clamp_i32(rows) if clamp else H.biquad_i32(rows)
        x = adversarial_i32(rng, lanes * frames)
        run_both(bes, op, cfg, n, words * n, x, lanes, frames, layout, rng)


@pytest.mark.parametrize("layout", [FM, LM])
def test_cascade_i32_parity(bes, layout):
    rng = np.random.default_rng(11 + layout)
    for (lanes, frames), n in zip(SHAPES, [1, 2, 3, 4, 5, 6, 7, 8, 1, 4, 8, 1, 2, 3, 2, 5, 1, 8, 2, 3]):
        cfg = H.biquad_i32(random_i32_sections(rng, n, False))
        x = adversarial_i32(rng, lanes * frames)
        run_both(bes, "cascade_i32_df1", cfg, n, 2 + 2 * n, x, lanes, frames, layout, rng)


def random_f32_sections(rng, n, clamp):
    rows = []
    o = H.oracle()
    for _ in range(n):
        if rng.integers(0, 3):
            sos = o.lowpass_sos(float(rng.uniform(0.001, 0.45)), gain=float(rng.uniform(0.1, 2.0)),
                                highpass=bool(rng.integers(0, 2)))
            q = _abi.BiquadF32()
            o.fn["biquad_f32_from_sos_f64"]((C.c_double * 6)(*sos), C.byref(q))
            ba = list(q.ba)
        else:
            ba = (rng.standard_normal(5) * 0.4).astype(np.float32).tolist()
        if clamp:
            lo, hi = sorted((rng.standard_normal(2) * 2).tolist())
            if rng.integers(0, 3) == 0:
                lo, hi = -np.inf, np.inf
            rows.append((ba, float(rng.standard_normal() * 0.1), lo, hi))
        else:
            rows.append(ba)
    return rows


F32_OPS = [("biquad_f32_df1", 4, False), ("biquad_f32_df1_clamp", 4, True),
           ("biquad_f32_df2t", 2, False), ("biquad_f32_df2t_clamp", 2, True)]


@pytest.mark.parametrize("op,words,clamp", F32_OPS)
@pytest.mark.parametrize("layout", [FM, LM])
def test_biquad_f32_parity(bes, op, words, clamp, layout):
    rng = np.random.default_rng(zlib.crc32(f"{op}-{layout}-f".encode()))
    for (lanes, frames), n in zip(SHAPES, [1, 2, 1, 3, 4, 1, 5, 1, 9, 2, 1, 1, 2, 1, 1, 3, 1, 2, 1, 1]):
        rows = random_f32_sections(rng, n, clamp)
        cfg = H.biquad_clamp_f32(rows) if clamp else H.biquad_f32(rows)
        x = adversarial_f32(rng, lanes * frames)
        run_both(bes, op, cfg, n, words * n, x, lanes, frames, layout, rng, is_float=True)


@pytest.mark.parametrize("layout", [FM, LM])
def test_cascade_f32_parity(bes, layout):
    rng = np.random.default_rng(13 + layout)
    for (lanes, frames), n in zip(SHAPES, [1, 2, 3, 4, 5, 6, 7, 8, 1, 4, 8, 1, 2, 3, 2, 5, 1, 8, 2, 3]):
        cfg = H.biquad_f32(random_f32_sections(rng, n, False))
        x = adversarial_f32(rng, lanes * frames)
        run_both(bes, "cascade_f32_df1", cfg, n, 2 + 2 * n, x, lanes, frames, layout, rng, is_float=True)


F64_OPS = [("biquad_f64_df1", 8, False), ("biquad_f64_df1_clamp", 8, True),
           ("biquad_f64_df2t", 4, False), ("biquad_f64_df2t_clamp", 4, True)]


def adversarial_f64(rng, shape):
    x = rng.standard_normal(size=shape)
    flat = x.reshape(-1)
    if flat.size >= 8:
        k = max(1, flat.size // 10)
        idx = rng.choice(flat.size, size=k, replace=False)
        flat[idx] = rng.choice(np.array([0.0, -0.0, 1e-310, -3e-320, 1e300, -1e300, 1.0]), size=k)
    return x


@pytest.mark.parametrize("op,words,clamp", F64_OPS)
@pytest.mark.parametrize("layout", [FM, LM])
def test_biquad_f64_parity(bes, op, words, clamp, layout):
    """`Biquad<f64>`: the same generic impls; held to 0 ULP (allowed: 1)."""
    ob, gb = bes
    rng = np.random.default_rng(zlib.crc32(f"{op}-{layout}-d".encode()))
    for (lanes, frames), n in zip(SHAPES, [1, 2, 1, 3, 4, 1, 5, 1, 9, 2, 1, 1, 2, 1, 1, 3, 1, 2, 1, 1]):
        rows = random_f32_sections(rng, n, clamp)  # f32-representable coefficients are valid f64 ones
        rows = [(r[0], r[1], r[2], r[3]) if clamp else r for r in rows]
        cfg = H.biquad_clamp_f64(rows) if clamp else H.biquad_f64(rows)
        x = adversarial_f64(rng, lanes * frames)
        init = rng.standard_normal(size=(words * n // 2, lanes)).view(np.uint32).reshape(words * n // 2, lanes, 2)
        init = np.ascontiguousarray(init.transpose(0, 2, 1)).reshape(words * n, lanes)  # value v -> words 2v (lo), 2v+1 (hi)
        for inplace in (False, True):
            so, sg = init.copy(), init.copy()
            rco, yo = ob.stream(op, cfg, n, so, x.copy(), lanes, frames, layout, inplace=inplace)
            rcg, yg = gb.stream(op, cfg, n, sg, x.copy(), lanes, frames, layout, inplace=inplace)
            assert rco == 0 and rcg == 0, H.engine().err()
            assert H.ulp_diff_f64(yo, yg).max(initial=0) <= F32_ULP_TOL
            assert np.array_equal(so, sg)
            rco, yo = ob.stream(op, cfg, n, so, x[::-1].copy(), lanes, frames, layout)
            rcg, yg = gb.stream(op, cfg, n, sg, x[::-1].copy(), lanes, frames, layout)
            assert H.ulp_diff_f64(yo, yg).max(initial=0) <= F32_ULP_TOL and np.array_equal(so, sg)


@pytest.mark.parametrize("layout", [FM, LM])
def test_cascade_f64_parity(bes, layout):
    ob, gb = bes
    rng = np.random.default_rng(17 + layout)
    for (lanes, frames), n in zip(SHAPES, [1, 2, 3, 4, 5, 6, 7, 8, 1, 4, 8, 1, 2, 3, 2, 5, 1, 8, 2, 3]):
        cfg = H.biquad_f64(random_f32_sections(rng, n, False))
        x = adversarial_f64(rng, lanes * frames)
        so, sg = np.zeros((4 + 4 * n, lanes), np.uint32), np.zeros((4 + 4 * n, lanes), np.uint32)
        for rep in range(2):
            rco, yo = ob.stream("cascade_f64_df1", cfg, n, so, x, lanes, frames, layout)
            rcg, yg = gb.stream("cascade_f64_df1", cfg, n, sg, x, lanes, frames, layout)
            assert rco == 0 and rcg == 0
            assert H.ulp_diff_f64(yo, yg).max(initial=0) <= F32_ULP_TOL and np.array_equal(so, sg)
        # same samples as n separate DF1 sections
        _, yr = gb.stream("biquad_f64_df1", cfg, n, np.zeros((8 * n, lanes), np.uint32), x, lanes, frames, layout)
        _, yc = gb.stream("cascade_f64_df1", cfg, n, np.zeros((4 + 4 * n, lanes), np.uint32), x, lanes, frames, layout)
        assert np.array_equal(yr.view(np.uint64), yc.view(np.uint64))


def test_f32_nonfinite_and_denormal_tail(bes):
    """inf/NaN propagate like the reference (NaN compares as NaN); an impulse
    response decaying into denormals stays bit-identical (denormals enabled)."""
    ob, gb = bes
    o = H.oracle()
    q = _abi.BiquadF32()
    o.fn["biquad_f32_from_sos_f64"]((C.c_double * 6)(*o.lowpass_sos(0.2)), C.byref(q))
    cfg = H.biquad_f32([list(q.ba)])
    frames = 3000
    x = np.zeros(frames, np.float32)
    x[0] = 1e-30
    for op, words in (("biquad_f32_df1", 4), ("biquad_f32_df2t", 2)):
        so, sg = np.zeros((words, 1), np.uint32), np.zeros((words, 1), np.uint32)
        _, yo = ob.stream(op, cfg, 1, so, x, 1, frames, LM)
        _, yg = gb.stream(op, cfg, 1, sg, x, 1, frames, LM)
        assert np.any((yo != 0) & (np.abs(yo) < 1.1754944e-38)), "test must reach the denormal range"
        assert np.array_equal(yo.view(np.uint32), yg.view(np.uint32))
    x = np.array([1.0, np.inf, -np.inf, np.nan, 1.0, 2.0], np.float32)
    so, sg = np.zeros((4, 1), np.uint32), np.zeros((4, 1), np.uint32)
    _, yo = ob.stream("biquad_f32_df1", cfg, 1, so, x, 1, x.size, FM)
    _, yg = gb.stream("biquad_f32_df1", cfg, 1, sg, x, 1, x.size, FM)
    assert np.array_equal(np.isnan(yo), np.isnan(yg))
    assert H.ulp_diff_f32(yo, yg).max() == 0


def test_empty_and_error_paths(bes):
    ob, gb = bes
    e = H.engine()
    cfg = H.biquad_i32([([1 << 30, 0, 0, 0, 0], 30)])
    # empty slice of sections: identity copy (dsp-process/src/compose.rs:63-65)
    x = np.arange(64 * 5, dtype=np.int32)
    rc, y = gb.stream("biquad_i32_df1", None, 0, np.zeros((4, 64), np.uint32), x, 64, 5, FM)
    assert rc == 0 and np.array_equal(y, x)
    # zero frames: state must come back untouched
    st = np.arange(4 * 3, dtype=np.uint32).reshape(4, 3)
    rc, _ = gb.stream("biquad_i32_df1", cfg, 1, st, np.zeros(0, np.int32), 3, 0, LM)
    assert rc == 0 and np.array_equal(st, np.arange(12, dtype=np.uint32).reshape(4, 3))
    # contract violations are reported, never abort
    bad = H.biquad_i32([([1, 0, 0, 0, 0], 32)])  # const assert F < 32
    rc, _ = gb.stream("biquad_i32_df1", bad, 1, np.zeros((4, 1), np.uint32), np.zeros(4, np.int32), 1, 4, FM)
    assert rc == _abi.IDSP_EINVAL and "frac" in e.err()
    rc, _ = gb.stream("biquad_i32_df1", cfg, 1, np.zeros((4, 1), np.uint32), np.zeros(4, np.int32), 1, 4, 7)
    assert rc == _abi.IDSP_EINVAL and "layout" in e.err()
    rc, _ = gb.stream("cascade_i32_df1", H.biquad_i32([([1, 0, 0, 0, 0], 3)] * 9), 9, np.zeros((20, 1), np.uint32),
                      np.zeros(4, np.int32), 1, 4, FM)
    assert rc == _abi.IDSP_EINVAL


# ------------------------------------------------------------------- hbf
def _cascade(kind, tap_set, stages):
    cfg = _abi.HbfCascadeF32()
    assert H.oracle().fn[f"hbf_{kind}_cascade"](tap_set, stages, C.byref(cfg)) == 0
    return cfg


HBF_CASES = [(_cascade, 0, s) for s in (1, 2, 3, 4, 5)] + [(_cascade, 1, s) for s in (2, 4, 5)]


def hbf_shapes(stages):
    ch = 4096 >> stages
    return [(1, 1), (3, 5), (2, ch - 1), (2, ch), (3, ch + 1), (1, 2 * ch + 7), (17, 40),
            (4, 1), (4, ch + 3), (8, 130), (12, 2 * ch + 5), (20, 70),  # whole 4-lane workgroups: FM block kernel
            # whole 16-lane groups: the FM ring kernel (/16); rounds of 1024 input samples: one, ragged, the first / last two (SAFE) and FAST ones
            (64, 3 * (1024 >> stages) + 1), (16, 1), (16, (1024 >> stages) - 1), (48, 2 * (1024 >> stages) + 5), (16, 7 * (1024 >> stages) + 3),
            # LM blocked kernel (hbf_blk.h): stage s >= 2 runs every 2^(s-1) rounds of 1024 input samples — whole runs, a flushed
            # partial run behind them, two runs of the deepest stage (/32: 8 rounds) and a ragged tail
            (3, 9 * (1024 >> stages)), (2, 6 * (1024 >> stages) + 4), (2, 17 * (1024 >> stages) + 2)]


@pytest.mark.parametrize("tap_set,stages", [(c[1], c[2]) for c in HBF_CASES])
@pytest.mark.parametrize("layout", [LM, FM])
@pytest.mark.parametrize("kind", ["dec", "int"])
def test_hbf_parity(bes, kind, layout, tap_set, stages):
    ob, gb = bes
    cfg = _cascade(kind, tap_set, stages)
    R = 1 << stages
    words = H.oracle().fn[f"hbf_{kind}_state_words"](C.byref(cfg))
    rng = np.random.default_rng(1000 * stages + 10 * tap_set + layout)
    for lanes, frames in hbf_shapes(stages):
        init = rng.standard_normal(size=(words, lanes)).astype(np.float32).view(np.uint32)
        so, sg = init.copy(), init.copy()
        nin, nout = (frames * R, frames) if kind == "dec" else (frames, frames * R)
        for rep in range(2):  # second call continues from the written-back history
            x = adversarial_f32(rng, lanes * nin)
            rco, yo = ob.cfgcall(f"hbf_{kind}_f32", cfg, so, x, (lanes * nout,), np.float32, lanes, frames, layout)
            rcg, yg = gb.cfgcall(f"hbf_{kind}_f32", cfg, sg, x, (lanes * nout,), np.float32, lanes, frames, layout)
            assert rco == 0 and rcg == 0, H.engine().err()
            assert H.ulp_diff_f32(yo, yg).max(initial=0) <= F32_ULP_TOL, (lanes, frames, rep)
            assert H.ulp_diff_f32(so.view(np.float32), sg.view(np.float32)).max(initial=0) <= F32_ULP_TOL


@pytest.mark.parametrize("kind", ["dec", "int"])
def test_hbf_custom_taps_generic_path(bes, kind):
    """Tap counts without a specialised instance (M = 1, 7, 9) take the runtime-M path."""
    ob, gb = bes
    rng = np.random.default_rng(5)
    taps = [(rng.standard_normal(m) * 0.3).astype(np.float32).tolist() for m in (7, 1, 9)]
    cfg = H.hbf_cfg(taps)
    words = H.oracle().fn[f"hbf_{kind}_state_words"](C.byref(cfg))
    lanes, frames, R = 5, 300, 8
    nin, nout = (frames * R, frames) if kind == "dec" else (frames, frames * R)
    for layout in (LM, FM):
        so, sg = np.zeros((words, lanes), np.uint32), np.zeros((words, lanes), np.uint32)
        x = adversarial_f32(rng, lanes * nin)
        _, yo = ob.cfgcall(f"hbf_{kind}_f32", cfg, so, x, (lanes * nout,), np.float32, lanes, frames, layout)
        rc, yg = gb.cfgcall(f"hbf_{kind}_f32", cfg, sg, x, (lanes * nout,), np.float32, lanes, frames, layout)
        assert rc == 0
        assert H.ulp_diff_f32(yo, yg).max() <= F32_ULP_TOL
        assert np.array_equal(so, sg)


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
@pytest.mark.parametrize("layout", [LM, FM])
def test_fir_sym_parity(bes, kind, layout):
    """Same-rate Type I-IV linear-phase FIR (src/hbf.rs:70-138)."""
    ob, gb = bes
    rng = np.random.default_rng(300 + 10 * kind + layout)
    for m in (1, 4, 23, 32):
        cfg = _abi.FirSymF32()
        cfg.kind, cfg.m = kind, m
        for k, v in enumerate((rng.standard_normal(m) * 0.3).astype(np.float32)):
            cfg.taps[k] = v
        words = H.oracle().fn["fir_sym_state_words"](C.byref(cfg))
        for lanes, frames in [(1, 1), (3, 255), (2, 4096), (5, 4097), (17, 9000)]:
            init = rng.standard_normal(size=(words, lanes)).astype(np.float32).view(np.uint32)
            so, sg = init.copy(), init.copy()
            for rep in range(2):
                x = adversarial_f32(rng, lanes * frames)
                rco, yo = ob.cfgcall("fir_sym_f32_process", cfg, so, x, (lanes * frames,), np.float32, lanes, frames, layout)
                rcg, yg = gb.cfgcall("fir_sym_f32_process", cfg, sg, x, (lanes * frames,), np.float32, lanes, frames, layout)
                assert rco == 0 and rcg == 0, H.engine().err()
                assert H.ulp_diff_f32(yo, yg).max(initial=0) <= F32_ULP_TOL
                assert np.array_equal(so, sg)


# ------------------------------------------------------- dds / lowpass / lockin
def test_cossin_parity_all_octants(bes):
    ob, gb = bes
    rng = np.random.default_rng(3)
    p = np.concatenate([
        rng.integers(I32_MIN, I32_MAX, size=200001, dtype=np.int64, endpoint=True).astype(np.int32),
        np.array([0, 1, -1, I32_MIN, I32_MAX, 1 << 29, (1 << 29) - 1, 1 << 30, -(1 << 30), (1 << 15) - 1], np.int32),
        (np.arange(-4096, 4096, dtype=np.int64) * (1 << 19)).astype(np.int32),
    ])
    _, co = ob.cossin(p)
    rc, cg = gb.cossin(p)
    assert rc == 0 and np.array_equal(co, cg)


def test_atan2_parity_all_octants(bes):
    ob, gb = bes
    rng = np.random.default_rng(4)
    edge = np.array([0, 1, -1, 2, 3, I32_MAX, I32_MIN, I32_MIN + 1, 1 << 30, -(1 << 30), (1 << 27) - 1, 1 << 27], np.int64)
    xy = np.concatenate([
        rng.integers(I32_MIN, I32_MAX, size=(3000001, 2), dtype=np.int64, endpoint=True),
        rng.integers(-600, 601, size=(200000, 2)),
        np.stack([np.repeat(edge, edge.size), np.tile(edge, edge.size)], 1),
    ]).astype(np.int32)
    _, ao = ob.atan2(xy)
    rc, ag = gb.atan2(xy)
    assert rc == 0 and np.array_equal(ao, ag)
    rc, e = gb.atan2(np.empty((0, 2), np.int32))
    assert rc == 0 and e.size == 0


@pytest.mark.parametrize("layout", [FM, LM])
def test_dds_parity(bes, layout):
    ob, gb = bes
    rng = np.random.default_rng(21 + layout)
    # from 256 FrameMajor frames the two-threads-per-lane form evaluates cossin through the full-circle table (dds.hip: CosTab<true>)
    for lanes, frames in [(1, 1), (64, 31), (65, 32), (100, 33), (7, 200), (64, 256), (100, 300), (130, 1000), (2049, 257), (1, 4096)]:
        st = rng.integers(0, 1 << 32, size=(2, lanes), dtype=np.uint64).astype(np.uint32)
        so, sg = st.copy(), st.copy()
        _, yo = ob.dds(so, lanes, frames, layout)
        rc, yg = gb.dds(sg, lanes, frames, layout)
        assert rc == 0 and np.array_equal(yo, yg) and np.array_equal(so, sg)


def lowpass_ks(rng, order, cascade):
    ks = []
    for _ in range(cascade):
        k = int(rng.integers(1 << 16, 1 << 29))
        if order == 1:
            ks.append([int(rng.integers(1, I32_MAX))])
        else:  # [k^2 / 2^32, -k * sqrt(2)]  (src/lowpass.rs:37-38), plus arbitrary values
            ks.append([int(k * k >> 32) or 1, -int(k * 1.4142135623730951)])
    if rng.integers(0, 3) == 0:
        ks[0] = rng.integers(I32_MIN, I32_MAX, size=order, dtype=np.int64).tolist()
    return ks


@pytest.mark.parametrize("order", [1, 2])
@pytest.mark.parametrize("cascade", [1, 2, 3, 4])
@pytest.mark.parametrize("layout", [FM, LM])
def test_lowpass_and_lockin_parity(bes, order, cascade, layout):
    ob, gb = bes
    rng = np.random.default_rng(100 * order + 10 * cascade + layout)
    for lanes, frames in [(1, 3), (64, 47), (65, 49), (130, 100), (9, 300)]:
        cfg = H.lockin_cfg(lowpass_ks(rng, order, cascade))
        x = adversarial_i32(rng, lanes * frames)
        # Lowpass cascade alone
        words = 2 * order * cascade
        st = rng.integers(0, 1 << 32, size=(words, lanes), dtype=np.uint64).astype(np.uint32)
        so, sg = st.copy(), st.copy()
        _, yo = ob.cfgcall("lowpass_i32", cfg, so, x, (lanes * frames,), np.int32, lanes, frames, layout)
        rc, yg = gb.cfgcall("lowpass_i32", cfg, sg, x, (lanes * frames,), np.int32, lanes, frames, layout)
        assert rc == 0 and np.array_equal(yo, yg) and np.array_equal(so, sg)
        # Lockin
        words = 2 + 2 * words
        st = rng.integers(0, 1 << 32, size=(words, lanes), dtype=np.uint64).astype(np.uint32)
        so, sg = st.copy(), st.copy()
        for _ in range(2):
            _, yo = ob.cfgcall("lockin_i32_process", cfg, so, x, (lanes * frames * 2,), np.int32, lanes, frames, layout)
            rc, yg = gb.cfgcall("lockin_i32_process", cfg, sg, x, (lanes * frames * 2,), np.int32, lanes, frames, layout)
            assert rc == 0 and np.array_equal(yo, yg) and np.array_equal(so, sg)
        # polar read-outs fused into the same pass; the three entries share one state
        for name, dt in (("lockin_i32_arg", np.int32), ("lockin_i32_norm_sqr", np.int64), ("lockin_i32_arg", np.int32)):
            _, yo = ob.cfgcall(name, cfg, so, x, (lanes * frames,), dt, lanes, frames, layout)
            rc, yg = gb.cfgcall(name, cfg, sg, x, (lanes * frames,), dt, lanes, frames, layout)
            assert rc == 0 and np.array_equal(yo, yg) and np.array_equal(so, sg), name
    # whole 16-frame batches: the multi-wave kernel in both layouts (LaneMajor takes it only for these), 1..4 batches;
    # (3, 8) and (129, 24) stay on the LaneMajor stream kernels
    for lanes, frames in [(3, 8), (70, 16), (129, 24), (130, 48), (64, 64)]:
        cfg = H.lockin_cfg(lowpass_ks(rng, order, cascade))
        x = adversarial_i32(rng, lanes * frames)
        st = rng.integers(0, 1 << 32, size=(2 + 4 * order * cascade, lanes), dtype=np.uint64).astype(np.uint32)
        so, sg = st.copy(), st.copy()
        for name, width, dt in (("lockin_i32_arg", 1, np.int32), ("lockin_i32_process", 2, np.int32),
                                ("lockin_i32_norm_sqr", 1, np.int64), ("lockin_i32_arg", 1, np.int32)):
            _, yo = ob.cfgcall(name, cfg, so, x, (lanes * frames * width,), dt, lanes, frames, layout)
            rc, yg = gb.cfgcall(name, cfg, sg, x, (lanes * frames * width,), dt, lanes, frames, layout)
            assert rc == 0 and np.array_equal(yo, yg) and np.array_equal(so, sg), (name, lanes, frames)


def test_lockin_arg_equals_lockin_then_atan2(bes):
    """The fused entry is the composition the reference writes as `lockin.process(..).arg()`:
    HIP lock-in -> HIP atan2 (two passes) == HIP fused pass, at a size that takes the tiled paths."""
    _, gb = bes
    rng = np.random.default_rng(77)
    lanes, frames = 320, 256
    cfg = H.lockin_cfg([[1 << 22, -(1 << 27)]] * 2)
    x = adversarial_i32(rng, lanes * frames)
    st = np.zeros((2 + 16, lanes), np.uint32)
    st[0] = rng.integers(0, 1 << 32, lanes, dtype=np.uint64).astype(np.uint32)
    st[1] = rng.integers(0, 1 << 32, lanes, dtype=np.uint64).astype(np.uint32)
    for layout in (FM, LM):
        s1, s2, s3 = st.copy(), st.copy(), st.copy()
        rc, iq = gb.cfgcall("lockin_i32_process", cfg, s1, x, (lanes * frames * 2,), np.int32, lanes, frames, layout)
        assert rc == 0
        rc, want = gb.atan2(iq.reshape(-1, 2))
        assert rc == 0
        rc, arg = gb.cfgcall("lockin_i32_arg", cfg, s2, x, (lanes * frames,), np.int32, lanes, frames, layout)
        assert rc == 0 and np.array_equal(arg, np.asarray(want).ravel()) and np.array_equal(s1, s2)
        rc, pw = gb.cfgcall("lockin_i32_norm_sqr", cfg, s3, x, (lanes * frames,), np.int64, lanes, frames, layout)
        z = iq.reshape(-1, 2).astype(np.int64)
        assert rc == 0 and np.array_equal(pw, z[:, 0] * z[:, 0] + z[:, 1] * z[:, 1]) and np.array_equal(s1, s3)


def test_concurrent_streams_distinct_states(bes):
    """Thread-safety contract of the ABI: calls on different streams with distinct state buffers
    are independent (the reference's `&self` config / `&mut S` state split, process.rs:70-80)."""
    import torch

    ob, _ = bes
    e = H.engine()
    rng = np.random.default_rng(99)
    lanes, frames = 4096, 700
    cfg = H.biquad_i32(random_i32_sections(rng, 2, False))
    xs = [adversarial_i32(rng, lanes * frames) for _ in range(4)]
    want = []
    for x in xs:
        so = np.zeros((8, lanes), np.uint32)
        _, y = ob.stream("biquad_i32_df1", cfg, 2, so, x, lanes, frames, FM)
        want.append((y, so))
    streams = [torch.cuda.Stream() for _ in xs]
    xd = [torch.from_numpy(x).cuda() for x in xs]
    yd = [torch.empty_like(t) for t in xd]
    sd = [torch.zeros((8, lanes), dtype=torch.int32, device="cuda") for _ in xs]
    torch.cuda.synchronize()
    for rep in range(3):  # interleave launches over the streams, 3 chunks each
        a, b = rep * 233, min(frames, (rep + 1) * 233 + (1 if rep == 2 else 0))
        b = frames if rep == 2 else b
        for s, x, y, st in zip(streams, xd, yd, sd):
            xv, yv = x.view(frames, lanes)[a:b], y.view(frames, lanes)[a:b]
            assert e.stream("biquad_i32_df1", cfg, 2, st, xv, yv, lanes, b - a, FM, C.c_void_p(s.cuda_stream)) == 0
    torch.cuda.synchronize()
    for (y, so), yg, sg in zip(want, yd, sd):
        assert np.array_equal(yg.cpu().numpy(), y)
        assert np.array_equal(sg.cpu().numpy().view(np.uint32), so)


def test_cossin_atan2_unaligned_buffers_take_the_scalar_path(bes):
    """The vector kernels need 16-byte aligned buffers; a 4-byte offset view must give the same values."""
    import torch

    ob, gb = bes
    e = H.engine()
    rng = np.random.default_rng(77)
    n = 100003
    p = rng.integers(I32_MIN, I32_MAX, size=n, dtype=np.int64, endpoint=True).astype(np.int32)
    _, want = ob.cossin(p)
    buf = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
    buf[1:] = torch.from_numpy(p).cuda()
    out = torch.zeros(2 * n + 1, dtype=torch.int32, device="cuda")
    assert e.fn["cossin_i32"](C.c_void_p(buf.data_ptr() + 4), C.c_void_p(out.data_ptr() + 4), n, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out[1:].cpu().numpy().reshape(-1, 2), want)
    xy = rng.integers(I32_MIN, I32_MAX, size=(n, 2), dtype=np.int64, endpoint=True).astype(np.int32)
    _, want = ob.atan2(xy)
    buf = torch.zeros(2 * n + 1, dtype=torch.int32, device="cuda")
    buf[1:] = torch.from_numpy(xy.reshape(-1)).cuda()
    out = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
    assert e.fn["atan2_i32"](C.c_void_p(buf.data_ptr() + 4), C.c_void_p(out.data_ptr() + 4), n, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(out[1:].cpu().numpy(), want)
