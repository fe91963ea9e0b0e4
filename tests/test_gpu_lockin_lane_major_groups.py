"""LaneMajor lock-in on the multi-wave kernel after round 4 (idsp_amd/csrc/lockin_waves.h): output lines leave in groups of eight
batches per lane (four with an external LO), the roles run in loops of their own, and launches of two or more workgroups per CU
with 2048-8192 frames start staggered.  Against the oracle, bit for bit, outputs and written-back state:
* batch counts around the group length — a single batch, one short of a group, a whole group, one more, several groups plus a
  remainder — so that every flush position of the held vectors is taken, on whole and ragged 64-lane workgroups, two calls on
  one state;
* every bank that runs on the kernel: `[Lowpass<N>; K]` with `Complex<i32>` and `norm_sqr` read-outs, `[Biquad; n]` arms, the
  external-LO forms (i32 lowpass / biquad arms, f32 biquad arms);
* one launch large enough to be staggered (513 workgroups, 2064 frames)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import _harness as H
from tests import _lockin_generic_cases as G

pytestmark = pytest.mark.gpu
LM = H.LM
DEV = "cuda:0"
FRAMES = [16, 112, 128, 144, 208, 272, 400]  # 1, 7, 8, 9, 13, 17, 25 batches of 16 frames
LANES = [64, 70, 129]


def dev(a):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(DEV)


def is_waves(e):
    return e.last_kernel().startswith("lockin_waves_kernel[4 waves per 64 lanes]")


@pytest.mark.parametrize("entry,width,ndt,tdt", [("lockin_i32_process", 2, np.int32, torch.int32), ("lockin_i32_norm_sqr", 1, np.int64, torch.int64)])
@pytest.mark.parametrize("order,cascade", [(2, 2), (1, 1), (2, 4)])
def test_lowpass_arms_every_flush_position(gpu, entry, width, ndt, tdt, order, cascade):
    o, e = H.oracle(), H.engine()
    rng = np.random.default_rng(4100 + 10 * order + cascade + width)
    ks = [[1 << 22, -(1 << 27)][:order] for _ in range(cascade)]
    cfg = H.lockin_cfg(ks)
    for lanes in LANES:
        for frames in FRAMES:
            st = rng.integers(0, 1 << 32, size=(2 + 4 * order * cascade, lanes), dtype=np.uint64).astype(np.uint32)
            so, sg = st.copy(), dev(st)
            for rep in range(2):
                x = rng.integers(-(1 << 31), (1 << 31) - 1, size=lanes * frames, dtype=np.int64).astype(np.int32)
                yo = np.empty(lanes * frames * width, ndt)
                yg = torch.full((lanes * frames * width,), -77, dtype=tdt, device=DEV)
                assert o.cfgcall(entry, cfg, so, x, yo, lanes, frames, LM) == 0
                assert e.cfgcall(entry, cfg, sg, dev(x), yg, lanes, frames, LM) == 0, e.err()
                torch.cuda.synchronize()
                assert is_waves(e), e.last_kernel()
                assert np.array_equal(yg.cpu().numpy(), yo), (entry, lanes, frames, rep)
                assert np.array_equal(sg.cpu().numpy().view(np.uint32), so), (entry, lanes, frames, rep)


def test_biquad_arms_and_external_lo_every_flush_position(gpu):
    o, e = H.oracle(), H.engine()
    rng = np.random.default_rng(4200)
    arr, _ = G.sections_i32(2, rng)
    arrf, _ = G.sections_f32(2, rng)
    cfg = H.lockin_cfg([[1 << 22, -(1 << 27)], [1 << 21, -(1 << 26)]])
    lp_words = H.oracle().fn["lockin_state_words"](C.byref(cfg)) - 2
    for lanes in (64, 70):
        for frames in FRAMES:
            x = [rng.integers(-(1 << 28), 1 << 28, lanes * frames, dtype=np.int32) for _ in range(2)]
            lo = [rng.integers(-(1 << 31), (1 << 31) - 1, lanes * frames * 2, dtype=np.int64).astype(np.int32) for _ in range(2)]
            xf = [rng.standard_normal(lanes * frames).astype(np.float32) for _ in range(2)]
            lof = [rng.standard_normal(lanes * frames * 2).astype(np.float32) for _ in range(2)]
            cases = [("lockin_i32_biquad_process", arr, 2, np.zeros((2 + 16, lanes), np.uint32), x, None, np.int32),
                     ("lockin_i32_lo_process", cfg, None, rng.integers(0, 1 << 32, (lp_words, lanes), dtype=np.uint64).astype(np.uint32), x, lo, np.int32),
                     ("lockin_i32_biquad_lo_process", arr, 2, rng.integers(-(1 << 20), 1 << 20, (16, lanes)).astype(np.int32).view(np.uint32), x, lo, np.int32),
                     ("lockin_f32_biquad_lo_process", arrf, 2, rng.standard_normal((16, lanes)).astype(np.float32).view(np.uint32), xf, lof, np.float32)]
            cases[0][3][:2] = rng.integers(0, 1 << 32, (2, lanes), dtype=np.uint64).astype(np.uint32)
            for name, c, n, st0, xs, los, ydt in cases:
                so, sg = st0.copy(), dev(st0)
                for rep in range(2):
                    yo = np.empty(lanes * frames * 2, ydt)
                    yg = torch.full((lanes * frames * 2,), -77, dtype=torch.float32 if ydt == np.float32 else torch.int32, device=DEV)
                    if los is None:
                        rco = o.stream(name, c, n, so, xs[rep], yo, lanes, frames, LM)
                        rcg = e.stream(name, c, n, sg, dev(xs[rep]), yg, lanes, frames, LM)
                    else:
                        rco = G.call_lo(o, name, c, n, so, xs[rep], los[rep], yo, lanes, frames, LM, False)
                        rcg = G.call_lo(e, name, c, n, sg, dev(xs[rep]), dev(los[rep]), yg, lanes, frames, LM, True)
                    torch.cuda.synchronize()
                    assert rco == 0 and rcg == 0, e.err()
                    assert is_waves(e), (name, e.last_kernel())
                    assert np.array_equal(yg.cpu().numpy().view(np.uint32), yo.view(np.uint32)), (name, lanes, frames, rep)
                    assert np.array_equal(sg.cpu().numpy().view(np.uint32), so), (name, lanes, frames, rep)


def test_a_staggered_launch(gpu):
    """513 workgroups (the last one ragged) x 2064 frames: the launcher's start-up stagger is on (>= 512 workgroups, 2048-8192 frames);
    the oracle does not know about it."""
    o, e = H.oracle(), H.engine()
    rng = np.random.default_rng(4300)
    lanes, frames = 512 * 64 + 40, 2064
    cfg = H.lockin_cfg([[1 << 22, -(1 << 27)], [1 << 21, -(1 << 26)]])
    st = rng.integers(0, 1 << 32, size=(2 + 4 * 2 * 2, lanes), dtype=np.uint64).astype(np.uint32)
    x = rng.integers(-(1 << 31), (1 << 31) - 1, size=lanes * frames, dtype=np.int64).astype(np.int32)
    so, sg = st.copy(), dev(st)
    yo = np.empty(lanes * frames * 2, np.int32)
    yg = torch.full((lanes * frames * 2,), -77, dtype=torch.int32, device=DEV)
    assert o.cfgcall("lockin_i32_process", cfg, so, x, yo, lanes, frames, LM) == 0
    assert e.cfgcall("lockin_i32_process", cfg, sg, dev(x), yg, lanes, frames, LM) == 0, e.err()
    torch.cuda.synchronize()
    assert is_waves(e), e.last_kernel()
    assert np.array_equal(yg.cpu().numpy(), yo) and np.array_equal(sg.cpu().numpy().view(np.uint32), so)


@pytest.mark.parametrize("entry,width,ndt,tdt", [("lockin_i32_process", 2, np.int32, torch.int32), ("lockin_i32_arg", 1, np.int32, torch.int32),
                                                  ("lockin_i32_norm_sqr", 1, np.int64, torch.int64)])
def test_rows_that_are_not_whole_batches(gpu, entry, width, ndt, tdt):
    """LaneMajor rows of 32 + 4 k frames: the whole 16-frame batches of every row on the multi-wave kernel at the call's row pitch, the
    last frames % 16 on a stream kernel behind it; two calls on one state; frame counts off the 4-frame grid stay on the stream kernels."""
    o, e = H.oracle(), H.engine()
    rng = np.random.default_rng(4400 + width + (ndt == np.int64))
    cfg = H.lockin_cfg([[1 << 22, -(1 << 27)], [1 << 21, -(1 << 26)]])
    for lanes, frames in [(64, 36), (70, 100), (129, 1000), (64, 2076), (200, 44), (64, 28), (64, 34)]:
        st = rng.integers(0, 1 << 32, size=(18, lanes), dtype=np.uint64).astype(np.uint32)
        so, sg = st.copy(), dev(st)
        for rep in range(2):
            x = rng.integers(-(1 << 31), (1 << 31) - 1, size=lanes * frames, dtype=np.int64).astype(np.int32)
            yo = np.empty(lanes * frames * width, ndt)
            yg = torch.full((lanes * frames * width,), -77, dtype=tdt, device=DEV)
            assert o.cfgcall(entry, cfg, so, x, yo, lanes, frames, LM) == 0
            assert e.cfgcall(entry, cfg, sg, dev(x), yg, lanes, frames, LM) == 0, e.err()
            torch.cuda.synchronize()
            split = frames >= 32 and frames % 4 == 0
            assert e.last_kernel().startswith("lockin_waves_kernel + stream kernel (last frames % 16)") == split, (e.last_kernel(), frames)
            assert np.array_equal(yg.cpu().numpy(), yo), (entry, lanes, frames, rep)
            assert np.array_equal(sg.cpu().numpy().view(np.uint32), so), (entry, lanes, frames, rep)


def test_biquad_arm_rows_that_are_not_whole_batches(gpu):
    """`Lockin<[Biquad; n]>`, phase form: the same row split as for the lowpass arms."""
    o, e = H.oracle(), H.engine()
    rng = np.random.default_rng(4500)
    for n in (1, 3):
        arr, _ = G.sections_i32(n, rng)
        for lanes, frames in [(64, 36), (70, 100), (129, 1000), (64, 34)]:
            st = np.zeros((2 + 8 * n, lanes), np.uint32)
            st[:2] = rng.integers(0, 1 << 32, (2, lanes), dtype=np.uint64).astype(np.uint32)
            st[2:] = rng.integers(-(1 << 20), 1 << 20, (8 * n, lanes)).astype(np.int32).view(np.uint32)
            so, sg = st.copy(), dev(st)
            for rep in range(2):
                x = rng.integers(-(1 << 28), 1 << 28, lanes * frames, dtype=np.int32)
                yo = np.empty(lanes * frames * 2, np.int32)
                yg = torch.full((lanes * frames * 2,), -77, dtype=torch.int32, device=DEV)
                assert o.stream("lockin_i32_biquad_process", arr, n, so, x, yo, lanes, frames, LM) == 0
                assert e.stream("lockin_i32_biquad_process", arr, n, sg, dev(x), yg, lanes, frames, LM) == 0, e.err()
                torch.cuda.synchronize()
                assert e.last_kernel().startswith("lockin_waves_kernel + stream kernel (last frames % 16)") == (frames % 4 == 0), e.last_kernel()
                assert np.array_equal(yg.cpu().numpy(), yo) and np.array_equal(sg.cpu().numpy().view(np.uint32), so), (n, lanes, frames, rep)


def test_external_lo_rows_that_are_not_whole_batches(gpu):
    """the external-LO forms (i32 lowpass arms, i32 / f32 biquad arms): whole batches of every row on the multi-wave kernel, the rest on the
    stream kernel at the call's row pitch (x, LO and y alike)"""
    o, e = H.oracle(), H.engine()
    rng = np.random.default_rng(4600)
    arr, _ = G.sections_i32(2, rng)
    arrf, _ = G.sections_f32(2, rng)
    cfg = H.lockin_cfg([[1 << 22, -(1 << 27)], [1 << 21, -(1 << 26)]])
    lp_words = H.oracle().fn["lockin_state_words"](C.byref(cfg)) - 2
    for lanes, frames in [(64, 36), (70, 100), (129, 1000)]:
        x = [rng.integers(-(1 << 28), 1 << 28, lanes * frames, dtype=np.int32) for _ in range(2)]
        lo = [rng.integers(-(1 << 31), (1 << 31) - 1, lanes * frames * 2, dtype=np.int64).astype(np.int32) for _ in range(2)]
        xf = [rng.standard_normal(lanes * frames).astype(np.float32) for _ in range(2)]
        lof = [rng.standard_normal(lanes * frames * 2).astype(np.float32) for _ in range(2)]
        cases = [("lockin_i32_lo_process", cfg, None, rng.integers(0, 1 << 32, (lp_words, lanes), dtype=np.uint64).astype(np.uint32), x, lo, np.int32),
                 ("lockin_i32_biquad_lo_process", arr, 2, rng.integers(-(1 << 20), 1 << 20, (16, lanes)).astype(np.int32).view(np.uint32), x, lo, np.int32),
                 ("lockin_f32_biquad_lo_process", arrf, 2, rng.standard_normal((16, lanes)).astype(np.float32).view(np.uint32), xf, lof, np.float32)]
        for name, c, n, st0, xs, los, ydt in cases:
            so, sg = st0.copy(), dev(st0)
            for rep in range(2):
                yo = np.empty(lanes * frames * 2, ydt)
                yg = torch.full((lanes * frames * 2,), -77, dtype=torch.float32 if ydt == np.float32 else torch.int32, device=DEV)
                rco = G.call_lo(o, name, c, n, so, xs[rep], los[rep], yo, lanes, frames, LM, False)
                rcg = G.call_lo(e, name, c, n, sg, dev(xs[rep]), dev(los[rep]), yg, lanes, frames, LM, True)
                torch.cuda.synchronize()
                assert rco == 0 and rcg == 0, e.err()
                assert np.array_equal(yg.cpu().numpy().view(np.uint32), yo.view(np.uint32)), (name, lanes, frames, rep, e.last_kernel())
                assert np.array_equal(sg.cpu().numpy().view(np.uint32), so), (name, lanes, frames, rep)
