"""Cic decimator / interpolator on the HIP path through the C ABI vs the CPU oracle: bit-exact
outputs and written-back state for i32 and i64, both layouts, vector (R % 4 == 0) and scalar
chunk widths, ragged lane counts, continuation from arbitrary state."""
import os

import numpy as np
import pytest

from idsp_amd import _abi
from tests import _cic_cases as K
from tests import _harness as H
from tests._backends import GpuBackend, OracleBackend

pytestmark = pytest.mark.gpu
MISALIGNED = bool(os.environ.get("IDSP_TEST_MISALIGN"))


@pytest.fixture(scope="module")
def bes(gpu):
    return OracleBackend(), GpuBackend()


@pytest.mark.parametrize("dtype", [np.int32, np.int64], ids=["i32", "i64"])
@pytest.mark.parametrize("kind", ["dec", "int"])
@pytest.mark.parametrize("layout", [K.FM, K.LM])
def test_cic_parity(bes, kind, dtype, layout):
    ob, gb = bes
    rng = np.random.default_rng(40 + layout + (2 if kind == "dec" else 0) + (4 if dtype == np.int64 else 0))
    shapes = [(1, 1), (3, 17), (64, 9), (65, 12), (257, 5), (300, 40), (1024, 8)]
    for (n, m, rate), (lanes, frames) in zip(K.CONFIGS, shapes + shapes):
        cfg = _abi.Cic(n, m, rate)
        R = rate + 1
        words = K.state_words(gb, cfg, dtype)
        assert words == K.state_words(ob, cfg, dtype)
        init = K.random_state(rng, words, lanes)
        so, sg = init.copy(), init.copy()
        for part in range(2):
            x = K.samples(rng, dtype, lanes * frames * (R if kind == "dec" else 1))
            rco, yo = K.run(ob, kind, dtype, cfg, so, x, lanes, frames, layout)
            rcg, yg = K.run(gb, kind, dtype, cfg, sg, x, lanes, frames, layout)
            assert rco == 0 and rcg == 0, H.engine().err()
            assert np.array_equal(yo, yg), (n, m, rate, lanes, frames)
            assert np.array_equal(so, sg)


def test_cic_errors_and_helpers(bes):
    import ctypes as C

    ob, gb = bes
    for n, m, rate in K.CONFIGS:
        cfg = _abi.Cic(n, m, rate)
        for h in ("cic_gain", "cic_gain_log2", "cic_response_length"):
            assert gb.helper(h, C.byref(cfg)) == ob.helper(h, C.byref(cfg))
    x = np.zeros(8, np.int32)
    st = np.zeros((16, 1), np.uint32)
    assert K.run(gb, "dec", np.int32, _abi.Cic(0, 1, 1), st, x, 1, 4, K.LM)[0] == -1 and "order" in H.engine().err()
    assert K.run(gb, "int", np.int32, _abi.Cic(3, 5, 1), st, x, 1, 4, K.LM)[0] == -1 and "cic.rs:36" in H.engine().err()


@pytest.mark.parametrize("dtype", [np.int32, np.int64], ids=["i32", "i64"])
@pytest.mark.parametrize("kind", ["dec", "int"])
def test_cic_lane_major_tile_kernels(bes, kind, dtype):
    """Whole waves + vector chunk widths take the LANE_MAJOR tile kernels (64 lanes x 16 vectors through LDS);
    frame counts that are and are not whole tiles, continuation, every chunk width with a tile kernel."""
    ob, gb = bes
    rng = np.random.default_rng(90 + (1 if kind == "dec" else 0) + (2 if dtype == np.int64 else 0))
    epv = 16 // np.dtype(dtype).itemsize
    for vpc, lanes, frames in [(1, 64, 40), (2, 128, 37), (4, 192, 64), (8, 64, 19), (16, 128, 8), (4, 64, 3)]:
        R = vpc * epv
        cfg = _abi.Cic(int(rng.integers(1, 7)), int(rng.integers(1, 5)), R - 1)
        words = K.state_words(gb, cfg, dtype)
        init = K.random_state(rng, words, lanes)
        so, sg = init.copy(), init.copy()
        for part in range(2):
            x = K.samples(rng, dtype, lanes * frames * (R if kind == "dec" else 1))
            rco, yo = K.run(ob, kind, dtype, cfg, so, x, lanes, frames, K.LM)
            rcg, yg = K.run(gb, kind, dtype, cfg, sg, x, lanes, frames, K.LM)
            assert rco == 0 and rcg == 0, H.engine().err()
            assert np.array_equal(yo, yg), (vpc, lanes, frames, cfg.order, cfg.comb_delay)
            assert np.array_equal(so, sg)


@pytest.mark.parametrize("dtype", [np.int32, np.int64], ids=["i32", "i64"])
@pytest.mark.parametrize("kind", ["dec", "int"])
def test_cic_wave_per_lane(bes, kind, dtype):
    """LANE_MAJOR calls with whole 16-byte pieces per chunk and 64 or more chunks take the wave-per-lane kernels
    (cic_ring.h: frames integrated in parallel, a scan over the wave): every piece count, every order and comb delay,
    whole and ragged blocks, FAST rounds in the middle (5 blocks and more), continuation from the written-back state
    (chunked == whole), lane counts that are not multiples of anything."""
    ob, gb = bes
    rng = np.random.default_rng(140 + (1 if dtype == np.int64 else 0) + (2 if kind == "int" else 0))
    epv = 16 // np.dtype(dtype).itemsize
    shapes = [(1, 3, 64), (2, 5, 65), (4, 7, 127), (8, 2, 128), (4, 33, 200), (2, 1, 449), (1, 9, 64 * 7), (8, 3, 64 * 6 + 1), (4, 4, 1000)]
    orders = [1, 2, 3, 4, 5, 6, 3, 3, 3]
    for (ppt, lanes, frames), n in zip(shapes, orders):
        R = ppt * epv
        per = R if kind == "dec" else 1
        cfg = _abi.Cic(n, int(rng.integers(1, 5)), R - 1)
        words = K.state_words(gb, cfg, dtype)
        init = K.random_state(rng, words, lanes)
        so, sg, sw = init.copy(), init.copy(), init.copy()
        xs, ys = [], []
        for part in range(2):
            x = K.samples(rng, dtype, lanes * frames * per)
            xs.append(x.reshape(lanes, frames * per))
            rco, yo = K.run(ob, kind, dtype, cfg, so, x, lanes, frames, K.LM)
            rcg, yg = K.run(gb, kind, dtype, cfg, sg, x, lanes, frames, K.LM)
            assert rco == 0 and rcg == 0, H.engine().err()
            if not MISALIGNED:  # tests/test_gpu_misaligned.py replays this suite off the 16-byte grid: lane-per-thread kernels there
                assert H.engine().fn["last_kernel"]().decode().startswith(f"cic_{kind}_ring[LaneMajor]")
            assert np.array_equal(yo, yg), (ppt, lanes, frames, cfg.order, cfg.comb_delay, part)
            assert np.array_equal(so, sg), (ppt, lanes, frames, cfg.order, cfg.comb_delay, part)
            ys.append(np.asarray(yg).reshape(lanes, -1))
        # both parts in one call from the same initial state: same outputs, same final state
        xw = np.ascontiguousarray(np.concatenate(xs, axis=1)).reshape(-1)
        rcw, yw = K.run(gb, kind, dtype, cfg, sw, xw, lanes, 2 * frames, K.LM)
        assert rcw == 0 and np.array_equal(sw, sg)
        assert np.array_equal(np.asarray(yw).reshape(lanes, -1), np.concatenate(ys, axis=1))


@pytest.mark.parametrize("kind", ["dec", "int"])
def test_cic_wave_per_lane_extremes(bes, kind):
    """Wrapping: full-scale inputs through order 6 at rate 32 overflow every integrator many times over."""
    ob, gb = bes
    for dtype, ppt in ((np.int32, 8), (np.int64, 8)):
        epv = 16 // np.dtype(dtype).itemsize
        R, lanes, frames = ppt * epv, 2, 256
        cfg = _abi.Cic(6, 4, R - 1)
        words = K.state_words(gb, cfg, dtype)
        info = np.iinfo(dtype)
        x = np.full(lanes * frames * (R if kind == "dec" else 1), info.max, dtype)
        x[1::3] = info.min
        so = np.full((words, lanes), 0xFFFFFFFF, np.uint32)
        sg = so.copy()
        rco, yo = K.run(ob, kind, dtype, cfg, so, x, lanes, frames, K.LM)
        rcg, yg = K.run(gb, kind, dtype, cfg, sg, x, lanes, frames, K.LM)
        assert rco == 0 and rcg == 0
        assert np.array_equal(yo, yg) and np.array_equal(so, sg)


@pytest.mark.parametrize("dtype", [np.int32, np.int64], ids=["i32", "i64"])
def test_cic_interpolator_wave_per_lane_frame_major(bes, dtype):
    """FRAME_MAJOR interpolators over whole groups of 16 lanes (cic_int_ring_fm: 16 waves per workgroup, the frame rows
    of all 16 lanes stored as contiguous runs): every piece count, ragged last blocks, continuation; other lane counts
    stay on the lane-per-thread kernel with the same results."""
    ob, gb = bes
    rng = np.random.default_rng(150 + (1 if dtype == np.int64 else 0))
    epv = 16 // np.dtype(dtype).itemsize
    shapes = [(1, 16, 64), (2, 48, 65), (4, 32, 130), (8, 16, 64 * 3 + 7), (4, 160, 256), (4, 24, 100)]
    for (ppt, lanes, frames), n in zip(shapes, [1, 2, 3, 6, 3, 4]):
        R = ppt * epv
        cfg = _abi.Cic(n, int(rng.integers(1, 5)), R - 1)
        words = K.state_words(gb, cfg, dtype)
        init = K.random_state(rng, words, lanes)
        so, sg = init.copy(), init.copy()
        for part in range(2):
            x = K.samples(rng, dtype, lanes * frames)
            rco, yo = K.run(ob, "int", dtype, cfg, so, x, lanes, frames, K.FM)
            rcg, yg = K.run(gb, "int", dtype, cfg, sg, x, lanes, frames, K.FM)
            assert rco == 0 and rcg == 0, H.engine().err()
            want = "cic_int_ring[FrameMajor]" if lanes % 16 == 0 and not MISALIGNED else "cic_int_kernel"
            assert H.engine().fn["last_kernel"]().decode().startswith(want)
            assert np.array_equal(yo, yg), (ppt, lanes, frames, cfg.order, cfg.comb_delay, part)
            assert np.array_equal(so, sg), (ppt, lanes, frames, cfg.order, cfg.comb_delay, part)
