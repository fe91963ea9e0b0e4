"""FRAME_MAJOR at up to 24576 lanes and from 512 frames: the compute + mover pair kernel (`stream_frame_major_pair`,
idsp_amd/csrc/lane_stream.h, round 6 — one wave walks its 32 lanes' columns of a tile while the other brings the next tile in by
LDS-DMA and takes the previous one out).  Every 4-byte biquad-family entry the kernel takes (cheap sections), lane counts that leave a
partial last workgroup, frame counts around the 128-frame tiles (whole tiles, a short last tile, one frame more than four tiles), padded
rows, a lane block of a wider tensor, out of place and in place, against the oracle bit for bit (outputs, written-back state, untouched
neighbours); the kernel taken is asserted through `idsp_last_kernel()`.  Anything it does not take (long chains, 8-byte samples, rows
off the 16-byte grid, more lanes, fewer frames) must still land on the kernels of rounds 2-5 with the same result.
Reference semantics: every frame holds one sample per lane (`View<FrameMajor>`, dsp-process/src/view.rs:10-17), lanes are independent
filters (`Lanes::process`, dsp-process/src/compose.rs:468-494), chunked calls continue the stream (process.rs:122-141)."""
import numpy as np
import pytest

from tests import test_gpu_frame_major_staged as FMS
from tests.test_gpu_pitch import cases

pytestmark = pytest.mark.gpu
PAIR = "stream_frame_major_pair["
# (lanes, frames, pitch, lane offset)
SHAPES = [(32, 512, 32, 0), (64, 513, 64, 0), (100, 767, 112, 0), (4, 600, 16, 0), (2048, 1024, 2048, 0), (4100, 530, 4112, 0), (1000, 1280, 2048, 512),
          (16384, 520, 16384, 0)]  # pitches and lane offsets on the 64-byte grid (16 lanes)


def kernel_of(eng):
    return eng.fn["last_kernel"]().decode()


def takes(op, n, dt):
    """True / False: the name `idsp_last_kernel()` must (not) report; None: not pinned (chains of more than 4 sections run as passes of up to 4 and the
    name is that of the last, short pass).  The kernel takes cheap 4-byte processors (COST <= 60, dispatch_thresholds.h): single i32 DF1 / dither sections
    with or without clamp, f32 chains of up to two sections; not the wide-state section, not chains beyond, not f64."""
    if dt == np.float64 or "wide" in op:
        return False
    if n > 4:
        return None
    if dt == np.float32:
        return n <= 2
    return n == 1


def test_every_cheap_biquad_entry_on_the_pair_kernel(gpu):
    rng = np.random.default_rng(601)
    seen = 0
    for op, cfg, n, words, dt in cases(rng):
        for i, (lanes, frames, pitch, off) in enumerate(SHAPES):
            if lanes > 5000 and takes(op, n, dt) is not True:
                continue
            inplace = bool(i & 1)
            FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, inplace, off=off)
            k = kernel_of(gpu)
            if takes(op, n, dt) is True:
                assert k.startswith(PAIR), (op, n, lanes, frames, k)
                seen += 1
            elif takes(op, n, dt) is False:
                assert not k.startswith(PAIR), (op, n, lanes, frames, k)
    assert seen >= 40


def test_limits_of_the_pair_kernel(gpu):
    """one 16-lane group more than 768 workgroups, one frame fewer than four tiles, rows off the 16-byte / 64-byte grid: the kernels of rounds 2-5"""
    rng = np.random.default_rng(602)
    op, cfg, n, words, dt = [c for c in cases(rng) if c[0] == "biquad_i32_df1" and c[2] == 1][0]
    FMS.run_case(gpu, op, cfg, n, words, dt, rng, 16384, 512, 16384, False)
    assert kernel_of(gpu).startswith(PAIR), kernel_of(gpu)
    FMS.run_case(gpu, op, cfg, n, words, dt, rng, 24576, 512, 24576, True)
    assert kernel_of(gpu).startswith(PAIR), kernel_of(gpu)
    FMS.run_case(gpu, op, cfg, n, words, dt, rng, 24592, 512, 24592, False)
    assert kernel_of(gpu).startswith("stream_frame_major_sweep["), kernel_of(gpu)
    FMS.run_case(gpu, op, cfg, n, words, dt, rng, 16388, 512, 16388, False)  # rows off the 64-byte grid
    assert kernel_of(gpu).startswith("stream_frame_major_staged["), kernel_of(gpu)
    FMS.run_case(gpu, op, cfg, n, words, dt, rng, 16384, 511, 16384, False)
    assert kernel_of(gpu).startswith("stream_frame_major_staged["), kernel_of(gpu)
    FMS.run_case(gpu, op, cfg, n, words, dt, rng, 1001, 600, 1001, False)
    assert not kernel_of(gpu).startswith(PAIR), kernel_of(gpu)
    FMS.run_case(gpu, op, cfg, n, words, dt, rng, 4100, 600, 4100, False)  # rows off the 64-byte grid: the staged kernel's plain accesses win there
    assert kernel_of(gpu).startswith("stream_frame_major_staged["), kernel_of(gpu)


def test_chunked_calls_continue_the_stream(gpu):
    """two calls of 600 + 700 frames == one call of 1300 frames (state written back by the compute wave)"""
    import ctypes as C

    import torch

    from tests import _harness as H
    from tests.test_gpu_pitch import DEV, init_state, p, sample, tdtype

    rng = np.random.default_rng(603)
    for op, cfg, n, words, dt in cases(rng):
        if takes(op, n, dt) is not True:
            continue
        lanes, f1, f2 = 96, 600, 700
        xh = sample(rng, dt, lanes * (f1 + f2)).reshape(f1 + f2, lanes)
        st0 = init_state(rng, dt, words * n, lanes)
        o = H.oracle()
        want, so = np.empty_like(xh), st0.copy()
        assert o.stream(op, cfg, n, so, xh, want, lanes, f1 + f2, H.FM) == 0
        xd = torch.from_numpy(xh).to(DEV)
        yd = torch.empty_like(xd)
        sg = torch.from_numpy(st0.view(np.int32)).to(DEV)
        esz = xd.element_size()
        for a, b in ((0, f1), (f1, f1 + f2)):
            rc = gpu.fn[op](C.cast(cfg, C.c_void_p), n, p(sg), C.c_void_p(xd.data_ptr() + a * lanes * esz), C.c_void_p(yd.data_ptr() + a * lanes * esz), lanes, b - a, H.FM, None)
            assert rc == 0 and kernel_of(gpu).startswith(PAIR), (op, kernel_of(gpu))
        torch.cuda.synchronize()
        assert np.array_equal(yd.cpu().numpy().view(np.uint8), want.view(np.uint8)), op
        assert np.array_equal(sg.cpu().numpy().view(np.uint32), so), op


def test_remainder_of_a_long_call_beside_the_whole_rounds(gpu):
    """65536 + 2064 lanes x 600 frames: the whole round on the sweep kernel, the 2064 lanes beside it (second stream, lane_stream.h) stay on the STAGED kernel even
    in a long call — beside the sweep kernel the pair kernel's remainder costs 6-10 % — every output and the state against the oracle."""
    rng = np.random.default_rng(604)
    op, cfg, n, words, dt = [c for c in cases(rng) if c[0] == "biquad_i32_df1" and c[2] == 1][0]
    lanes = 65536 + 2064
    FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, 600, lanes, False)
    assert kernel_of(gpu).startswith("stream_frame_major_sweep + stream_frame_major_staged (remainder, second stream)<"), kernel_of(gpu)
