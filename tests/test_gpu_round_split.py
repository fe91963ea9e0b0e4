"""FRAME_MAJOR lane counts just above a whole number of rounds of 256 lane blocks (65540, 69632, 131076 lanes ...): the
launcher runs the whole rounds on the LDS-DMA kernel and the remainder BESIDE them, on a second stream, on the staged
single-wave kernel (idsp_amd/csrc/lane_stream.h, "whole rounds + remainder") — both as lane blocks of the caller's tensors,
with the state planes (and the coefficient planes of a `ByLane` bank) at the call's pitch.  The oracle decides, outputs and
written-back state, out of place and in place, dense rows and a lane block of a wider tensor with the neighbours untouched;
and the call must still be ordered on the caller's stream (the result is read back right after it on that stream).
Reference semantics: any N in `Lanes<C>`, lanes independent (dsp-process/src/compose.rs:468-494)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import _bylane_cases as B
from tests import _harness as H
from tests import test_gpu_frame_major_staged as FMS
from tests._backends import GpuBackend, OracleBackend
from tests.test_gpu_pitch import cases

pytestmark = pytest.mark.gpu
SPLIT = " + stream_frame_major_staged (remainder, second stream)<"  # after stream_frame_major_lds (rows off the 64-byte grid) or _sweep (on it)


def is_split(k):
    return k.startswith(("stream_frame_major_lds" + SPLIT, "stream_frame_major_sweep" + SPLIT))


def kernel_of(eng):
    return eng.fn["last_kernel"]().decode()


def test_remainder_beside_the_whole_rounds(gpu):
    rng = np.random.default_rng(401)
    cs = [c for c in cases(rng) if c[4] != np.float64 and c[2] <= 2]  # single-pass, 4-byte, LDS-eligible
    # (lanes, frames, pitch, lane offset): remainders of 4, 4096, 260 (ragged staged wave), 20480 lanes; two rounds + 8
    shapes = [(65540, 21, 65540, 0), (69632, 40, 69632, 0), (65796, 17, 65800, 4), (86016, 16, 86016, 0), (131080, 19, 131136, 32)]
    for i, (lanes, frames, pitch, off) in enumerate(shapes):
        for j, (op, cfg, n, words, dt) in enumerate(cs):
            if (i + j) % 3:
                continue
            FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, bool((i + j) & 1), off=off)
            assert is_split(kernel_of(gpu)), (op, lanes, kernel_of(gpu))
    # a remainder above the limit, and a whole number of rounds, stay one launch
    op, cfg, n, words, dt = cs[0]
    FMS.run_case(gpu, op, cfg, n, words, dt, rng, 65536 + 24576, 16, 65536 + 24576, False)
    assert kernel_of(gpu).startswith("stream_frame_major_sweep[2 blocks/workgroup]<"), kernel_of(gpu)


def test_bylane_bank_moves_its_coefficient_planes_with_the_remainder(gpu):
    rng = np.random.default_rng(402)
    ob, gb = OracleBackend(), GpuBackend()
    lanes, frames = 65536 + 516, 24
    for op, dt, frac, words, clamp in (("biquad_i32_df1", np.int32, 29, 4, False), ("biquad_f32_df2t_clamp", np.float32, None, 2, True)):
        coef = B.coef_planes(rng, dt, 1, lanes, clamp, frac if frac is not None else 0)
        x = B.samples(rng, dt, lanes * frames)
        so, sg = np.zeros((words, lanes), np.uint32), np.zeros((words, lanes), np.uint32)
        rc, yo = ob.bylane(op, coef, frac, 1, so, x, lanes, frames, H.FM)
        assert rc == 0
        rc, yg = gb.bylane(op, coef, frac, 1, sg, x, lanes, frames, H.FM)
        assert rc == 0 and is_split(kernel_of(gpu)), kernel_of(gpu)
        assert np.array_equal(yo.view(np.uint32), yg.view(np.uint32)) and np.array_equal(so, sg), op


def test_call_stays_ordered_on_the_callers_stream(gpu):
    """Launch on a non-default stream, then read the result back ON THAT STREAM without a device-wide sync: the remainder
    ran on the library's second stream and must have been joined."""
    rng = np.random.default_rng(403)
    o = H.oracle()
    op, cfg, n, words, dt = [c for c in cases(rng) if c[0] == "biquad_i32_df1" and c[2] == 1][0]
    lanes, frames = 65536 + 8192, 64
    xh = rng.integers(-(1 << 30), 1 << 30, size=(frames, lanes), dtype=np.int32)
    want, so = np.empty_like(xh), np.zeros((4, lanes), np.uint32)
    assert o.stream(op, cfg, 1, so, xh, want, lanes, frames, H.FM) == 0
    s = torch.cuda.Stream()
    host = torch.empty((frames, lanes), dtype=torch.int32).pin_memory()
    with torch.cuda.stream(s):
        xd = torch.from_numpy(xh).to("cuda", non_blocking=False)
        yd = torch.full_like(xd, -1)
        sd = torch.zeros((4, lanes), dtype=torch.int32, device="cuda")
        for _ in range(3):  # back-to-back calls reuse the side stream and its events
            sd.zero_()
            assert gpu.stream(op, cfg, 1, sd, xd, yd, lanes, frames, H.FM, C.c_void_p(s.cuda_stream)) == 0
        host.copy_(yd, non_blocking=True)
    s.synchronize()
    assert is_split(kernel_of(gpu))
    assert np.array_equal(host.numpy(), want) and np.array_equal(sd.cpu().numpy().view(np.uint32), so)
