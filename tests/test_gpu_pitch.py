"""`<entry>_pitch` twins of the biquad family (include/idsp_hip.h): explicit row pitches of x and y.
  * pitch == dense is bit-identical to the plain entry;
  * padded LANE_MAJOR rows / a lane block of a wider FRAME_MAJOR tensor give the oracle's dense result on the valid
    elements and leave the padding untouched;
  * a pitch shorter than a row and an in-place call with two different pitches are IDSP_EINVAL."""
import ctypes as C

import numpy as np
import pytest
import torch

from idsp_amd import _abi
from tests import _harness as H
from tests import _bylane_cases as B

pytestmark = pytest.mark.gpu
FM, LM = H.FM, H.LM
DEV = "cuda:0"
SENT = -1234567


def p(t):
    return C.c_void_p(t.data_ptr())


def tdtype(dt):
    return {np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}[np.dtype(dt)]


def cases(rng):
    """(entry, cfg array, sections, state words per section, dtype)"""
    ri = [(rng.integers(-(1 << 29), 1 << 29, size=5).tolist(), 29) for _ in range(6)]
    rc = [(r[0], 29, 77, -(1 << 28), 1 << 28) for r in ri]
    rf = [(rng.standard_normal(5) * 0.3).tolist() for _ in range(6)]
    rfc = [(r, 0.01, -0.5, 0.6) for r in rf]
    return [
        ("biquad_i32_df1", H.biquad_i32(ri[:1]), 1, 4, np.int32), ("biquad_i32_df1", H.biquad_i32(ri), 6, 4, np.int32),
        ("biquad_i32_df1_clamp", H.biquad_clamp_i32(rc[:2]), 2, 4, np.int32),
        ("biquad_i32_dither", H.biquad_i32(ri[:1]), 1, 5, np.int32), ("biquad_i32_dither_clamp", H.biquad_clamp_i32(rc[:1]), 1, 5, np.int32),
        ("biquad_i32_wide", H.biquad_i32(ri[:1]), 1, 6, np.int32), ("biquad_i32_wide_clamp", H.biquad_clamp_i32(rc[:1]), 1, 6, np.int32),
        ("biquad_f32_df1", H.biquad_f32(rf[:1]), 1, 4, np.float32), ("biquad_f32_df1_clamp", H.biquad_clamp_f32(rfc[:1]), 1, 4, np.float32),
        ("biquad_f32_df2t", H.biquad_f32(rf[:5]), 5, 2, np.float32), ("biquad_f32_df2t_clamp", H.biquad_clamp_f32(rfc[:1]), 1, 2, np.float32),
        ("biquad_f64_df1", H.biquad_f64(rf[:1]), 1, 8, np.float64), ("biquad_f64_df2t", H.biquad_f64(rf[:2]), 2, 4, np.float64),
        ("biquad_f64_df1_clamp", H.biquad_clamp_f64(rfc[:1]), 1, 8, np.float64), ("biquad_f64_df2t_clamp", H.biquad_clamp_f64(rfc[:1]), 1, 4, np.float64),
    ]


def sample(rng, dt, n):
    if np.dtype(dt) == np.int32:
        return rng.integers(-(1 << 31), (1 << 31) - 1, size=n, dtype=np.int64).astype(np.int32)
    return rng.standard_normal(n).astype(dt)


def init_state(rng, dt, nwords, lanes):
    """random state planes [nwords, lanes] of 32-bit words; finite floats for the float types (f64 = lo, hi planes)"""
    if np.dtype(dt) == np.int32:
        return rng.integers(0, 1 << 32, size=(nwords, lanes), dtype=np.uint64).astype(np.uint32)
    if np.dtype(dt) == np.float32:
        return rng.standard_normal((nwords, lanes)).astype(np.float32).view(np.uint32)
    u = rng.standard_normal((nwords // 2, lanes)).view(np.uint64)
    st = np.empty((nwords, lanes), np.uint32)
    st[0::2] = (u & 0xFFFFFFFF).astype(np.uint32)
    st[1::2] = (u >> 32).astype(np.uint32)
    return st


def run_padded(eng, op, cfg, n, words, dt, rng, lanes, frames, layout, xpad, ypad, lane_off=0):
    """x / y live in padded buffers: rows of `row` valid elements at pitches row + xpad / row + ypad, starting `lane_off`
    elements into the buffer; returns (rc, y valid region, y buffer, state)."""
    o = H.oracle()
    row, rows = (frames, lanes) if layout == LM else (lanes, frames)
    xp, yp = row + xpad, row + ypad
    xh = sample(rng, dt, lanes * frames).reshape(rows, row)
    want = np.empty_like(xh)
    st0 = init_state(rng, dt, words * n, lanes)
    so = st0.copy()
    assert o.stream(op, cfg, n, so, xh, want, lanes, frames, layout) == 0
    t = tdtype(dt)
    xb = torch.full((rows * xp + lane_off,), SENT, dtype=t, device=DEV)
    yb = torch.full((rows * yp + lane_off,), SENT, dtype=t, device=DEV)
    xv = xb[lane_off:].view(rows, xp)
    xv[:, :row] = torch.from_numpy(xh).to(DEV)
    sg = torch.from_numpy(st0.view(np.int32)).to(DEV)
    rc = eng.fn[op + "_pitch"](C.cast(cfg, C.c_void_p), n, p(sg), C.c_void_p(xb.data_ptr() + lane_off * xb.element_size()), xp,
                               C.c_void_p(yb.data_ptr() + lane_off * yb.element_size()), yp, lanes, frames, layout, None)
    torch.cuda.synchronize()
    yv = yb[lane_off:].view(rows, yp)
    return rc, yv[:, :row].cpu().numpy(), yv[:, row:].cpu().numpy(), want, sg.cpu().numpy().view(np.uint32), so, xv


@pytest.mark.parametrize("layout", [FM, LM])
def test_padded_rows_match_the_dense_oracle(gpu, layout):
    rng = np.random.default_rng(31 + layout)
    for op, cfg, n, words, dt in cases(rng):
        for lanes, frames, xpad, ypad, off in ((70, 130, 5, 9, 0), (512, 203, 64, 128, 0), (256, 64, 4, 4, 3), (1, 33, 0, 7, 0), (300, 1, 12, 0, 1)):
            rc, got, pad, want, sg, so, _ = run_padded(gpu, op, cfg, n, words, dt, rng, lanes, frames, layout, xpad, ypad, off)
            assert rc == 0, (op, gpu.err())
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (op, lanes, frames, layout)
            assert np.array_equal(sg, so), (op, "state")
            assert (pad == SENT).all(), (op, "padding of y must stay untouched")


def test_dense_pitch_equals_plain_entry_and_errors(gpu):
    rng = np.random.default_rng(5)
    for op, cfg, n, words, dt in cases(rng):
        for layout in (FM, LM):
            lanes, frames = 512, 77
            row = frames if layout == LM else lanes
            x = torch.from_numpy(sample(rng, dt, lanes * frames)).to(DEV)
            ya, yb = torch.empty_like(x), torch.empty_like(x)
            nwords = words * n
            sa = torch.zeros((nwords, lanes), dtype=torch.int32, device=DEV)
            sb = torch.zeros_like(sa)
            assert gpu.stream(op, cfg, n, sa, x, ya, lanes, frames, layout) == 0
            assert gpu.fn[op + "_pitch"](C.cast(cfg, C.c_void_p), n, p(sb), p(x), row, p(yb), 0, lanes, frames, layout, None) == 0
            torch.cuda.synchronize()
            assert torch.equal(ya.view(torch.uint8), yb.view(torch.uint8)) and torch.equal(sa, sb), (op, layout)
            # shorter than a row / in place with different pitches
            assert gpu.fn[op + "_pitch"](C.cast(cfg, C.c_void_p), n, p(sb), p(x), row - 1, p(yb), 0, lanes, frames, layout, None) == _abi.IDSP_EINVAL
            assert gpu.fn[op + "_pitch"](C.cast(cfg, C.c_void_p), n, p(sb), p(x), row + 4, p(x), row, lanes, frames, layout, None) == _abi.IDSP_EINVAL
    # n == 0: the empty slice copies x to y (compose.rs:63-65) — with pitches, row by row
    lanes, frames = 10, 7
    xb = torch.arange(lanes * 12, dtype=torch.int32, device=DEV)
    yb = torch.full((lanes * 9,), SENT, dtype=torch.int32, device=DEV)
    assert gpu.fn["biquad_i32_df1_pitch"](None, 0, None, p(xb), 12, p(yb), 9, lanes, frames, LM, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(yb.view(lanes, 9)[:, :frames], xb.view(lanes, 12)[:, :frames]) and (yb.view(lanes, 9)[:, frames:] == SENT).all()


def test_frame_major_lane_block_in_place_on_the_lds_kernel(gpu):
    """A 65536-lane block at lane offset 8192 of a 98304-lane FRAME_MAJOR tensor, in place, wide enough for the LDS-DMA kernel
    (16-byte aligned rows), against the oracle; neighbours untouched.  (A 16384-lane block of a 32768-lane tensor the same
    way on the staged few-lanes kernel.)"""
    o = H.oracle()
    rng = np.random.default_rng(9)
    L, lanes, frames, off = 98304, 65536, 35, 8192
    xh = sample(rng, np.int32, L * frames).reshape(frames, L)
    cfg = H.biquad_i32([(rng.integers(-(1 << 29), 1 << 29, size=5).tolist(), 30)])
    sub = np.ascontiguousarray(xh[:, off:off + lanes])
    want = np.empty_like(sub)
    so = np.zeros((4, lanes), np.uint32)
    assert o.stream("biquad_i32_df1", cfg, 1, so, sub, want, lanes, frames, FM) == 0
    xd = torch.from_numpy(xh).to(DEV)
    sg = torch.zeros((4, lanes), dtype=torch.int32, device=DEV)
    ptr = C.c_void_p(xd.data_ptr() + off * 4)
    assert gpu.fn["biquad_i32_df1_pitch"](C.cast(cfg, C.c_void_p), 1, p(sg), ptr, L, ptr, L, lanes, frames, FM, None) == 0
    torch.cuda.synchronize()
    assert gpu.fn["last_kernel"]().decode().startswith("stream_frame_major_sweep[1 block/workgroup]<")
    got = xd.cpu().numpy()
    assert np.array_equal(got[:, off:off + lanes], want) and np.array_equal(sg.cpu().numpy().view(np.uint32), so)
    assert np.array_equal(got[:, :off], xh[:, :off]) and np.array_equal(got[:, off + lanes:], xh[:, off + lanes:])


def test_bylane_pitch(gpu):
    o = H.oracle()
    rng = np.random.default_rng(12)
    from tests._backends import OracleBackend

    ob = OracleBackend()
    for op, dtype, words, clamp in B.OPS[::3]:
        for layout in (FM, LM):
            lanes, frames, nsec = 300, 41, 2
            frac = 29 if dtype == np.int32 else None
            coef = B.coef_planes(rng, dtype, nsec, lanes, clamp, 29)
            xh = B.samples(rng, dtype, lanes * frames)
            so = np.zeros((words * nsec, lanes), np.uint32)
            rc, want = ob.bylane(op, coef, frac, nsec, so, xh, lanes, frames, layout)
            assert rc == 0
            row, rows = (frames, lanes) if layout == LM else (lanes, frames)
            xp, yp = row + 8, row + 20
            t = tdtype(dtype)
            xb = torch.full((rows, xp), 0, dtype=t, device=DEV)
            xb[:, :row] = torch.from_numpy(np.asarray(xh).reshape(rows, row)).to(DEV)
            yb = torch.full((rows, yp), SENT, dtype=t, device=DEV)
            cd = torch.from_numpy(np.ascontiguousarray(coef)).to(DEV)
            sg = torch.zeros((words * nsec, lanes), dtype=torch.int32, device=DEV)
            args = (p(cd),) + (() if frac is None else (frac,)) + (nsec, p(sg), p(xb), xp, p(yb), yp, lanes, frames, layout, None)
            assert gpu.fn[op + "_bylane_pitch"](*args) == 0, gpu.err()
            torch.cuda.synchronize()
            assert np.array_equal(yb[:, :row].cpu().numpy().view(np.uint8), np.asarray(want).reshape(rows, row).view(np.uint8)), (op, layout)
            assert np.array_equal(sg.cpu().numpy().view(np.uint32), so) and (yb[:, row:] == SENT).all()
