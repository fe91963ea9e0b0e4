"""C3, C4 and C5 at FULL size with EVERY element compared against the CPU oracle (C2 has had this since round 1 in
tests/test_gpu_fullsize.py; there the other three compare a lane subset and rely on chunked == whole / layout
equivalence over the rest).  Inputs are SURVEY.md §8d's streams — `numpy.random.default_rng(seed)`, seeds C3 = 3,
C4 = 4, C5 = 5.  The oracle runs over all lanes in lane blocks on a thread pool (ctypes releases the GIL; lanes never
interact, dsp-process/src/compose.rs:468-494), each block with its own slice of the state planes.

  C3  hbf::HbfDec /16 (HBF_TAPS stages 3,2,1,0; src/hbf.rs:412-421), f32, 16384 lanes x 65536 input samples, LANE_MAJOR:
      all 16384 x 4096 outputs and the 118 state words per lane, 0 ULP.
  C4  Lockin<[Lowpass<2>; 2]> (src/lockin.rs:30-39), i32, 32768 lanes x 4096, FRAME_MAJOR, per-lane random Accu step:
      all 32768 x 4096 Complex<i32> outputs and the 18 state words per lane, bit-exact.
  C5  Biquad<f32> DF2T (src/iir/biquad.rs:418-428) over 2^20 lanes x 4096 in ONE launch (the regime of the persistent
      column-panel grid): the oracle filters a 65536-lane slab, the device tensor is that slab tiled 16 times along the
      lanes, and every one of the 2^20 x 4096 outputs (and both state planes) must equal the slab's result, 0 ULP."""
import ctypes as C
import math
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

from idsp_amd import _abi
from tests import _harness as H

pytestmark = pytest.mark.gpu
FM, LM = H.FM, H.LM
DEV = "cuda:0"


def _threads():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else int(float(q) / float(per) + 0.5)
    except (OSError, ValueError):
        quota = None
    n = len(os.sched_getaffinity(0))
    return max(1, min(n, quota) if quota else n)


def oracle_all_lanes(call, x_lm, y_lm, state, block=128):
    """call(state_block, x_block, y_block, lanes_of_block) over LANE_MAJOR rows, lane blocks in parallel; `state` is the
    [words, lanes] plane array (updated in place)."""
    lanes = x_lm.shape[0]

    def work(lo):
        hi = min(lanes, lo + block)
        st = np.ascontiguousarray(state[:, lo:hi])
        rc = call(st, x_lm[lo:hi], y_lm[lo:hi], hi - lo)
        assert rc == 0
        state[:, lo:hi] = st

    with ThreadPoolExecutor(max_workers=_threads()) as pool:
        list(pool.map(work, range(0, lanes, block)))


def test_c3_every_output_against_the_oracle(gpu):
    lanes, frames, R, words = 16384, 4096, 16, 118
    o = H.oracle()
    cfg = _abi.HbfCascadeF32()
    assert o.fn["hbf_dec_cascade"](0, 4, C.byref(cfg)) == 0
    x = np.random.default_rng(3).standard_normal((lanes, frames * R), dtype=np.float32)
    xd = torch.from_numpy(x).to(DEV)
    yd = torch.empty((lanes, frames), dtype=torch.float32, device=DEV)
    sd = torch.zeros((words, lanes), dtype=torch.int32, device=DEV)
    assert gpu.cfgcall("hbf_dec_f32", cfg, sd, xd, yd, lanes, frames, LM) == 0
    torch.cuda.synchronize()
    want = np.empty((lanes, frames), np.float32)
    st = np.zeros((words, lanes), np.uint32)
    oracle_all_lanes(lambda s, a, b, n: o.cfgcall("hbf_dec_f32", cfg, s, a, b, n, frames, LM), x, want, st)
    got = yd.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int(H.ulp_diff_f32(got, want).max())
    assert np.array_equal(sd.cpu().numpy().view(np.uint32), st)
    assert gpu.fn["last_kernel"]().decode().startswith("hbf_dec_blk[LaneMajor]")
    # the same tensor FRAME_MAJOR ([frame][lane][R] chunks): every output again
    xf = xd.view(lanes, frames, R).permute(1, 0, 2).contiguous()
    del xd
    yf = torch.empty((frames, lanes), dtype=torch.float32, device=DEV)
    sd.zero_()
    assert gpu.cfgcall("hbf_dec_f32", cfg, sd, xf, yf, lanes, frames, FM) == 0
    torch.cuda.synchronize()
    assert np.array_equal(yf.t().contiguous().cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(sd.cpu().numpy().view(np.uint32), st)


def test_c4_every_output_against_the_oracle(gpu):
    lanes, frames, words = 32768, 4096, 18
    o = H.oracle()
    k = math.pi * (1 << 31) * 1e-3  # src/lowpass.rs:29-46: f0 = 1e-3 fn
    lp = [int(k * k / (1 << 32)), -int(k * math.sqrt(2.0))]
    cfg = H.lockin_cfg([lp, lp])
    rng = np.random.default_rng(4)
    x = rng.integers(-(1 << 28), 1 << 28, size=(frames, lanes), dtype=np.int32)  # FRAME_MAJOR
    st0 = np.zeros((words, lanes), np.uint32)
    st0[1] = rng.integers(0, 1 << 32, size=lanes, dtype=np.uint64).astype(np.uint32)  # Accu.step (src/accu.rs:16-41), state = 0
    xd = torch.from_numpy(x).to(DEV)
    sd = torch.from_numpy(st0.view(np.int32).copy()).to(DEV)
    yd = torch.empty((frames, lanes, 2), dtype=torch.int32, device=DEV)
    assert gpu.cfgcall("lockin_i32_process", cfg, sd, xd, yd, lanes, frames, FM) == 0
    torch.cuda.synchronize()
    assert gpu.fn["last_kernel"]().decode().startswith("lockin_waves_kernel")
    x_lm = np.ascontiguousarray(x.T)
    want = np.empty((lanes, frames, 2), np.int32)
    st = st0.copy()
    oracle_all_lanes(lambda s, a, b, n: o.cfgcall("lockin_i32_process", cfg, s, a, b, n, frames, LM), x_lm, want, st)
    got = yd.cpu().numpy()
    assert np.array_equal(got, want.transpose(1, 0, 2))
    assert np.array_equal(sd.cpu().numpy().view(np.uint32), st)
    # LANE_MAJOR launch of the same tensor: every output again
    xl = torch.from_numpy(x_lm).to(DEV)
    sd = torch.from_numpy(st0.view(np.int32).copy()).to(DEV)
    yl = torch.empty((lanes, frames, 2), dtype=torch.int32, device=DEV)
    assert gpu.cfgcall("lockin_i32_process", cfg, sd, xl, yl, lanes, frames, LM) == 0
    torch.cuda.synchronize()
    assert np.array_equal(yl.cpu().numpy(), want) and np.array_equal(sd.cpu().numpy().view(np.uint32), st)


def test_c5_every_output_of_the_one_launch_against_an_oracle_slab(gpu):
    lanes, frames, slab = 1 << 20, 4096, 65536
    o = H.oracle()
    q = _abi.BiquadF32()
    assert o.fn["biquad_f32_from_sos_f64"]((C.c_double * 6)(*o.lowpass_sos(0.01)), C.byref(q)) == 0
    cfg = (_abi.BiquadF32 * 1)(q)
    xs = np.random.default_rng(5).standard_normal((frames, slab), dtype=np.float32)  # SURVEY §8d C5 stream, a 65536-lane slab
    x_lm = np.ascontiguousarray(xs.T)
    want = np.empty((slab, frames), np.float32)
    st = np.zeros((2, slab), np.uint32)
    oracle_all_lanes(lambda s, a, b, n: o.stream("biquad_f32_df2t", cfg, 1, s, a, b, n, frames, LM), x_lm, want, st, block=512)
    del x_lm
    reps = lanes // slab
    xd = torch.from_numpy(xs).to(DEV).repeat(1, reps)  # [frames][2^20]: lane l carries slab lane l % 65536
    yd = torch.empty_like(xd)
    sd = torch.zeros((2, lanes), dtype=torch.int32, device=DEV)
    assert gpu.stream("biquad_f32_df2t", cfg, 1, sd, xd, yd, lanes, frames, FM) == 0
    torch.cuda.synchronize()
    assert gpu.fn["last_kernel"]().decode().startswith("stream_frame_major_sweep[16 blocks/workgroup]<")  # one dense sweep (fm_sweep.h)
    del xd
    wd = torch.from_numpy(np.ascontiguousarray(want.T)).to(DEV).view(torch.int32)  # [frames][slab]
    yv = yd.view(torch.int32).view(frames, reps, slab)
    for r in range(reps):
        assert torch.equal(yv[:, r], wd), f"lanes {r * slab}..{(r + 1) * slab}"
    sw = torch.from_numpy(st.view(np.int32).copy()).to(DEV)
    assert torch.equal(sd.view(2, reps, slab), sw[:, None, :].expand(2, reps, slab))
