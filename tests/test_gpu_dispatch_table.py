"""The kernel the default dispatch takes at every BASELINE shape and on both sides of every lane-count cliff of
idsp_amd/csrc/dispatch_thresholds.h, pinned through `idsp_last_kernel()` ("upkeep": one table of thresholds and a test
that pins the kernel chosen at each BASELINE shape and at each cliff).  Short calls (32 frames, C3: 64 input frames per output tile) —
the kernel choice does not depend on the frame count above 16 — so the whole table runs in seconds; results are only checked to be
written (parity is everywhere else).  Reference loop nests replaced: dsp-process/src/compose.rs:468-494 (`Lanes`), process.rs:122-141."""
import ctypes as C

import numpy as np
import pytest
import torch

from idsp_amd import _abi
from tests import _harness as H

pytestmark = pytest.mark.gpu
DEV = "cuda"
FM, LM = H.FM, H.LM


def last(eng):
    return eng.fn["last_kernel"]().decode()


def biquad_kernel(gpu, op, lanes, frames, layout, dtype, words, pitch=None, n=1):
    o = H.oracle()
    sos = (C.c_double * 6)(*o.lowpass_sos(0.01))
    if dtype == torch.int32:
        q = _abi.BiquadI32()
        assert o.fn["biquad_i32_from_sos"](sos, 30, C.byref(q)) == 0
        cfg = (_abi.BiquadI32 * n)(*([q] * n))
    else:
        q = _abi.BiquadF32()
        assert o.fn["biquad_f32_from_sos_f64"](sos, C.byref(q)) == 0
        cfg = (_abi.BiquadF32 * n)(*([q] * n))
    pitch = pitch or (lanes if layout == FM else frames)
    rows = frames if layout == FM else lanes
    x = torch.zeros(rows * pitch, dtype=dtype, device=DEV)
    y = torch.full_like(x, 7)
    st = torch.zeros((words * n, lanes), dtype=torch.int32, device=DEV)
    rc = gpu.fn[op + "_pitch"](C.cast(cfg, C.c_void_p), n, C.c_void_p(st.data_ptr()), C.c_void_p(x.data_ptr()), pitch, C.c_void_p(y.data_ptr()), pitch,
                               lanes, frames, layout, None)
    torch.cuda.synchronize()
    assert rc == 0, gpu.err()
    assert int(y.view(rows, pitch)[:, : (lanes if layout == FM else frames)].abs().max()) == 0, "zero input, zero state: zero output must have been written"
    return last(gpu)


SWEEP = "stream_frame_major_sweep["
ODD = " + stream_frame_major_few (lanes % 4, second stream)"
# (lanes, expected start of the kernel name, expected end or None): i32 DF1, FrameMajor, dense rows
I32_FM = [
    (65536, SWEEP + "1 block/workgroup]<", None),                       # C2
    (131072, SWEEP + "2 blocks/workgroup]<", None),                     # C5 shard at 8 GPUs
    (262144, SWEEP + "4 blocks/workgroup]<", None),
    (524288, SWEEP + "8 blocks/workgroup]<", None),
    (1048576, SWEEP + "16 blocks/workgroup]<", None),                   # C5
    (100000, SWEEP + "2 blocks/workgroup]<", None),                     # narrow blocks (208 lanes)
    (65552, "stream_frame_major_sweep + stream_frame_major_staged (remainder, second stream)<", None),  # a little above one round
    (65537, "stream_frame_major_lds[XCD-contiguous blocks]<", ODD),     # cliff: rows off the grid, last lane beside
    (131073, SWEEP + "2 blocks/workgroup, XCD-contiguous]<", ODD),      # cliff: off the grid beyond 98304 lanes
    (49152, SWEEP + "1 block/workgroup]<", None),
    (32768, SWEEP + "1 block/workgroup]<", None),                       # several frames per segment
    (32769, SWEEP + "1 block/workgroup, XCD-contiguous]<", ODD),        # cliff: rows off the grid, several frames per segment up to 53248 lanes
    (24576, SWEEP + "1 block/workgroup]<", None),
    (24560, "stream_frame_major_staged[32 lanes/wave]<", None),         # below kSweepMinLanesFps
    (16384, "stream_frame_major_staged[32 lanes/wave]<", None),
    (8192, "stream_frame_major_staged[32 lanes/wave]<", None),
    (8176, "stream_frame_major_staged[16 lanes/wave]<", None),
]


@pytest.mark.parametrize("lanes,start,end", I32_FM, ids=[str(s[0]) for s in I32_FM])
def test_i32_df1_frame_major_lane_counts(gpu, lanes, start, end):
    k = biquad_kernel(gpu, "biquad_i32_df1", lanes, 32, FM, torch.int32, 4)
    assert k.startswith(start) and (end is None or k.endswith(end)) and (end is not None or not k.endswith(ODD)), (lanes, k)


def test_c5_and_lane_major_biquads(gpu):
    assert biquad_kernel(gpu, "biquad_f32_df2t", 1 << 20, 32, FM, torch.float32, 2).startswith(SWEEP + "16 blocks/workgroup]<")
    assert biquad_kernel(gpu, "biquad_f32_df2t", 1 << 17, 32, FM, torch.float32, 2).startswith(SWEEP + "2 blocks/workgroup]<")
    assert biquad_kernel(gpu, "biquad_i32_df1", 65536, 4096, LM, torch.int32, 4).startswith("stream_lane_major_staged<")
    assert biquad_kernel(gpu, "biquad_i32_df1", 32768, 4096, LM, torch.int32, 4).startswith("stream_lane_major_staged[32 lanes/wave]<")
    assert biquad_kernel(gpu, "biquad_i32_df1", 16384, 4096, LM, torch.int32, 4).startswith("stream_lane_major_staged[16 lanes/wave]<")


def test_several_sweeps_per_launch_only_when_eight_blocks_wide(gpu):
    """fm_sweep.h `sweep_takes`: a family whose largest blocks-per-workgroup count does not cover the lanes in ONE sweep keeps round 4's dispatch,
    unless its sweeps are 8 blocks per workgroup wide (two-section chains: 8, three: 2)."""
    assert biquad_kernel(gpu, "biquad_i32_df1", 1 << 19, 32, FM, torch.int32, 4, n=2).startswith(SWEEP + "8 blocks/workgroup]<")
    assert biquad_kernel(gpu, "biquad_i32_df1", 1 << 20, 32, FM, torch.int32, 4, n=2).startswith(SWEEP + "8 blocks/workgroup]<")  # two sweeps
    assert biquad_kernel(gpu, "biquad_i32_df1", 1 << 17, 32, FM, torch.int32, 4, n=3).startswith(SWEEP + "2 blocks/workgroup]<")
    assert not biquad_kernel(gpu, "biquad_i32_df1", 1 << 18, 32, FM, torch.int32, 4, n=3).startswith(SWEEP)                      # would be two narrow sweeps


def test_c3_and_c4(gpu):
    # C3: HbfDec /16 (hbf.rs:385-421), 16384 lanes, both layouts
    cfgs = _abi.HbfCascadeF32()
    assert gpu.fn["hbf_dec_cascade"](0, 4, C.byref(cfgs)) == 0
    words = gpu.fn["hbf_dec_state_words"](C.byref(cfgs))
    lanes, frames = 16384, 128
    for layout, want in ((FM, "hbf_dec_ring[FrameMajor]"), (LM, "hbf_dec_blk[LaneMajor]")):
        x = torch.zeros(lanes * frames * 16, dtype=torch.float32, device=DEV)
        y = torch.full((lanes * frames,), 3.0, dtype=torch.float32, device=DEV)
        st = torch.zeros((words, lanes), dtype=torch.int32, device=DEV)
        assert gpu.fn["hbf_dec_f32"](C.byref(cfgs), C.c_void_p(st.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), lanes, frames, layout, None) == 0
        torch.cuda.synchronize()
        assert last(gpu).startswith(want) and float(y.abs().max()) == 0.0, last(gpu)
    # C4: lock-in [Lowpass<2>; 2] (lockin.rs:30-39), 32768 lanes, both layouts
    cfg = _abi.LockinI32()
    cfg.order, cfg.cascade = 2, 2
    for c in range(2):
        cfg.k[c][0], cfg.k[c][1] = 1 << 20, -(1 << 26)
    words = gpu.fn["lockin_state_words"](C.byref(cfg))
    lanes, frames = 32768, 256
    for layout in (FM, LM):
        x = torch.zeros(lanes * frames, dtype=torch.int32, device=DEV)
        y = torch.full((lanes * frames * 2,), 5, dtype=torch.int32, device=DEV)
        st = torch.zeros((words, lanes), dtype=torch.int32, device=DEV)
        assert gpu.fn["lockin_i32_process"](C.byref(cfg), C.c_void_p(st.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), lanes, frames, layout, None) == 0
        torch.cuda.synchronize()
        assert last(gpu).startswith("lockin_waves_kernel[4 waves per 64 lanes]") and int(y.abs().max()) == 0, last(gpu)
