"""`idsp_last_kernel()`: the diagnostic that lets bench.py (and a user) see which kernel a call dispatched to names the
kernel of every family, not only the stream kernels."""
import ctypes as C

import numpy as np
import pytest
import torch

from idsp_amd import _abi
from tests import _harness as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def p(t):
    return C.c_void_p(t.data_ptr())


def test_every_family_reports_its_kernel(gpu):
    name = lambda: gpu.fn["last_kernel"]().decode()
    lanes, frames = 512, 64
    x = torch.zeros(lanes * frames, dtype=torch.int32, device=DEV)
    y = torch.empty(lanes * frames * 2, dtype=torch.int32, device=DEV)
    st = torch.zeros((32, lanes), dtype=torch.int32, device=DEV)
    cfg = H.biquad_i32([([1 << 28, 0, 0, 0, 0], 30)])
    assert gpu.stream("biquad_i32_df1", cfg, 1, st, x, y, lanes, frames, H.LM) == 0 and name().startswith("stream_lane_major_staged[16 lanes/wave]<")
    assert "Df1I32<false>" in name()
    assert gpu.stream("biquad_i32_df1", cfg, 1, st, x, y, lanes, frames, H.FM) == 0 and name().startswith("stream_frame_major_staged[16 lanes/wave]<")
    lc = H.lockin_cfg([[1 << 20, -(1 << 27)]] * 2)
    assert gpu.cfgcall("lockin_i32_process", lc, st, x, y, lanes, frames, H.FM) == 0 and name().startswith("lockin_stages_kernel[8 waves")
    assert gpu.cfgcall("lockin_i32_process", lc, st, x, y, lanes, frames - 1, H.FM) == 0 and name().startswith("lockin_waves_kernel[6 waves")
    hc = _abi.HbfCascadeF32()
    assert gpu.fn["hbf_dec_cascade"](0, 4, C.byref(hc)) == 0
    xf = torch.zeros(lanes * 16 * 16, dtype=torch.float32, device=DEV)
    yf = torch.empty(lanes * 16, dtype=torch.float32, device=DEV)
    sth = torch.zeros((118, lanes), dtype=torch.int32, device=DEV)
    assert gpu.cfgcall("hbf_dec_f32", hc, sth, xf, yf, lanes, 16, H.LM) == 0 and name().startswith("hbf_dec_blk[LaneMajor]<")
    assert gpu.cfgcall("hbf_dec_f32", hc, sth, xf, yf, lanes, 16, H.FM) == 0 and name().startswith("hbf_dec_ring[FrameMajor]<")
    hc.taps[0][0] += 1e-3  # not a built-in tap set any more
    assert gpu.cfgcall("hbf_dec_f32", hc, sth, xf, yf, lanes, 16, H.LM) == 0 and name().startswith("hbf_dec_kernel (generic")
    cc = _abi.Cic(3, 1, 15)
    assert gpu.cfgcall("cic_dec_i32", cc, st, torch.zeros(lanes * 16 * 16, dtype=torch.int32, device=DEV), y, lanes, 16, H.FM) == 0
    assert name() == "cic_dec_kernel"
    assert gpu.fn["cossin_i32"](p(x), p(y), lanes, None) == 0 and name() == "cossin_kernel"
    assert gpu.fn["atan2_i32"](p(y), p(x), lanes, None) == 0 and name() == "atan2_kernel"
    torch.cuda.synchronize()
