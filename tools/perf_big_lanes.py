#!/usr/bin/env python3
"""Every FrameMajor processor family at 2^18 .. 2^20 lanes x 4096 frames (development tool; run twice, with and without
IDSP_DIAG=1 IDSP_NO_SWEEP=1, to compare the dense-sweep kernel with the round-4 dispatch at lane counts beyond one sweep of a family's
largest blocks-per-workgroup count).  One JSON line per kernel, like tools/perf_configs.py."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import perf_configs as PC  # noqa: E402

FM = 0
it = 5
lgs = [int(a) for a in sys.argv[1:]] or [18, 19, 20]
for lg in lgs:
    n = 1 << lg
    PC.biquad("biquad_i32_df1", torch.int32, 4, n, 4096, FM, 1, it, "big")
    PC.biquad("biquad_f32_df2t", torch.float32, 2, n, 4096, FM, 1, it, "big")
    PC.biquad("biquad_f32_df1", torch.float32, 4, n, 4096, FM, 1, it, "big")
    PC.biquad("biquad_f32_df2t_clamp", torch.float32, 2, n, 4096, FM, 1, it, "big")
    PC.biquad_bylane("biquad_f32_df2t", torch.float32, 2, 5, n, 4096, FM, 1, it, "big")
    PC.biquad_bylane("biquad_i32_df1", torch.int32, 4, 5, n, 4096, FM, 1, it, "big")
    PC.biquad_bylane("biquad_i32_df1_clamp", torch.int32, 4, 8, n, 4096, FM, 1, it, "big")
    PC.biquad("biquad_i32_df1_clamp", torch.int32, 4, n, 4096, FM, 1, it, "big")
    PC.biquad("biquad_i32_wide", torch.int32, 6, n, 4096, FM, 1, it, "big")
    PC.biquad("biquad_i32_df1", torch.int32, 4, n, 4096, FM, 2, it, "big")
    PC.biquad("biquad_f32_df2t", torch.float32, 2, n, 4096, FM, 2, it, "big")
    PC.biquad("cascade_i32_df1", torch.int32, 4, n, 4096, FM, 2, it, "big")
    PC.biquad("normal_f32_df1", torch.float32, 4, n, 4096, FM, 1, it, "big")
    PC.biquad("biquad_i32_df1", torch.int32, 4, n, 4096, FM, 4, it, "big")
