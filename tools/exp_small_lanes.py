"""Biquad (i32 DF1, one section, and the 8-section cascade) at small lane counts, both layouts: where the launch no longer
fills the chip.  One JSON line per shape (tools/perf_configs.py conventions)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import perf_configs as P
for lanes, frames in ((1024, 65536), (4096, 16384), (8192, 8192), (16384, 4096), (32768, 4096), (49152, 4096)):
    for layout in (0, 1):
        P.biquad("biquad_i32_df1", torch.int32, 4, lanes, frames, layout, 1, 10, "small")
        P.biquad("biquad_f32_df2t", torch.float32, 2, lanes, frames, layout, 1, 10, "small")
