"""i32 DF1 FrameMajor x 4096 frames at lane counts (LANES=a,b,...) around 16384 whose rows start off the 64-byte grid: which
lanes-per-wave form of the staged kernel (IDSP_DIAG=1 IDSP_FM_LANES_PER_WAVE=16/32/64) wins with plain accesses (round 4)."""
import sys, os, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_configs as P
for lanes in [int(a) for a in os.environ.get("LANES", "16384,16385,16388,20484,24580,12292,8196").split(",")]:
    P.biquad("biquad_i32_df1", torch.int32, 4, lanes, 4096, 0, 1, 10, os.environ.get("TAG", "r"))
