"""i32 DF1 FrameMajor x 4096 frames at lane counts whose rows start off the 64-byte / 16-byte grid (staged kernel below 49152 lanes,
LDS-DMA kernel with the XCD-contiguous block order above)."""
import sys, os, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_configs as P
for lanes in (32768, 32769, 32772, 32784, 16384, 16385, 16388, 40001, 49156, 65536, 65540, 65000, 65537, 100000, 131072, 131076):
    P.biquad("biquad_i32_df1", torch.int32, 4, lanes, 4096, 0, 1, 10, os.environ.get("TAG", "ragged"))
