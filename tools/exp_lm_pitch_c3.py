"""Are the LaneMajor half-band / Cic kernels sensitive to rows that are not a power of two long?  (The lock-in was: every lane at
the same offset of its row at the same time, profiles/NOTES.md "lanes in phase".)  C3 shape with 4096 + d output frames per lane."""
import sys, os, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_configs as P
for d in (0, 8, 16, 32, 64, 136):
    P.hbf("dec", 4, 16384, 4096 + d, 1, 10, "pitch")
for d in (0, 8, 32, 136):
    P.hbf("int", 4, 16384, 4096 + d, 1, 10, "pitch")
    P.cic("dec", torch.int32, 3, 15, 16384, 4096 + d, 1, 10, "pitch")
