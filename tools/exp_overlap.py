#!/usr/bin/env python3
"""Do a 65536-lane LDS-DMA launch and a small staged launch on a second stream overlap?  Wall time of both, launched
independently (no events between the streams), against each alone."""
import ctypes as C, json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from idsp_amd import _abi
from idsp_amd._lib import call, load
import perf_configs as P
fn, _ = load()
q = _abi.BiquadI32(); call("biquad_i32_from_sos", (C.c_double * 6)(*P.lowpass_sos(0.01)), 30, C.byref(q)); cfg = (_abi.BiquadI32 * 1)(q)
frames = 4096
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
xh = torch.randint(-(1 << 24), 1 << 24, (frames * 65536,), dtype=torch.int32, device="cuda"); yh = torch.empty_like(xh); sh = torch.zeros((4, 65536), dtype=torch.int32, device="cuda")
for tail in (4, 4096, 8192, 16384):
    xt = torch.randint(-(1 << 24), 1 << 24, (frames * tail,), dtype=torch.int32, device="cuda"); yt = torch.empty_like(xt); st = torch.zeros((4, tail), dtype=torch.int32, device="cuda")
    head = lambda: call("biquad_i32_df1", C.cast(cfg, C.c_void_p), 1, P.p(sh), P.p(xh), P.p(yh), 65536, frames, 0, C.c_void_p(s1.cuda_stream))
    tl = lambda s: call("biquad_i32_df1", C.cast(cfg, C.c_void_p), 1, P.p(st), P.p(xt), P.p(yt), tail, frames, 0, C.c_void_p(s.cuda_stream))
    def wall(f, n=50):
        for _ in range(20): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print(json.dumps({"tail": tail, "head_alone_ms": round(wall(head), 4), "tail_alone_ms": round(wall(lambda: tl(s2)), 4),
                      "both_two_streams_ms": round(wall(lambda: (head(), tl(s2))), 4), "both_one_stream_ms": round(wall(lambda: (head(), tl(s1))), 4),
                      "tail_first_two_streams_ms": round(wall(lambda: (tl(s2), head())), 4)}), flush=True)
