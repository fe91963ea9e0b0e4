"""LaneMajor i32 DF1 x `--frames` frames over a list of lane counts through the C ABI (GPU): where the lane count stops fitting one generation
of workgroups (round 6: 65537 lanes ran 0.50 ms where 65536 run 0.38).  One process, one allocation (sized for the largest count).
usage: python tools/exp_lm_lanes.py [--layout lm|fm] [--frames 4096] [--iters 20] LANES [LANES ...]"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idsp_amd import _abi  # noqa: E402
from idsp_amd._lib import load  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("lanes", nargs="+", type=int)
ap.add_argument("--layout", default="lm")
ap.add_argument("--frames", type=int, default=4096)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--op", default="biquad_i32_df1")
ap.add_argument("--inplace", action="store_true", help="y == x")
a = ap.parse_args()
fn, _ = load()
dev = torch.device("cuda", 0)
big = max(a.lanes)
f32 = "f32" in a.op
x = (torch.randn(big * a.frames, device=dev) if f32 else torch.randint(-(1 << 24), 1 << 24, (big * a.frames,), dtype=torch.int32, device=dev))
y = x if a.inplace else torch.empty_like(x)
st = torch.zeros(8 * big, dtype=torch.int32, device=dev)
layout = 1 if a.layout == "lm" else 0
sos = (C.c_double * 6)(2.4e-4, 4.8e-4, 2.4e-4, 1.0, -1.955, 0.956)
if f32:
    q = _abi.BiquadF32()
    assert fn["biquad_f32_from_sos_f64"](sos, C.byref(q)) == 0
else:
    q = _abi.BiquadI32()
    assert fn["biquad_i32_from_sos"](sos, 30, C.byref(q)) == 0
stream = torch.cuda.current_stream()
for lanes in a.lanes:
    def run():
        rc = fn[a.op](C.byref(q), 1, C.c_void_p(st.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), lanes, a.frames, layout, None)
        assert rc == 0
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    name = fn["last_kernel"]().decode()
    ts = []
    for _ in range(a.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(a.iters):
            run()
        e1.record(stream)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / a.iters)
    med = statistics.median(ts)
    print(json.dumps({"op": a.op, "inplace": a.inplace, "layout": a.layout, "lanes": lanes, "frames": a.frames, "ms": round(med, 4), "frac_hbm_peak": round(8 * lanes * a.frames / (med * 1e-3) / 8e12, 4),
                      "kernel": name.split("<")[0]}), flush=True)
