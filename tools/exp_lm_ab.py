"""A/B timing of i32 DF1 biquad variants IN ONE PROCESS on the same buffers (GPU): small libraries holding only the i32 DF1 unit
(build/exp_lm/lib_<name>.so: idsp_amd/csrc/biquad_i32_df1.hip compiled with an experiment macro + api_util.o) take turns.
usage: python tools/exp_lm_ab.py [--layout lm|fm] [--lanes 65536] [--frames 4096] [--rounds 5] [--iters 20] NAME [NAME ...]"""
import argparse
import ctypes as C
import json
import math
import os
import statistics
import sys

import torch

ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+")
ap.add_argument("--layout", default="lm")
ap.add_argument("--lanes", type=int, default=65536)
ap.add_argument("--frames", type=int, default=4096)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dir", default="build/exp_lm")
a = ap.parse_args()
dev = torch.device("cuda", 0)


class Rec(C.Structure):
    _fields_ = [("ba", C.c_int32 * 5), ("frac", C.c_int32)]


w0 = math.pi * 0.01
fcos, fsin = math.cos(w0), math.sin(w0)
alpha, b = 0.5 * fsin * math.sqrt(2.0), 0.5 * (1.0 - fcos)
sos = (C.c_double * 6)(b, 2 * b, b, 1 + alpha, -2 * fcos, 1 - alpha)
x = torch.randint(-(1 << 24), 1 << 24, (a.lanes * a.frames,), dtype=torch.int32, device=dev)
y = torch.empty_like(x)
st = torch.zeros((4, a.lanes), dtype=torch.int32, device=dev)
layout = 1 if a.layout == "lm" else 0
stream = torch.cuda.current_stream()
runs, ref = {}, None
for n in a.names:
    lib = C.CDLL(os.path.join(a.dir, f"lib_{n}.so"))
    rec = Rec()
    assert lib.idsp_biquad_i32_from_sos(sos, 30, C.byref(rec)) == 0
    f = lib.idsp_biquad_i32_df1
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
    runs[n] = (lambda f=f, rec=rec: f(C.byref(rec), 1, st.data_ptr(), x.data_ptr(), y.data_ptr(), a.lanes, a.frames, layout, None))
    st.zero_()
    assert runs[n]() == 0
    torch.cuda.synchronize()
    chk = int(y.to(torch.int64).sum().item())
    ref = chk if ref is None else ref
    assert chk == ref, (n, "output differs from the first variant")
for r in runs.values():
    for _ in range(20):
        r()
torch.cuda.synchronize()
res = {n: [] for n in runs}
for _ in range(a.rounds):
    for n, r in runs.items():
        for _ in range(3):
            r()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(a.iters):
            r()
        e1.record(stream)
        torch.cuda.synchronize()
        res[n].append(e0.elapsed_time(e1) / a.iters)
alg = 8 * a.lanes * a.frames
for n, v in res.items():
    med = statistics.median(v)
    print(json.dumps({"variant": n, "layout": a.layout, "lanes": a.lanes, "ms_median_of_rounds": round(med, 4), "ms_rounds": [round(t, 4) for t in v],
                      "frac_hbm_peak": round(alg / (med * 1e-3) / 8e12, 4)}))
