#!/usr/bin/env python3
"""Round 5: the FrameMajor biquad entries through the C ABI against the PLACEMENT of y — y - x swept over one row pitch in 64 steps,
x and y inside ONE allocation, plus y == x (in place) — for C5 (f32 DF2T, 2^20 lanes x 4096 frames, row pitch 4 MiB: 64 KiB steps),
its 8-GPU shard (131072 lanes) and C2 (i32 DF1, 65536 lanes: row pitch 256 KiB, 4 KiB steps).  One JSON line per shape:
fractions of the 8 TB/s HBM peak per offset, worst / best / mean, in place, and the kernel taken.
    python tools/exp_placement.py [c5] [c5s8] [c2] [steps per placement, default 5]"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idsp_amd import _abi
from idsp_amd._lib import call, load
import perf_configs as P

fn, _ = load()
FRAMES = 4096
SHAPES = {"c5": ("biquad_f32_df2t", 1 << 20, torch.float32, 2), "c5s8": ("biquad_f32_df2t", 1 << 17, torch.float32, 2), "c2": ("biquad_i32_df1", 65536, torch.int32, 4)}
which = [a for a in sys.argv[1:] if a in SHAPES] or ["c5", "c5s8", "c2"]
iters = next((int(a) for a in sys.argv[1:] if a.isdigit()), 5)
sos = (C.c_double * 6)(*P.lowpass_sos(0.01))
for name in which:
    op, lanes, dt, words = SHAPES[name]
    if dt == torch.float32:
        q = _abi.BiquadF32()
        call("biquad_f32_from_sos_f64", sos, C.byref(q))
        cfg = (_abi.BiquadF32 * 1)(q)
    else:
        q = _abi.BiquadI32()
        call("biquad_i32_from_sos", sos, 30, C.byref(q))
        cfg = (_abi.BiquadI32 * 1)(q)
    n, row = lanes * FRAMES, lanes * 4
    buf = torch.empty(2 * n + 2 * row // 4 + 1024, dtype=dt, device="cuda")
    x = buf[:n]
    if dt == torch.float32:
        x.uniform_(-1, 1)
    else:
        x.random_(-(1 << 24), 1 << 24)
    st = torch.zeros((words, lanes), dtype=torch.int32, device="cuda")
    base = buf.data_ptr() + n * 4
    fr = []
    for k in range(64):
        yp = C.c_void_p(base + k * (row // 64))
        run = lambda: call(op, C.cast(cfg, C.c_void_p), 1, P.p(st), P.p(x), yp, lanes, FRAMES, 0, P.sptr())
        med, _mn = P.timeit(run, iters, warm=2, warm_ms=20.0)
        fr.append(round(8 * n / (med * 1e-3) / 8e12, 3))
    kernel = fn["last_kernel"]().decode()[:60]
    run = lambda: call(op, C.cast(cfg, C.c_void_p), 1, P.p(st), P.p(x), P.p(x), lanes, FRAMES, 0, P.sptr())
    med, _mn = P.timeit(run, iters, warm=2, warm_ms=20.0)
    print(json.dumps({"shape": name, "op": op, "lanes": lanes, "frames": FRAMES, "step_bytes": row // 64, "frac_by_step": fr, "worst": min(fr), "best": max(fr),
                      "mean": round(sum(fr) / len(fr), 3), "inplace": round(8 * n / (med * 1e-3) / 8e12, 3), "kernel": kernel}), flush=True)
    del buf, x, st
    torch.cuda.empty_cache()
