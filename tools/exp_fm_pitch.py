#!/usr/bin/env python3
"""FrameMajor i32 DF1 at 65536 (and 65000) lanes x 4096 frames through `idsp_biquad_i32_df1_pitch` at row pitches with
different alignments: is it the RAGGED last block or the ROW ALIGNMENT that costs lane counts like 65000 their rate?"""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idsp_amd import _abi
from idsp_amd._lib import call, load
import perf_configs as P

fn, _ = load()
q = _abi.BiquadI32()
call("biquad_i32_from_sos", (C.c_double * 6)(*P.lowpass_sos(0.01)), 30, C.byref(q))
cfg = (_abi.BiquadI32 * 1)(q)
frames = 4096
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]]  # lanes:pitch pairs; default: the round-3 list
for lanes, pitch in shapes or ((65536, 65536), (65536, 65536 + 32), (65536, 65536 + 8), (65536, 65536 + 4), (65536, 65536 + 16), (65000, 65000), (65000, 65024), (65000, 65536), (65532, 65532), (65532, 65536)):
    x = torch.randint(-(1 << 24), 1 << 24, (frames * pitch,), dtype=torch.int32, device="cuda")
    y = torch.empty_like(x)
    st = torch.zeros((4, lanes), dtype=torch.int32, device="cuda")
    run = lambda: call("biquad_i32_df1_pitch", C.cast(cfg, C.c_void_p), 1, P.p(st), P.p(x), pitch, P.p(y), pitch, lanes, frames, 0, P.sptr())
    med, mn = P.timeit(run, 10)
    print(json.dumps({"lanes": lanes, "pitch": pitch, "pitch_bytes_mod_128": pitch * 4 % 128, "ms_median": round(med, 4), "frac_hbm_peak": round(8 * lanes * frames / (med * 1e-3) / 8e12, 4),
                      "kernel": fn["last_kernel"]().decode()[:60]}), flush=True)
    del x, y
