#!/usr/bin/env python3
"""FrameMajor i32 DF1 x 4096 frames on rows off the 16-byte grid at SMALL lane counts (where the staged single-wave kernel and
the few-lanes remainder kernel decide): lanes x pitch through `idsp_biquad_i32_df1_pitch`, run twice — default dispatch and
`IDSP_DIAG=1 IDSP_ALIGN16_ONLY=1` (round 2's rule: the register-window kernel)."""
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
shapes = [(1, 1), (3, 3), (8192, 8192), (8192, 8193), (8195, 8195), (16384, 16384), (16384, 16385), (16385, 16385), (32768, 32768), (32768, 32769), (32769, 32769),
          (49152, 49153), (49153, 49153), (16384, 16388), (32768, 32772), (8192, 8196)]
for lanes, pitch in shapes:
    x = torch.randint(-(1 << 24), 1 << 24, (frames * pitch,), dtype=torch.int32, device="cuda")
    y = torch.empty_like(x)
    st = torch.zeros((4, lanes), dtype=torch.int32, device="cuda")
    run = lambda: call("biquad_i32_df1_pitch", C.cast(cfg, C.c_void_p), 1, P.p(st), P.p(x), pitch, P.p(y), pitch, lanes, frames, 0, P.sptr())
    med, mn = P.timeit(run, 10)
    print(json.dumps({"lanes": lanes, "pitch": pitch, "ms_median": round(med, 4), "frac_hbm_peak": round(8 * lanes * frames / (med * 1e-3) / 8e12, 4),
                      "align16_only": os.environ.get("IDSP_ALIGN16_ONLY") == "1", "staged_no_xcdc": os.environ.get("IDSP_STAGED_NO_XCDC") == "1", "kernel": fn["last_kernel"]().decode()[:48] + " ... " + fn["last_kernel"]().decode()[-48:]}), flush=True)
    del x, y
