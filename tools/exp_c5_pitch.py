#!/usr/bin/env python3
"""C5 (f32 DF2T, 2^20 lanes x 4096 frames, FrameMajor) through `idsp_biquad_f32_df2t_pitch` at row pitches other than the dense 4 MiB:
do consecutive frames of a lane block meet in the same DRAM banks / channels at the power-of-two pitch?  Also 131072 / 262144 /
524288 lanes (the shards of the strong-scaling curve)."""
import ctypes as C, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idsp_amd import _abi
from idsp_amd._lib import call, load
import perf_configs as P

fn, _ = load()
q = _abi.BiquadF32()
call("biquad_f32_from_sos_f64", (C.c_double * 6)(*P.lowpass_sos(0.01)), C.byref(q))
cfg = (_abi.BiquadF32 * 1)(q)
frames = 4096
cases = [(1 << 20, d) for d in (0, 64, 512, 2048, 8192, 65536, 65536 + 2048)] + [(1 << 19, d) for d in (0, 2048, 8192)] + [(1 << 18, d) for d in (0, 2048)] + [(1 << 17, d) for d in (0, 2048)]
if len(sys.argv) > 1:
    cases = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]]
for lanes, extra in cases:
    pitch = lanes + extra
    x = torch.empty(frames * pitch, dtype=torch.float32, device="cuda").uniform_(-1, 1)
    y = torch.empty_like(x)
    st = torch.zeros((2, lanes), dtype=torch.float32, device="cuda")
    run = lambda: call("biquad_f32_df2t_pitch", C.cast(cfg, C.c_void_p), 1, P.p(st), P.p(x), pitch, P.p(y), pitch, lanes, frames, 0, P.sptr())
    med, mn = P.timeit(run, 8)
    print(json.dumps({"lanes": lanes, "pitch": pitch, "extra_bytes": extra * 4, "ms_median": round(med, 4), "ms_min": round(mn, 4),
                      "frac_hbm_peak": round(8 * lanes * frames / (med * 1e-3) / 8e12, 4), "kernel": fn["last_kernel"]().decode()[:50]}), flush=True)
    del x, y
    torch.cuda.empty_cache()
