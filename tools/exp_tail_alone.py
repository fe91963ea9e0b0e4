import ctypes as C, json, os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
from idsp_amd import _abi
from idsp_amd._lib import call, load
import perf_configs as P
fn, _ = load()
q = _abi.BiquadI32(); call("biquad_i32_from_sos", (C.c_double * 6)(*P.lowpass_sos(0.01)), 30, C.byref(q)); cfg = (_abi.BiquadI32 * 1)(q)
frames = 4096
for lanes, pitch in ((4, 65540), (4, 4), (64, 65600), (256, 65792), (4096, 69632), (4096, 4096), (8192, 73728), (16384, 81920)):
    x = torch.randint(-(1 << 24), 1 << 24, (frames * pitch,), dtype=torch.int32, device="cuda"); y = torch.empty_like(x)
    st = torch.zeros((4, lanes), dtype=torch.int32, device="cuda")
    run = lambda: call("biquad_i32_df1_pitch", C.cast(cfg, C.c_void_p), 1, P.p(st), P.p(x), pitch, P.p(y), pitch, lanes, frames, 0, P.sptr())
    med, mn = P.timeit(run, 10)
    print(json.dumps({"lanes": lanes, "pitch": pitch, "ms": round(med, 4), "kernel": fn["last_kernel"]().decode()[:48]}), flush=True)
