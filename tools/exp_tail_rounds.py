"""Lane counts just above whole rounds of 256-lane blocks: the call as dispatched (whole rounds + remainder on a second stream) against IDSP_DIAG=1 (no split).
What a 4-lane remainder costs is not the remainder (0.12 ms alone, tools/exp_tail_alone.py) but the rows of the whole tensor leaving the 64-byte grid."""
import ctypes as C, json, os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
from idsp_amd import _abi
from idsp_amd._lib import call, load
import perf_configs as P
fn, _ = load()
q = _abi.BiquadI32(); call("biquad_i32_from_sos", (C.c_double * 6)(*P.lowpass_sos(0.01)), 30, C.byref(q)); cfg = (_abi.BiquadI32 * 1)(q)
frames = 4096
for lanes in (65536, 65540, 131072, 131076, 131136, 131328, 135168, 196608, 196612):
    x = torch.randint(-(1 << 24), 1 << 24, (frames * lanes,), dtype=torch.int32, device="cuda"); y = torch.empty_like(x)
    st = torch.zeros((4, lanes), dtype=torch.int32, device="cuda")
    run = lambda: call("biquad_i32_df1", C.cast(cfg, C.c_void_p), 1, P.p(st), P.p(x), P.p(y), lanes, frames, 0, P.sptr())
    med, mn = P.timeit(run, 10)
    print(json.dumps({"diag": os.environ.get("IDSP_DIAG", "0"), "lanes": lanes, "ms": round(med, 4), "frac": round(8 * lanes * frames / med / 8e9, 3), "kernel": fn["last_kernel"]().decode()[:70]}), flush=True)
    del x, y
