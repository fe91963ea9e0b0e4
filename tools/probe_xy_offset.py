"""How the C2 kernel's duration depends on where the output buffer sits relative to the input (y - x).
   python tools/probe_xy_offset.py  ->  one line per offset: median / min ms over 60 launches after 60 warm-up launches."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from idsp_amd import _abi  # noqa: E402
from idsp_amd._lib import call  # noqa: E402
from perf_configs import lowpass_sos  # noqa: E402

dev = torch.device("cuda:0")
lanes, frames = 65536, 4096
N = lanes * frames
cfg = _abi.BiquadI32()
call("biquad_i32_from_sos", (C.c_double * 6)(*lowpass_sos(0.01)), 30, C.byref(cfg))
OP = os.environ.get("IDSP_PROBE_OP", "biquad_i32_df1")  # or biquad_i32_df1_clamp / biquad_i32_wide_clamp (state words 4 / 6)
if "clamp" in OP:
    cfgs = (_abi.BiquadClampI32 * 1)()
    cfgs[0].ba[:] = list(cfg.ba)
    cfgs[0].frac, cfgs[0].u, cfgs[0].min, cfgs[0].max = 30, 3, -(1 << 30), 1 << 30
else:
    cfgs = (_abi.BiquadI32 * 1)(cfg)
stream = torch.cuda.Stream(device=dev)
sp = C.c_void_p(stream.cuda_stream)
state = torch.zeros((6, lanes), dtype=torch.int32, device=dev)
big = torch.empty(2 * N + (16 << 20), dtype=torch.int32, device=dev)
x = big[:N]
x.copy_(torch.randint(-(1 << 24), 1 << 24, (N,), dtype=torch.int32, device=dev))


def measure(y):
    def step():
        call(OP, C.cast(cfgs, C.c_void_p), 1, C.c_void_p(state.data_ptr()), C.c_void_p(x.data_ptr()),
             C.c_void_p(y.data_ptr()), lanes, frames, 0, sp)
    for _ in range(60):
        step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[30], ts[0]


deltas = [0, 256, 1024, 4096] + [k << 14 for k in range(1, 65)] + [1 << 21, 1 << 22, 1 << 23, 1 << 24]
if os.environ.get("IDSP_PROBE_QUICK"):
    deltas = [0, 1 << 14, 1 << 15, 3 << 14, 1 << 17, 1 << 18, 1 << 21, 1 << 22, 1 << 24]
for d in deltas:
    w = d // 4
    med, mn = measure(big[N + w:2 * N + w])
    print(f"y - x = 1 GiB + {d:>9d} B  median {med:.4f}  min {mn:.4f} ms", flush=True)
med, mn = measure(x)
print(f"in place                      median {med:.4f}  min {mn:.4f} ms")
