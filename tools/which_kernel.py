#!/usr/bin/env python3
"""Which kernel(s) does a FrameMajor i32 DF1 call of a given lane count dispatch to, and how long does it take?
usage: python tools/which_kernel.py LANES[:PITCH] [...]   (GPU; PITCH = lanes between frames, idsp_biquad_i32_df1_pitch)"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from idsp_amd import _abi  # noqa: E402
from idsp_amd._lib import call, load  # noqa: E402
from tools.perf_configs import lowpass_sos, p, sptr, timeit  # noqa: E402

dev = torch.device("cuda:0")
for arg in sys.argv[1:]:
    lanes, _, pitch = arg.partition(":")
    lanes = int(lanes)
    pitch = int(pitch) if pitch else lanes
    frames = 4096
    q = _abi.BiquadI32()
    call("biquad_i32_from_sos", (C.c_double * 6)(*lowpass_sos(0.01)), 30, C.byref(q))
    x = torch.randint(-(1 << 24), 1 << 24, (pitch * frames,), dtype=torch.int32, device=dev)
    y = torch.empty_like(x)
    st = torch.zeros((4, lanes), dtype=torch.int32, device=dev)
    if pitch == lanes:
        run = lambda: call("biquad_i32_df1", C.byref(q), 1, p(st), p(x), p(y), lanes, frames, 0, sptr())  # noqa: E731
    else:
        run = lambda: call("biquad_i32_df1_pitch", C.byref(q), 1, p(st), p(x), pitch, p(y), pitch, lanes, frames, 0, sptr())  # noqa: E731
    med, mn = timeit(run, 10)
    print(arg, round(med, 4), load()[0]["last_kernel"]().decode()[:150], flush=True)
