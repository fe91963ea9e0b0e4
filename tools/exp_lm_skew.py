#!/usr/bin/env python3
"""LaneMajor C2: does it matter that every wave of a launch reads the same offset modulo the lane pitch at the same time?  One launch over
65536 lanes against TWO concurrent launches over 32768 lanes each (two streams, `_pitch` entry on lane blocks of one LaneMajor tensor), the
second started `skew` tiles of work later (a dummy launch of `skew` x 128 frames on the second stream in front of it), so that the two halves
of the chip sit at different offsets inside their rows."""
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
LM = 1
frames, lanes = 4096, 65536
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randint(-(1 << 24), 1 << 24, (frames * lanes,), dtype=torch.int32, device="cuda")
y = torch.empty_like(x)
xd = torch.randint(-(1 << 24), 1 << 24, (frames * 32768,), dtype=torch.int32, device="cuda")
yd = torch.empty_like(xd)
sta = torch.zeros((4, lanes), dtype=torch.int32, device="cuda")
st1 = torch.zeros((4, lanes // 2), dtype=torch.int32, device="cuda")
st2 = torch.zeros((4, lanes // 2), dtype=torch.int32, device="cuda")
std = torch.zeros((4, lanes // 2), dtype=torch.int32, device="cuda")
half = lanes // 2


def whole():
    call("biquad_i32_df1", C.cast(cfg, C.c_void_p), 1, P.p(sta), P.p(x), P.p(y), lanes, frames, LM, C.c_void_p(s1.cuda_stream))


ev_f, ev_j = torch.cuda.Event(), torch.cuda.Event()


def split(skew):
    ev_f.record(s1)
    s2.wait_event(ev_f)
    call("biquad_i32_df1_pitch", C.cast(cfg, C.c_void_p), 1, P.p(st1), P.p(x), frames, P.p(y), frames, half, frames, LM, C.c_void_p(s1.cuda_stream))
    if skew:  # keeps half of the chip busy elsewhere for `skew` tiles of time
        call("biquad_i32_df1_pitch", C.cast(cfg, C.c_void_p), 1, P.p(std), P.p(xd), frames, P.p(yd), frames, half, 128 * skew, LM, C.c_void_p(s2.cuda_stream))
    call("biquad_i32_df1_pitch", C.cast(cfg, C.c_void_p), 1, P.p(st2), C.c_void_p(x.data_ptr() + half * frames * 4), frames,
         C.c_void_p(y.data_ptr() + half * frames * 4), frames, half, frames, LM, C.c_void_p(s2.cuda_stream))
    ev_j.record(s2)
    s1.wait_event(ev_j)


def timeit(f):
    for _ in range(30):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(21):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s1); f(); b.record(s1); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


byts = 8 * lanes * frames
m, lo = timeit(whole)
print(json.dumps({"form": "one launch", "ms_median": round(m, 4), "ms_min": round(lo, 4), "frac": round(byts / m / 8e9, 3)}), flush=True)
for skew in (0, 1, 2, 4, 8):
    m, lo = timeit(lambda: split(skew))
    extra = 8 * half * 128 * skew
    print(json.dumps({"form": "two launches of 32768 lanes", "skew_tiles": skew, "ms_median": round(m, 4), "ms_min": round(lo, 4),
                      "frac_incl_dummy_bytes": round((byts + extra) / m / 8e9, 3), "frac_job_only": round(byts / m / 8e9, 3)}), flush=True)
