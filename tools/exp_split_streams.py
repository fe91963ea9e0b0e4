#!/usr/bin/env python3
"""Lane counts between whole rounds of 256 lane blocks (73728 lanes = 288 blocks: 0.53 of the HBM peak, 32 CUs get two
workgroups): run the whole rounds on the LDS-DMA kernel and the remainder CONCURRENTLY on a second stream (it takes the
staged single-wave kernel), both through the `_pitch` entry on lane blocks of one FrameMajor tensor."""
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
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for lanes in (65536, 69632, 73728, 81920, 90112, 100000, 114688, 147456, 163840):
    x = torch.randint(-(1 << 24), 1 << 24, (frames * lanes,), dtype=torch.int32, device="cuda")
    y = torch.empty_like(x)
    head = lanes // 65536 * 65536
    tail = lanes - head
    sth = torch.zeros((4, head), dtype=torch.int32, device="cuda")
    stt = torch.zeros((4, max(tail, 1)), dtype=torch.int32, device="cuda")
    sta = torch.zeros((4, lanes), dtype=torch.int32, device="cuda")
    def whole():
        call("biquad_i32_df1", C.cast(cfg, C.c_void_p), 1, P.p(sta), P.p(x), P.p(y), lanes, frames, 0, C.c_void_p(s1.cuda_stream))
    ev_f, ev_j = torch.cuda.Event(), torch.cuda.Event()
    def split():
        ev_f.record(s1)
        s2.wait_event(ev_f)
        call("biquad_i32_df1_pitch", C.cast(cfg, C.c_void_p), 1, P.p(sth), P.p(x), lanes, P.p(y), lanes, head, frames, 0, C.c_void_p(s1.cuda_stream))
        if tail:
            call("biquad_i32_df1_pitch", C.cast(cfg, C.c_void_p), 1, P.p(stt), C.c_void_p(x.data_ptr() + head * 4), lanes, C.c_void_p(y.data_ptr() + head * 4), lanes, tail, frames, 0,
                 C.c_void_p(s2.cuda_stream))
            ev_j.record(s2)
            s1.wait_event(ev_j)
    res = {}
    for name, f in (("whole", whole), ("split", split)):
        for _ in range(30): f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s1); f(); b.record(s1); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        ts.sort(); res[name] = ts[len(ts) // 2]
    print(json.dumps({"lanes": lanes, "head": head, "tail": tail, "whole_ms": round(res["whole"], 4), "split_ms": round(res["split"], 4),
                      "whole_frac": round(8 * lanes * frames / res["whole"] / 8e9, 3), "split_frac": round(8 * lanes * frames / res["split"] / 8e9, 3)}), flush=True)
    del x, y
