"""A/B timing of half-band decimator variants IN ONE PROCESS on the same buffers (GPU): every variant is a small library built by
tools/exp_hbf_blk.sh (build/exp_hbf_blk/libidsp_hip_<name>.so); the variants take turns, `--rounds` times, `--iters` launches each, so
that clock state, buffer placement and box are the same for all of them.  Timing only — parity is the tests' business.
usage: python tools/exp_hbf_ab.py [--layout lm|fm] [--rounds 5] [--iters 10] NAME [NAME ...]      (NAME "product" = the engine)"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idsp_amd import _abi  # noqa: E402
from idsp_amd._lib import load  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+")
ap.add_argument("--layout", default="lm")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--lanes", type=int, default=16384)
ap.add_argument("--frames", type=int, default=4096)
ap.add_argument("--stages", type=int, default=4)
ap.add_argument("--dir", default="build/exp_hbf_blk")
ap.add_argument("--zero", action="store_true", help="all-zero input (draws less power: a clock diagnostic, not a result)")
a = ap.parse_args()
fn, _ = load()
dev = torch.device("cuda", 0)
cfg = _abi.HbfCascadeF32()
assert fn["hbf_dec_cascade"](0, a.stages, C.byref(cfg)) == 0
R = 1 << a.stages
words = fn["hbf_dec_state_words"](C.byref(cfg))
x = torch.randn(a.lanes * a.frames * R, dtype=torch.float32, device=dev)
if a.zero:
    x.zero_()
y = torch.zeros(a.lanes * a.frames, dtype=torch.float32, device=dev)
st = torch.zeros((words, a.lanes), dtype=torch.int32, device=dev)
layout = 1 if a.layout == "lm" else 0
stream = torch.cuda.current_stream()


def entry(name):
    if name == "product":
        f = fn["hbf_dec_f32"]
        return lambda: f(C.byref(cfg), C.c_void_p(st.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), a.lanes, a.frames, layout, None)
    lib = C.CDLL(os.path.join(a.dir, f"libidsp_hip_{name}.so"))
    f = lib.idsp_hbf_dec_f32
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
    return lambda: f(C.byref(cfg), st.data_ptr(), x.data_ptr(), y.data_ptr(), a.lanes, a.frames, layout, None)


runs = {n: entry(n) for n in a.names}
# every variant from the same zero state: outputs and written-back state must equal the first one's bit for bit (a variant that differs is reported, not timed as a result)
ref = None
for n, r in runs.items():
    st.zero_()
    y.zero_()
    assert r() == 0
    torch.cuda.synchronize()
    got = (y.clone(), st.clone())
    if ref is None:
        ref = got
    else:
        same = bool(torch.equal(got[0].view(torch.int32), ref[0].view(torch.int32))) and bool(torch.equal(got[1], ref[1]))
        print(json.dumps({"variant": n, "bit_exact_with": a.names[0], "same": same}), flush=True)
for r in runs.values():  # warm-up: clocks, code objects
    for _ in range(20):
        assert r() == 0
torch.cuda.synchronize()
res = {n: [] for n in runs}
for _ in range(a.rounds):
    for n, r in runs.items():
        for _ in range(3):
            r()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
        for e0, e1 in evs:
            e0.record(stream)
            r()
            e1.record(stream)
        torch.cuda.synchronize()
        res[n].append(statistics.median(e0.elapsed_time(e1) for e0, e1 in evs))
alg = 4 * a.lanes * a.frames * R + 4 * a.lanes * a.frames
for n, v in res.items():
    med = statistics.median(v)
    print(json.dumps({"variant": n, "layout": a.layout, "ms_median_of_rounds": round(med, 4), "ms_rounds": [round(t, 4) for t in v],
                      "frac_hbm_peak": round(alg / (med * 1e-3) / 8e12, 4)}))
