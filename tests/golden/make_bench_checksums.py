#!/usr/bin/env python3
"""Make tests/golden/bench_checksums.json: the CPU oracle's checksums of bench.py's workloads.

For every input bench.py can be asked to run at the full configuration sizes — C2 (i32 DF1, FRAME_MAJOR
65536 x 4096) with the input stream of rank r = 0..7, and C5 (f32 DF2T, 2^20 lanes x 4096) per block of
131072 lanes — the oracle (oracle/idsp_oracle.c, threaded over lane blocks) runs ONE pass from zero state and the
wrapping i64 sums of the 32-bit words of y and of the written-back state are recorded.  bench.py compares every
rank's sums with these after its timed region (`integrity`).  Sums of lane blocks add up, so any contiguous split
of C5 at multiples of 131072 lanes can be checked.

C3 (HbfDec /16, 16384 lanes x 65536 input samples) and C4 (lock-in, 32768 lanes x 4096): one entry each (every rank of
bench.py runs the same lanes), from the oracle's `idsp_ref_hbf_dec_f32` / `idsp_ref_lockin_i32_process` over LANE_MAJOR lane
blocks on a thread pool (the sums do not depend on the layout the GPU ran: the same words in another order).

Run from the repo root (about ten minutes on 8 cores):  python tests/golden/make_bench_checksums.py [c3 c4]   (names: only those
sections are recomputed, the rest of the file is kept)
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (input generators and configuration only)
import oracle  # noqa: E402
from idsp_amd import _abi  # noqa: E402


def lane_blocks(call, x_lm, y_lm, state, block=256):
    """call(state_block, x_block, y_block, lanes_of_block) over LANE_MAJOR lane blocks on a thread pool (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor

    lanes = x_lm.shape[0]

    def work(lo):
        hi = min(lanes, lo + block)
        st = np.ascontiguousarray(state[:, lo:hi])
        assert call(st, x_lm[lo:hi], y_lm[lo:hi], hi - lo) == 0
        state[:, lo:hi] = st

    with ThreadPoolExecutor(os.cpu_count() or 1) as ex:
        list(ex.map(work, range(0, lanes, block)))


def c3_c4(out, which):
    from tests import _harness as H

    o = H.oracle()
    cs = lambda a: bench.wrap64(int(np.ascontiguousarray(a).reshape(-1).view(np.int32).sum(dtype=np.int64)))  # noqa: E731
    if "c3" in which:
        c3 = bench.CONFIGS["c3"]
        lanes, frames, R = c3["lanes"], c3["frames"], c3["rate"]
        cfg = _abi.HbfCascadeF32()
        assert o.fn["hbf_dec_cascade"](0, 4, C.byref(cfg)) == 0
        x = np.empty((lanes, frames, R), np.float32)
        bench.blockwise(lambda f0, f1: bench.c3_input(np, 0, lanes, f0, f1, "lane", None, R), np, x, lanes, frames, "lane", R)
        x = x.reshape(lanes, frames * R)
        y = np.empty((lanes, frames), np.float32)
        st = np.zeros((c3["state_words"], lanes), np.uint32)
        lane_blocks(lambda s, a, b, n: o.cfgcall("hbf_dec_f32", cfg, s, a, b, n, frames, H.LM), x, y, st)
        out["c3"] = {"lanes": lanes, "frames": frames, "rate": R, "ranks": {"0": {"y": cs(y), "state": cs(st)}}}
        print("c3", out["c3"]["ranks"], flush=True)
    if "c4" in which:
        c4 = bench.CONFIGS["c4"]
        lanes, frames = c4["lanes"], c4["frames"]
        cfg = H.lockin_cfg([bench.lockin_k(), bench.lockin_k()])
        x = np.empty((lanes, frames), np.int32)
        bench.blockwise(lambda f0, f1: bench.c4_input(np, 0, lanes, f0, f1, "lane"), np, x, lanes, frames, "lane")
        y = np.empty((lanes, frames, 2), np.int32)
        st = np.zeros((c4["state_words"], lanes), np.uint32)
        st[1] = bench.c4_steps(np, 0, lanes).view(np.uint32)
        lane_blocks(lambda s, a, b, n: o.cfgcall("lockin_i32_process", cfg, s, a, b, n, frames, H.LM), x, y, st)
        out["c4"] = {"lanes": lanes, "frames": frames, "ranks": {"0": {"y": cs(y), "state": cs(st)}}}
        print("c4", out["c4"]["ranks"], flush=True)


def main():
    path = os.path.join(ROOT, "tests", "golden", "bench_checksums.json")
    only = [a for a in sys.argv[1:] if a in ("c3", "c4")]
    if only:
        with open(path) as fh:
            out = json.load(fh)
        c3_c4(out, only)
        with open(path, "w") as fh:
            json.dump(out, fh, indent=1)
            fh.write("\n")
        return
    lib = oracle.load()
    mt = lib.idsp_ref_biquad_mt_reps
    mt.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                   C.c_int, C.c_int, C.c_int]
    threads = os.cpu_count() or 1
    sos = (C.c_double * 6)(*bench.lowpass_sos(bench.F0))

    def sums(kind, rec, words, x, lanes, frames):
        y = np.empty_like(x)
        st = np.zeros((words, lanes), np.uint32)
        assert mt(kind, C.byref(rec), 1, st.ctypes.data, x.ctypes.data, y.ctypes.data, lanes, frames, 0, threads, 1) == 0
        cs = lambda a: bench.wrap64(int(a.reshape(-1).view(np.int32).sum(dtype=np.int64)))  # noqa: E731
        return {"y": cs(y), "state": cs(st)}

    out = {"note": "CPU oracle, one pass from zero state, wrapping i64 sums of 32-bit words; made by tests/golden/make_bench_checksums.py"}
    c2 = bench.CONFIGS["c2"]
    q = _abi.BiquadI32()
    assert lib.idsp_ref_biquad_i32_from_sos(sos, bench.FRAC, C.byref(q)) == 0
    out["c2"] = {"layout": "frame", "lanes": c2["lanes"], "frames": c2["frames"], "ranks": {}}
    for r in range(8):
        x = bench.c2_input_host(c2["frames"], c2["lanes"], "frame", r)
        out["c2"]["ranks"][str(r)] = sums(0, q, 4, x, c2["lanes"], c2["frames"])
        print("c2 rank", r, out["c2"]["ranks"][str(r)], flush=True)
    c5 = bench.CONFIGS["c5"]
    f = _abi.BiquadF32()
    assert lib.idsp_ref_biquad_f32_from_sos_f64(sos, C.byref(f)) == 0
    out["c5"] = {"layout": "frame", "lanes": c5["lanes"], "frames": c5["frames"], "block_lanes": bench.C5_BLOCK, "blocks": []}
    for b in range(c5["lanes"] // bench.C5_BLOCK):
        x = bench.c5_input_host(b * bench.C5_BLOCK, bench.C5_BLOCK, c5["frames"], "frame")
        out["c5"]["blocks"].append(sums(1, f, 2, x, bench.C5_BLOCK, c5["frames"]))
        print("c5 block", b, out["c5"]["blocks"][-1], flush=True)
    c3_c4(out, ("c3", "c4"))
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    main()
