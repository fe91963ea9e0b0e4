#!/usr/bin/env python3
"""Make tests/golden/bench_checksums.json: the CPU oracle's checksums of bench.py's workloads.

For every input bench.py can be asked to run at the full configuration sizes — C2 (i32 DF1, FRAME_MAJOR
65536 x 4096) with the input stream of rank r = 0..7, and C5 (f32 DF2T, 2^20 lanes x 4096) per block of
131072 lanes — the oracle (oracle/idsp_oracle.c, threaded over lane blocks) runs ONE pass from zero state and the
wrapping i64 sums of the 32-bit words of y and of the written-back state are recorded.  bench.py compares every
rank's sums with these after its timed region (`integrity`).  Sums of lane blocks add up, so any contiguous split
of C5 at multiples of 131072 lanes can be checked.

Run from the repo root (about ten minutes on 8 cores):  python tests/golden/make_bench_checksums.py
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


def main():
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
    with open(os.path.join(ROOT, "tests", "golden", "bench_checksums.json"), "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    main()
