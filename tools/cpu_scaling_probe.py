#!/usr/bin/env python3
"""How does the CPU oracle's threaded i32 DF1 loop scale on this host?  Prints the cgroup CPU quota, the affinity mask
size and the rate at 1 .. nproc threads (LANE_MAJOR, 65536 x 4096, the bench's C2 tensor shape).  Diagnostic for
bench.py's cpu_baseline (round 2 measured 4x from 256 threads)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import oracle  # noqa: E402  (diagnostic of the CPU baseline leg)
from idsp_amd import _abi  # noqa: E402

for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(f, "=", open(f).read().strip())
    except OSError as e:
        print(f, "unreadable:", e)
print("affinity cpus:", len(os.sched_getaffinity(0)), "os.cpu_count:", os.cpu_count())
lib = oracle.load(native=True)
mt = lib.idsp_ref_biquad_mt_reps
mt.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int]
q = _abi.BiquadI32()
lib.idsp_ref_biquad_i32_from_sos((C.c_double * 6)(*bench.lowpass_sos(bench.F0)), 30, C.byref(q))
lanes, frames = 65536, 4096
x = bench.c2_input_host(frames, lanes, "lane", 0)
y = np.zeros_like(x)
for layout in (1, 0):
    for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        if th > (os.cpu_count() or 1):
            break
        st = np.zeros((4, lanes), np.uint32)
        mt(0, C.byref(q), 1, st.ctypes.data, x.ctypes.data, y.ctypes.data, lanes, frames, layout, th, 1)
        reps = 2 if th < 8 else 8
        t0 = time.perf_counter()
        mt(0, C.byref(q), 1, st.ctypes.data, x.ctypes.data, y.ctypes.data, lanes, frames, layout, th, reps)
        dt = time.perf_counter() - t0
        print(f"layout {'LM' if layout else 'FM'} threads {th:4d}: {reps * lanes * frames / dt / 1e6:10.1f} Msamples/s  ({reps * lanes * frames * 8 / dt / 1e9:7.1f} GB/s)", flush=True)
