#!/usr/bin/env python3
"""Headline benchmark: i32 DF1 biquad, 65536 lanes x 4096 samples per lane,
shared coefficients (BASELINE.json configs[1], SURVEY.md §8d "C2").

A step = one pass of the hot path over one batch: `idsp_biquad_i32_df1` on a
FRAME_MAJOR `[[i32; 65536]; 4096]` tensor (1 GiB in, 1 GiB out), state carried
from step to step like consecutive `block()` calls.  Inputs are resident in HBM
before the timed region.  With --gpus N (launched by torch.distributed.run, one
rank per GPU) every rank runs the same per-GPU workload on its own lane shard
(weak scaling); lanes never interact, so there is no data-path collective —
only the barriers that bracket the timed region and the MAX all-reduce of the
elapsed time.

Prints ONE JSON line on rank 0 (see the task's bench contract); `roofline`
prices the kernel against HBM (8 B of algorithmic traffic per sample) with the
kernel duration measured by HIP events on the launch stream; `cpu_baseline`
times the CPU oracle (a port of the reference's scalar loop — the reference is
Rust and cannot be built here) on the host cores of the same box.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LANES = 65536
FRAMES = 4096
FRAC = 30
F0 = 0.01
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_SAMPLE = 8   # 4 B read + 4 B written
STATE_WORDS = 4        # DirectForm1<i32>: x0 x1 y0 y1


def lowpass_sos(f0: float):
    """coefficients::Filter::default().critical_frequency(f0).lowpass() in f64
    (src/iir/coefficients.rs:259-283), gain 1, Q = 1/sqrt(2)."""
    w0 = math.tau * f0
    fsin, fcos = math.sin(w0), math.cos(w0)
    alpha = 0.5 * fsin * math.sqrt(2.0)
    b = 0.5 * (1.0 - fcos)
    return [b, 2.0 * b, b, 1.0 + alpha, -2.0 * fcos, 1.0 - alpha]


def cpu_baseline(seconds_budget: float = 12.0):
    """Time the CPU oracle (kind "port") on a bounded sample of the workload.

    Sample: 8192 of the 65536 lanes x 4096 samples, LANE_MAJOR (each lane a
    contiguous slice — the layout `Lanes::process_view` walks,
    dsp-process/src/compose.rs:478-494), same coefficients and input
    distribution; repeated until ~seconds_budget of CPU time, once on all host
    cores (lane blocks per thread) and once on one thread (the reference's
    serial lane loop)."""
    import numpy as np

    import oracle  # cpu_baseline leg only
    from idsp_amd import _abi

    try:
        lib = oracle.load(native=True)  # -march=native, built on this host
    except Exception:
        lib = oracle.load()
    fn = lib.idsp_ref_biquad_i32_df1_mt
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_int]
    fn.restype = C.c_int
    cfg = _abi.BiquadI32()
    lib.idsp_ref_biquad_i32_from_sos((C.c_double * 6)(*lowpass_sos(F0)), FRAC, C.byref(cfg))
    lanes, frames = 8192, FRAMES
    rng = np.random.default_rng(2)
    x = rng.integers(-(1 << 24), 1 << 24, size=lanes * frames, dtype=np.int32)
    y = np.empty_like(x)
    st = np.zeros((STATE_WORDS, lanes), dtype=np.uint32)
    cores = os.cpu_count() or 1

    def run(threads, budget):
        n, t0 = 0, time.perf_counter()
        while True:
            rc = fn(C.byref(cfg), 1, st.ctypes.data, x.ctypes.data, y.ctypes.data, lanes, frames, 1, threads)
            assert rc == 0
            n += 1
            dt = time.perf_counter() - t0
            if dt > budget:
                return n * lanes * frames / dt / 1e6

    all_cores = run(cores, seconds_budget * 0.6)
    one = run(1, seconds_budget * 0.4)
    return {
        "value": round(all_cores, 1), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "single_thread_value": round(one, 1),
        "sample": f"{lanes} of {LANES} lanes x {frames} samples, LANE_MAJOR, C oracle -O3 -march=native, "
                  f"repeated ~{seconds_budget:.0f} s; reference is Rust (no toolchain here)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=100)  # ~35 ms: the clocks take tens of ms to settle after an idle gap
    ap.add_argument("--layout", choices=["frame", "lane"], default="frame")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    # RCCL ("nccl" on ROCm) is the backend; IDSP_BENCH_BACKEND=gloo exists only so the multi-rank
    # control flow can be exercised on a single-GPU box (ranks then share the device).
    backend = os.environ.get("IDSP_BENCH_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from idsp_amd import _abi
    from idsp_amd._lib import call

    # Weak scaling: every rank owns a full 65536-lane shard of a world*65536-lane job.
    lanes, frames = LANES, FRAMES
    layout = _abi.FRAME_MAJOR if args.layout == "frame" else _abi.LANE_MAJOR
    cfg = _abi.BiquadI32()
    call("biquad_i32_from_sos", (C.c_double * 6)(*lowpass_sos(F0)), FRAC, C.byref(cfg))

    gen = torch.Generator(device=dev)
    gen.manual_seed(2 + rank)
    # Input and output come from one allocation with y - x = 1 GiB + 48 KiB.  The kernel reads x and writes y at the same
    # offsets at the same time, and how the two streams interleave over the HBM channels depends on (y - x): measured
    # 0.347-0.353 ms at +1 KiB / +16 KiB / +48 KiB / +128 KiB, 0.39-0.40 ms at +0 (also in place), +32 KiB, +256 KiB,
    # +512 KiB.  Two separate torch allocations land on either kind of offset from process to process.
    n = frames * lanes
    pad = (48 << 10) // 4
    arena = torch.empty(2 * n + pad, dtype=torch.int32, device=dev)
    x = arena[:n]
    x.copy_(torch.randint(-(1 << 24), 1 << 24, (n,), dtype=torch.int32, device=dev, generator=gen))
    y = arena[n + pad:]
    state = torch.zeros((STATE_WORDS, lanes), dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    sptr = C.c_void_p(stream.cuda_stream)
    cfgs = (_abi.BiquadI32 * 1)(cfg)

    def step():
        call("biquad_i32_df1", C.cast(cfgs, C.c_void_p), 1, C.c_void_p(state.data_ptr()), C.c_void_p(x.data_ptr()),
             C.c_void_p(y.data_ptr()), lanes, frames, layout, sptr)

    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)  # HIP events on the launch stream itself
        step()
        b.record(stream)
    stream.synchronize()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
    samples_step = lanes * frames
    alg_bytes = samples_step * BYTES_PER_SAMPLE + 2 * STATE_WORDS * 4 * lanes
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0

    # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command
    # (profiles/bench_c2_traffic.json; FETCH_SIZE x2 gfx950 correction, WRITE_SIZE as is)
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "bench_c2_traffic.json")) as f:
            tj = json.load(f)
        if layout == _abi.FRAME_MAJOR:
            traffic, traffic_src = tj["traffic_bytes_per_launch"], tj["source"]
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        out = {
            "metric": "i32_df1_biquad_64k_lanes_throughput",
            "value": round(world * samples_step * args.steps / elapsed / 1e6, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i32",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: 65536-lane i32 Biquad DF1 (Q30 lowpass f0=0.01), shared coeffs, "
                            "4096 samples/lane, per GPU",
                "lanes_per_gpu": lanes, "frames": frames, "layout": "FrameMajor" if layout == 0 else "LaneMajor",
                "parallelism": f"lane-split x{world}, no data-path collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "stream_frame_major_lds<Chain<Df1I32<false>,1>>" if layout == 0 else "stream_lane_major<Chain<Df1I32<false>,1>>",
                "kernel_ms": round(kern_ms, 4), "algorithmic_bytes": alg_bytes,
            },
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
