#!/usr/bin/env python3
"""Benchmarks of the hot path on MI355X, one JSON line per run (rank 0).

--config c2 (default; BASELINE.json configs[1], SURVEY.md §8d "C2", the headline metric):
    i32 DF1 biquad, 65536 lanes x 4096 samples per lane, shared Q30 coefficients,
    `idsp_biquad_i32_df1` on a FRAME_MAJOR `[[i32; 65536]; 4096]` tensor (1 GiB in, 1 GiB out).
    With --gpus N every rank runs this per-GPU workload on its own lanes (WEAK scaling).
--config c5 (BASELINE.json configs[4], "C5"):
    f32 DF2T biquad over 2^20 lanes x 4096 samples (16 GiB in, 16 GiB out in total), the lanes split
    contiguously over the ranks by idsp_amd.sharding.lane_shard (STRONG scaling: total work fixed).

A step = one pass of the hot path over the rank's batch, state carried from step to step like
consecutive `block()` calls (dsp-process/src/process.rs:122-127).  Inputs are resident in HBM before
the timed region.  Lanes never interact (dsp-process/src/compose.rs:468-494), so there is no data-path
collective — only the barriers that bracket the timed region and the MAX all-reduce of the elapsed time.

Timing: W untimed warm-up steps, then further untimed steps until --settle-ms of wall time have passed
(the first launches after an idle gap run at a lower clock: round 1's driver run with 5 warm-up steps read
0.403 ms where the steady state is 0.337 ms), then EXACTLY K timed steps between barrier +
synchronize pairs; `value` comes from that wall-clock interval (max over ranks).  Every timed step is also
bracketed by HIP events on the launch stream: `roofline` prices the MEDIAN of those kernel durations
against HBM (8 B of algorithmic traffic per sample + the state planes once each way), and reports the
minimum and the mean beside it.  x and y are two plain, separate allocations.

`cpu_baseline` times the CPU oracle (a C port of the reference's scalar loop — the reference is Rust and
cannot be built here) on the host cores of the same box, on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
F0 = 0.01
FRAC = 30

CONFIGS = {
    "c2": dict(
        metric="i32_df1_biquad_64k_lanes_throughput", entry="biquad_i32_df1", dtype="i32", lanes=65536, frames=4096,
        scaling="weak", state_words=4, bytes_per_sample=8, seed=2,
        workload="configs[1]: 65536-lane i32 Biquad DF1 (Q30 lowpass f0=0.01), shared coeffs, 4096 samples/lane, per GPU",
    ),
    "c5": dict(
        metric="f32_df2t_biquad_1M_lanes_throughput", entry="biquad_f32_df2t", dtype="f32", lanes=1 << 20, frames=4096,
        scaling="strong", state_words=2, bytes_per_sample=8, seed=5,
        workload="configs[4]: 2^20-lane f32 Biquad DF2T (lowpass f0=0.01), shared coeffs, 4096 samples/lane, "
                 "lanes split contiguously over the GPUs",
    ),
}


def lowpass_sos(f0: float):
    """coefficients::Filter::default().critical_frequency(f0).lowpass() in f64
    (src/iir/coefficients.rs:259-283), gain 1, Q = 1/sqrt(2)."""
    w0 = math.tau * f0
    fsin, fcos = math.sin(w0), math.cos(w0)
    alpha = 0.5 * fsin * math.sqrt(2.0)
    b = 0.5 * (1.0 - fcos)
    return [b, 2.0 * b, b, 1.0 + alpha, -2.0 * fcos, 1.0 - alpha]


def job_shard(cfg: dict, rank: int, world: int, lanes_override: int | None = None):
    """(first lane, lane count) of `rank`: weak scaling = a full per-GPU workload each,
    strong scaling = the contiguous block lane_shard() assigns."""
    from idsp_amd.sharding import lane_shard

    lanes = lanes_override or cfg["lanes"]
    if cfg["scaling"] == "weak":
        return rank * lanes, lanes
    lo, hi = lane_shard(lanes, rank, world)
    return lo, hi - lo


def algorithmic_bytes(cfg: dict, lanes: int, frames: int) -> int:
    """4 B read + 4 B written per sample, plus each state plane once in and once out."""
    return lanes * frames * cfg["bytes_per_sample"] + 2 * cfg["state_words"] * 4 * lanes


class HipEngine:
    """The product path: device buffers from torch, launches through the C ABI on one HIP stream."""

    def __init__(self, cfg: dict, lanes: int, frames: int, layout: str, seed: int, device_index: int):
        import torch

        from idsp_amd import _abi
        from idsp_amd._lib import call, load

        self.torch, self.call = torch, call
        self.fn, _ = load()
        self.dev = torch.device("cuda", device_index)
        self.lanes, self.frames = lanes, frames
        self.layout = _abi.FRAME_MAJOR if layout == "frame" else _abi.LANE_MAJOR
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(seed)
        n = lanes * frames
        sos = (C.c_double * 6)(*lowpass_sos(F0))
        if cfg["dtype"] == "i32":
            rec = _abi.BiquadI32()
            call("biquad_i32_from_sos", sos, FRAC, C.byref(rec))
            self.cfgs = (_abi.BiquadI32 * 1)(rec)
            self.x = torch.randint(-(1 << 24), 1 << 24, (n,), dtype=torch.int32, device=self.dev, generator=gen)
        else:
            rec = _abi.BiquadF32()
            call("biquad_f32_from_sos_f64", sos, C.byref(rec))
            self.cfgs = (_abi.BiquadF32 * 1)(rec)
            self.x = torch.empty(n, dtype=torch.float32, device=self.dev)
            self.x.normal_(generator=gen)
        self.y = torch.empty_like(self.x)  # a plain second allocation: no placement tuning
        self.state = torch.zeros((cfg["state_words"], lanes), dtype=torch.int32, device=self.dev)
        self.stream = torch.cuda.Stream(device=self.dev)
        self.entry = cfg["entry"]
        self._args = (C.cast(self.cfgs, C.c_void_p), 1, C.c_void_p(self.state.data_ptr()), C.c_void_p(self.x.data_ptr()),
                      C.c_void_p(self.y.data_ptr()), lanes, frames, self.layout, C.c_void_p(self.stream.cuda_stream))
        self.sync()

    def step(self):
        self.call(self.entry, *self._args)

    def sync(self):
        self.stream.synchronize()
        self.torch.cuda.synchronize()

    def timed_steps(self, k: int):
        """k launches, each between two HIP events recorded on the launch stream itself; returns a
        function that yields the k kernel durations in ms once the stream has been synchronised."""
        ev = [(self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)) for _ in range(k)]
        for a, b in ev:
            a.record(self.stream)
            self.step()
            b.record(self.stream)
        return lambda: [a.elapsed_time(b) for a, b in ev]

    def kernel_name(self) -> str:
        return self.fn["last_kernel"]().decode(errors="replace")

    def reduce_device(self, backend: str):
        return self.dev if backend == "nccl" else "cpu"


def run_timed(engine, steps: int, warmup: int, settle_ms: float, dist=None):
    """The contract's timed region.  Returns (elapsed seconds over exactly `steps` steps on this rank,
    per-step kernel durations in ms, untimed steps actually run)."""
    engine.sync()
    t_w = time.perf_counter()
    done = 0
    for _ in range(warmup):
        engine.step()
        done += 1
    engine.sync()
    while (time.perf_counter() - t_w) * 1e3 < settle_ms:  # clocks settle by time, not by launch count
        for _ in range(8):
            engine.step()
        done += 8
        engine.sync()
    if dist:
        dist.barrier()
    engine.sync()
    t0 = time.perf_counter()
    durations = engine.timed_steps(steps)
    engine.sync()
    if dist:
        dist.barrier()
    engine.sync()
    return time.perf_counter() - t0, durations(), done


def cpu_baseline(cfg: dict, seconds_budget: float = 12.0):
    """Time the CPU oracle (kind "port") on a bounded sample of the workload: 8192 lanes x 4096 samples,
    LANE_MAJOR (each lane a contiguous slice — what `Lanes::process_view` walks,
    dsp-process/src/compose.rs:478-494), same coefficients and input distribution; repeated until
    ~seconds_budget of wall time, once with the lanes dealt to all host cores (one block per thread)
    and once on one thread (the reference's serial lane loop)."""
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    import oracle  # cpu_baseline leg only
    from idsp_amd import _abi

    try:
        lib = oracle.load(native=True)  # -march=native, built on this host
        flavour = "-O3 -march=native"
    except Exception:
        lib = oracle.load()
        flavour = "-O3"
    fn = getattr(lib, "idsp_ref_" + cfg["entry"])
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
    fn.restype = C.c_int
    sos = (C.c_double * 6)(*lowpass_sos(F0))
    if cfg["dtype"] == "i32":
        rec = _abi.BiquadI32()
        lib.idsp_ref_biquad_i32_from_sos(sos, FRAC, C.byref(rec))
    else:
        rec = _abi.BiquadF32()
        lib.idsp_ref_biquad_f32_from_sos_f64(sos, C.byref(rec))
    lanes, frames = 8192, cfg["frames"]
    rng = np.random.default_rng(cfg["seed"])
    if cfg["dtype"] == "i32":
        x = rng.integers(-(1 << 24), 1 << 24, size=(lanes, frames), dtype=np.int32)
    else:
        x = rng.standard_normal(size=(lanes, frames), dtype=np.float32)
    y = np.empty_like(x)
    cores = os.cpu_count() or 1

    def run(threads, budget):
        from idsp_amd.sharding import lane_shard

        blocks = [lane_shard(lanes, t, threads) for t in range(threads)]
        states = [np.zeros((cfg["state_words"], hi - lo), dtype=np.uint32) for lo, hi in blocks]

        def work(i):
            lo, hi = blocks[i]
            if hi > lo:
                rc = fn(C.byref(rec), 1, states[i].ctypes.data, x[lo:hi].ctypes.data, y[lo:hi].ctypes.data, hi - lo, frames, 1)
                assert rc == 0

        n, t0 = 0, time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as pool:  # ctypes releases the GIL during the call
            while True:
                list(pool.map(work, range(threads)))
                n += 1
                dt = time.perf_counter() - t0
                if dt > budget:
                    return n * lanes * frames / dt / 1e6

    all_cores = run(min(cores, lanes), seconds_budget * 0.6)
    one = run(1, seconds_budget * 0.4)
    return {
        "value": round(all_cores, 1), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "single_thread_value": round(one, 1),
        "sample": f"{lanes} of {cfg['lanes']} lanes x {frames} samples, LANE_MAJOR, C oracle {flavour}, "
                  f"repeated ~{seconds_budget:.0f} s; reference is Rust (no toolchain here)",
    }


def committed_traffic(config: str, kernel: str):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/bench_<config>_traffic.json; FETCH_SIZE x2 gfx950 correction, WRITE_SIZE as is).  It is a
    profile of an earlier run of the same command, NOT a live counter: only quoted when the kernel that
    ran now is the kernel that was profiled, and labelled as coming from the file."""
    try:
        with open(os.path.join(ROOT, "profiles", f"bench_{config}_traffic.json")) as f:
            tj = json.load(f)
        if tj.get("kernel_prefix") and not kernel.startswith(tj["kernel_prefix"]):
            return None, None
        return tj["traffic_bytes_per_launch"], "file: " + tj["source"]
    except (OSError, KeyError, ValueError):
        return None, None


def report(cfg_name, cfg, args, world, lanes_rank, frames, elapsed, kern_ms, untimed, kernel, total_lanes):
    samples_all = total_lanes * frames  # samples per step over all ranks
    alg_bytes = algorithmic_bytes(cfg, lanes_rank, frames)
    med = statistics.median(kern_ms) if kern_ms else 0.0
    achieved = alg_bytes / (med * 1e-3) / 1e9 if med > 0 else 0.0
    traffic, traffic_src = committed_traffic(cfg_name if args.layout == "frame" else cfg_name + "_lane", kernel) if not args.lanes else (None, None)
    return {
        "metric": cfg["metric"],
        "value": round(samples_all * args.steps / elapsed / 1e6, 1),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
        "higher_is_better": True,
        "scaling": cfg["scaling"],
        "vs_baseline": None,
        "dtype": cfg["dtype"],
        "data": "synthetic",
        "config": {
            "workload": cfg["workload"], "name": cfg_name,
            "lanes_total": total_lanes, "lanes_per_gpu": lanes_rank, "frames": frames,
            "layout": "FrameMajor" if args.layout == "frame" else "LaneMajor",
            "parallelism": f"lane-split x{world}, no data-path collective",
            "buffers": "two separate plain allocations", "untimed_steps": untimed, "settle_ms": args.settle_ms,
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "kernel": kernel, "kernel_ms": round(med, 4), "kernel_ms_min": round(min(kern_ms), 4) if kern_ms else None,
            "kernel_ms_mean": round(sum(kern_ms) / len(kern_ms), 4) if kern_ms else None,
            "kernel_ms_stat": "median of the per-step HIP-event durations (rank 0)", "algorithmic_bytes": alg_bytes,
        },
    }


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="keep running untimed steps until this much wall time has passed since the first warm-up step")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--layout", choices=["frame", "lane"], default="frame")
    ap.add_argument("--lanes", type=int, default=0, help="override the configuration's lane count (diagnostics)")
    ap.add_argument("--frames", type=int, default=0, help="override the samples per lane (diagnostics)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args(argv)

    import torch

    cfg = CONFIGS[args.config]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dist = None
    # RCCL ("nccl" on ROCm) is the backend; IDSP_BENCH_BACKEND=gloo exists only so the multi-rank
    # control flow can be exercised on a single-GPU box (ranks then share the device).
    backend = os.environ.get("IDSP_BENCH_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    frames = args.frames or cfg["frames"]
    _, lanes_rank = job_shard(cfg, rank, world, args.lanes or None)
    total_lanes = (args.lanes or cfg["lanes"]) * (world if cfg["scaling"] == "weak" else 1)
    engine = HipEngine(cfg, lanes_rank, frames, args.layout, cfg["seed"] + rank, local)
    elapsed, kern_ms, untimed = run_timed(engine, args.steps, args.warmup, args.settle_ms, dist)
    t = torch.tensor([elapsed], dtype=torch.float64, device=engine.reduce_device(backend))
    if dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        out = report(args.config, cfg, args, world, lanes_rank, frames, elapsed, kern_ms, untimed, engine.kernel_name(), total_lanes)
        out["cpu_baseline"] = cpu_baseline(cfg) if world == 1 and not args.no_cpu else None
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
