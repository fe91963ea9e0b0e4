#!/usr/bin/env python3
"""Benchmarks of the hot path on MI355X, one JSON line per run (rank 0).

--config c2 (default; BASELINE.json configs[1], SURVEY.md §8d "C2", the headline metric):
    i32 DF1 biquad, 65536 lanes x 4096 samples per lane, shared Q30 coefficients,
    `idsp_biquad_i32_df1` on a FRAME_MAJOR `[[i32; 65536]; 4096]` tensor (1 GiB in, 1 GiB out).
    With --gpus N every rank runs this per-GPU workload on its own lanes (WEAK scaling).
    The line also carries a `c5` sub-object (below) so that one command yields both scaling curves.
    Beside `c5` the default line carries `c3` and `c4` sub-objects: the other two single-GPU configurations of BASELINE.json,
    each with its own timed region (<= 20 steps), `roofline` and `integrity`, so that one driver run measures all four.
--config c3 (BASELINE.json configs[2], "C3"):
    hbf::HbfDec /16 (HBF_TAPS stages 3,2,1,0; src/hbf.rs:385-421), f32, 16384 lanes x 65536 input samples per lane
    (4 GiB in, 256 MiB out), `idsp_hbf_dec_f32`; 4.25 algorithmic bytes per input sample.
--config c4 (BASELINE.json configs[3], "C4"):
    DDC lock-in Accu -> cossin -> mix -> [Lowpass<2>; 2] (src/lockin.rs:30-39), i32, 32768 lanes x 4096 samples, per-lane
    Accu step, `idsp_lockin_i32_process`; 12 algorithmic bytes per sample (4 in, Complex<i32> out).
    C3 / C4 inputs are counter hashes like C5's (below); with --gpus N every rank runs the same lanes (replicas of the
    per-GPU workload: WEAK scaling, one checksum on file serves every rank).
--config c5 (BASELINE.json configs[4], "C5"):
    f32 DF2T biquad over 2^20 lanes x 4096 samples (16 GiB in, 16 GiB out in total), the lanes split
    contiguously over the ranks by idsp_amd.sharding.lane_shard (STRONG scaling: total work fixed).

Ranks.  `python bench.py --gpus N` with no torchrun environment starts N ranks itself (it re-executes
this file under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`);
under an external torchrun (WORLD_SIZE set) it is one of the ranks.  One rank per GPU over RCCL (the
"nccl" backend on ROCm); when the box has fewer GPUs than ranks the ranks share devices and rendezvous
over gloo — a control-flow check, said so in the line (`ranks_share_device`).  `group_ranks` is the
result of a 1-int SUM all-reduce over the process group; `rccl_ranks` repeats it only when that group is an RCCL
communicator (null over gloo and for a single process without a group).

The default line also carries `inplace` sub-objects (C2 and C5 with y == x: `SplitInplace::inplace`,
dsp-process/src/process.rs:135-142, the mode the reference itself benchmarks) and ends with a compact `summary`
({config: [ms per step, roofline fraction, integrity match]}) as its LAST key.

A step = one pass of the hot path over the rank's batch, state carried from step to step like
consecutive `block()` calls (dsp-process/src/process.rs:122-127).  Inputs are resident in HBM before
the timed region.  Lanes never interact (dsp-process/src/compose.rs:468-494), so there is no data-path
collective — only the barriers that bracket the timed region, the MAX all-reduce of the elapsed time
and, outside the timed region, the 8-byte output-checksum all-reduce of SURVEY.md §8e.

Timing: W untimed warm-up steps, then further untimed steps until --settle-ms of wall time have passed
(the first launches after an idle gap run at a lower clock), then EXACTLY K timed steps between barrier +
synchronize pairs; `value` comes from that wall-clock interval (max over ranks).  The K launches are also
bracketed by ONE pair of HIP events on the launch stream: `roofline` prices that interval / K — the average
launch duration over the timed region — against HBM (8 B of algorithmic traffic per sample + the state
planes once each way); per-launch median / min come from further launches with their own event pairs,
after the timed region.  x and y are two plain, separate allocations.

Integrity (`integrity`): after the timed region every rank zeroes its state, runs ONE more step and sums
the 32-bit words of its output and of its written-back state (wrapping 64-bit sums).  The y sums meet in
`sharding.allreduce_checksum`; every rank's pair is compared with the CPU oracle's value for that rank's
input (tests/golden/bench_checksums.json, made by tests/golden/make_bench_checksums.py) and — at N = 1 —
with the oracle run live on the same tensor in the cpu_baseline leg.

Inputs.  C2: SURVEY.md §8d's stream, `numpy.random.default_rng(2).integers(-2^24, 2^24, (4096, 65536))`
on rank 0 (`default_rng([2, r])` on rank r > 0).  C5: a counter hash of (frame, global lane) evaluated on
the device (`c5_input`), uniform in [-1, 1) — the 2^32-sample tensor is then the same whatever the number of
ranks, which is what lets the strong-scaling checksums be compared across N.

`cpu_baseline` times the CPU oracle (a C port of the reference's scalar loops — the reference is Rust and
cannot be built here; the GPU box has no cargo / rustc either, profiles/r03_toolchain.txt) on the host cores of the
same box: pinned POSIX threads inside the C library over contiguous lane blocks, both layouts, with as many threads
as the container's CPU quota allows (`cores`) and with one thread.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
F0 = 0.01
FRAC = 30
MASK64 = (1 << 64) - 1
C5_BLOCK = 131072  # lanes per checksum block of C5 (= one rank's shard at N = 8)
C5_CPU_LANES = 16384  # lane prefix of C5 the cpu_baseline leg runs (and the live checksum comparison covers)

CONFIGS = {
    "c2": dict(
        metric="i32_df1_biquad_64k_lanes_throughput", entry="biquad_i32_df1", dtype="i32", lanes=65536, frames=4096,
        scaling="weak", state_words=4, bytes_per_sample=8, seed=2, oracle_kind=0,
        workload="configs[1]: 65536-lane i32 Biquad DF1 (Q30 lowpass f0=0.01), shared coeffs, 4096 samples/lane, per GPU",
    ),
    "c3": dict(
        metric="f32_hbf_dec16_16k_lanes_throughput", family="hbf_dec", entry="hbf_dec_f32", dtype="f32", lanes=16384, frames=4096,
        rate=16, scaling="weak", replicas=True, state_words=118, bytes_per_frame=68, samples_per_frame=16, seed=3,
        workload="configs[2]: hbf::HbfDec /16 (HBF_TAPS stages 3,2,1,0), f32, 16384 lanes x 65536 input samples per lane "
                 "(4096 output frames), per GPU",
    ),
    "c4": dict(
        metric="i32_lockin_32k_lanes_throughput", family="lockin", entry="lockin_i32_process", dtype="i32", lanes=32768, frames=4096,
        scaling="weak", replicas=True, state_words=18, bytes_per_frame=12, seed=4,
        workload="configs[3]: DDC lock-in Accu -> cossin -> mix -> [Lowpass<2>; 2] (f0 = 1e-3 fn), i32, 32768 lanes x 4096 "
                 "samples, per-lane Accu step, Complex<i32> out, per GPU",
    ),
    "c5": dict(
        metric="f32_df2t_biquad_1M_lanes_throughput", entry="biquad_f32_df2t", dtype="f32", lanes=1 << 20, frames=4096,
        scaling="strong", state_words=2, bytes_per_sample=8, seed=5, oracle_kind=1,
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
    if cfg.get("replicas"):
        return 0, lanes  # every rank the same lanes (C3 / C4)
    if cfg["scaling"] == "weak":
        return rank * lanes, lanes
    lo, hi = lane_shard(lanes, rank, world)
    return lo, hi - lo


def algorithmic_bytes(cfg: dict, lanes: int, frames: int) -> int:
    """biquads 4 B read + 4 B written per sample; HbfDec /16 4.25 B per input sample (16 x 4 B in + 4 B out
    per output frame); lock-in 4 B in + 8 B out per sample — plus each state plane once in and once out."""
    return lanes * frames * cfg.get("bytes_per_frame", cfg.get("bytes_per_sample", 8)) + 2 * cfg["state_words"] * 4 * lanes


# Read / write rates of a bare skeleton of nontemporal 16-byte loads / stores, 1024 waves (tools/ubench_cu_ceiling.hip, one box, committed:
# profiles/r06_ubench_cu_ceiling.txt): reads and writes share the HBM bus, and a copy runs as the sum of its directions.
READ_GBS_FILE, WRITE_GBS_FILE = 6900.0, 5450.0
WRITTEN_BYTES_PER_FRAME = {"c2": 4, "c3": 4, "c4": 8, "c5": 4}  # of `bytes_per_frame` / `bytes_per_sample`: the rest is read


def by_direction(cfg_name: str, cfg: dict, lanes: int, frames: int, kernel_ms: float):
    """The configuration's read and written bytes at the rates reads and writes reach each on their own: read time + write time, and the
    measured launch against it.  From file (a microbenchmark of another run), labelled so; the live yardstick of the line is `copy_gbs`."""
    w = WRITTEN_BYTES_PER_FRAME.get(cfg_name)
    if w is None or kernel_ms <= 0:
        return None
    total = cfg.get("bytes_per_frame", cfg.get("bytes_per_sample", 8))
    state = cfg["state_words"] * 4 * lanes
    rd, wr = lanes * frames * (total - w) + state, lanes * frames * w + state
    ms = (rd / READ_GBS_FILE + wr / WRITE_GBS_FILE) / 1e6
    return {"read_bytes": rd, "written_bytes": wr, "read_gbs": READ_GBS_FILE, "write_gbs": WRITE_GBS_FILE, "sum_of_directions_ms": round(ms, 4),
            "kernel_ms_over_it": round(kernel_ms / ms, 4), "source": "profiles/r06_ubench_cu_ceiling.txt (tools/ubench_cu_ceiling.hip, committed, another run)"}


def wrap64(v: int) -> int:
    """v modulo 2^64 as a signed 64-bit value (what a wrapping i64 sum holds)."""
    v &= MASK64
    return v - (1 << 64) if v >> 63 else v


# ---------------------------------------------------------------------------------------- inputs

def c2_input_host(frames: int, lanes: int, layout: str, rank: int):
    """SURVEY.md §8d C2 stream (PCG64): i32 uniform in [-2^24, 2^24); FRAME_MAJOR `[frames][lanes]`
    (LANE_MAJOR: the same tensor transposed, so that the layout-independent checksums on file hold for both)."""
    import numpy as np

    rng = np.random.default_rng(CONFIGS["c2"]["seed"] if rank == 0 else [CONFIGS["c2"]["seed"], rank])
    x = rng.integers(-(1 << 24), 1 << 24, size=(frames, lanes), dtype=np.int32)
    return x if layout == "frame" else np.ascontiguousarray(x.T)


def _c5_hash(idx, xp):
    """murmur3's 32-bit finaliser over a golden-ratio multiple of the index, in 64-bit lanes masked to
    32 bits (the same expression evaluates on numpy arrays and on torch tensors)."""
    m = 0xFFFFFFFF
    h = (idx * 0x9E3779B1 + 0x5BD1E995) & m
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & m
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & m
    h = h ^ (h >> 16)
    return h


def c5_input(xp, lane_lo: int, lanes: int, f_lo: int, f_hi: int, layout: str = "frame", device=None):
    """C5 samples of global lanes [lane_lo, lane_lo + lanes), frames [f_lo, f_hi): x = (h >> 8) * 2^-23 - 1
    with h = hash(frame * 2^20 + lane) — exact in f32, uniform in [-1, 1), independent of the sharding.
    `xp` is numpy or torch."""
    if xp.__name__ == "torch":
        f = xp.arange(f_lo, f_hi, dtype=xp.int64, device=device)
        l = xp.arange(lane_lo, lane_lo + lanes, dtype=xp.int64, device=device)
        idx = (f[:, None] << 20) + l[None, :] if layout == "frame" else (f[None, :] << 20) + l[:, None]
        return (_c5_hash(idx, xp) >> 8).to(xp.float32) * (2.0 ** -23) - 1.0
    f = xp.arange(f_lo, f_hi, dtype=xp.int64)
    l = xp.arange(lane_lo, lane_lo + lanes, dtype=xp.int64)
    idx = (f[:, None] << 20) + l[None, :] if layout == "frame" else (f[None, :] << 20) + l[:, None]
    return ((_c5_hash(idx, xp) >> 8).astype(xp.float32) * xp.float32(2.0 ** -23) - xp.float32(1.0)).astype(xp.float32)


def c3_input(xp, lane_lo: int, lanes: int, f_lo: int, f_hi: int, layout: str = "frame", device=None, rate: int = 16):
    """C3 input samples of lanes [lane_lo, lane_lo + lanes), output frames [f_lo, f_hi) (`rate` samples each): uniform in
    [-1, 1), x = (h >> 8) * 2^-23 - 1 with h = hash(2^32 + lane * 2^16 + sample).  FRAME_MAJOR [frame][lane][rate],
    LANE_MAJOR [lane][frame * rate]."""
    tor = xp.__name__ == "torch"
    kw = dict(device=device) if tor else {}
    f = xp.arange(f_lo, f_hi, dtype=xp.int64, **kw)
    l = xp.arange(lane_lo, lane_lo + lanes, dtype=xp.int64, **kw)
    k = xp.arange(rate, dtype=xp.int64, **kw)
    if layout == "frame":
        idx = (l[None, :, None] << 16) + f[:, None, None] * rate + k[None, None, :]
    else:
        idx = (l[:, None, None] << 16) + f[None, :, None] * rate + k[None, None, :]
    h = _c5_hash(idx + (1 << 32), xp) >> 8
    if tor:
        return h.to(xp.float32) * (2.0 ** -23) - 1.0
    return (h.astype(xp.float32) * xp.float32(2.0 ** -23) - xp.float32(1.0)).astype(xp.float32)


def c4_input(xp, lane_lo: int, lanes: int, f_lo: int, f_hi: int, layout: str = "frame", device=None):
    """C4 samples: i32 uniform in [-2^28, 2^28), x = (h >> 3) - 2^28 with h = hash(2^33 + frame * 2^20 + lane)."""
    tor = xp.__name__ == "torch"
    kw = dict(device=device) if tor else {}
    f = xp.arange(f_lo, f_hi, dtype=xp.int64, **kw)
    l = xp.arange(lane_lo, lane_lo + lanes, dtype=xp.int64, **kw)
    idx = (f[:, None] << 20) + l[None, :] if layout == "frame" else (f[None, :] << 20) + l[:, None]
    v = (_c5_hash(idx + (1 << 33), xp) >> 3) - (1 << 28)
    return v.to(xp.int32) if tor else v.astype(xp.int32)


def c4_steps(xp, lane_lo: int, lanes: int, device=None):
    """Per-lane `Accu` step (src/accu.rs:16-41): all 32 bits of hash(2^34 + lane), as i32."""
    tor = xp.__name__ == "torch"
    l = xp.arange(lane_lo, lane_lo + lanes, dtype=xp.int64, **(dict(device=device) if tor else {}))
    h = _c5_hash(l + (1 << 34), xp)
    h = h - ((h >> 31) << 32)  # two's complement view of the 32-bit pattern
    return h.to(xp.int32) if tor else h.astype(xp.int32)


def lockin_k():
    """[Lowpass<2>; 2] configuration of C4: k = pi 2^31 f0 / fn with f0 = 1e-3 fn; [k^2 / 2^32, -k sqrt 2] (src/lowpass.rs:29-46)."""
    k = math.pi * (1 << 31) * 1e-3
    return [int(k * k / (1 << 32)), -int(k * math.sqrt(2.0))]


def blockwise(gen, xp, out, lanes, frames, layout, per_frame=1):
    """Fill `out` ([frames][lanes](xR) or [lanes][frames](xR)) from gen(f0, f1) in pieces of ~16 M elements."""
    step = max(1, (1 << 24) // max(lanes * per_frame, 1))
    for f0 in range(0, frames, step):
        f1 = min(frames, f0 + step)
        if layout == "frame":
            out[f0:f1] = gen(f0, f1)
        else:
            out[:, f0:f1] = gen(f0, f1)
    return out


def c5_input_host(lane_lo: int, lanes: int, frames: int, layout: str = "frame"):
    """The C5 samples of a lane block on the host (numpy), built in pieces of ~4 M samples."""
    import numpy as np

    out = np.empty((frames, lanes) if layout == "frame" else (lanes, frames), dtype=np.float32)
    step = max(1, (1 << 22) // max(lanes, 1))
    for f0 in range(0, frames, step):
        f1 = min(frames, f0 + step)
        blk = c5_input(np, lane_lo, lanes, f0, f1, layout)
        if layout == "frame":
            out[f0:f1] = blk
        else:
            out[:, f0:f1] = blk
    return out


def expected_checksums():
    """Oracle checksums of the bench workloads (data, made offline by tests/golden/make_bench_checksums.py)."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "bench_checksums.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def expected_for(cfg_name: str, layout: str, rank: int, lane_lo: int, lanes: int, overridden: bool):
    """[y checksum, state checksum] the oracle gives for this rank's input, or None if not on file."""
    # the wrapping sums do not depend on the order of the words, and every config's LaneMajor input is the FrameMajor tensor
    # transposed: one table serves both layouts (C5's blocks are FrameMajor lane blocks: FrameMajor only)
    if overridden or (layout != "frame" and cfg_name == "c5"):
        return None
    tab = expected_checksums().get(cfg_name)
    if not tab:
        return None
    if cfg_name == "c2":
        e = tab.get("ranks", {}).get(str(rank))
        return [int(e["y"]), int(e["state"])] if e else None
    if cfg_name in ("c3", "c4"):  # every rank runs the same lanes
        e = tab.get("ranks", {}).get("0")
        return [int(e["y"]), int(e["state"])] if e else None
    blocks = tab.get("blocks", [])
    if tab.get("block_lanes") != C5_BLOCK or lane_lo % C5_BLOCK or lanes % C5_BLOCK:
        return None
    ids = range(lane_lo // C5_BLOCK, (lane_lo + lanes) // C5_BLOCK)
    if not ids or ids[-1] >= len(blocks):
        return None
    return [wrap64(sum(int(blocks[i]["y"]) for i in ids)), wrap64(sum(int(blocks[i]["state"]) for i in ids))]


# ---------------------------------------------------------------------------------------- engine

class HipEngine:
    """The product path: device buffers from torch, launches through the C ABI on one HIP stream."""

    FAMILIES = ("biquad", "hbf_dec", "lockin")  # configuration families this engine runs (rank_main skips the others' sub-objects)

    @staticmethod
    def device_setup(local_rank: int):
        """Bind this rank to its GPU; returns (local device index, number of visible devices)."""
        import torch

        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
        n = torch.cuda.device_count()
        torch.cuda.set_device(local_rank % n)
        return local_rank % n, n

    GUARD = True  # rank_main runs tools/perf_guard.py behind the configurations (GPU only)
    INPLACE = True  # this engine runs the `y == x` sub-objects (SplitInplace::inplace, dsp-process/src/process.rs:135-142)

    def __init__(self, cfg_name: str, cfg: dict, lane_lo: int, lanes: int, frames: int, layout: str, rank: int,
                 device_index: int, inplace: bool = False):
        import torch

        from idsp_amd import _abi
        from idsp_amd._lib import call, load

        self.torch, self.call = torch, call
        self.fn, _ = load()
        self.dev = torch.device("cuda", device_index)
        self.lanes, self.frames, self.frame_major = lanes, frames, layout == "frame"
        self.layout = _abi.FRAME_MAJOR if self.frame_major else _abi.LANE_MAJOR
        self.x_host = None
        self.family = cfg.get("family", "biquad")
        self.state_init = None  # rows of the state that are not zero at the start of a stream: {row: tensor}
        n_sections = 1
        if self.family == "hbf_dec":
            R = cfg["rate"]
            self.cfgs = _abi.HbfCascadeF32()
            call("hbf_dec_cascade", 0, R.bit_length() - 1, C.byref(self.cfgs))
            assert self.fn["hbf_dec_state_words"](C.byref(self.cfgs)) == cfg["state_words"]
            self.x = torch.empty(lanes * frames * R, dtype=torch.float32, device=self.dev)
            xv = self.x.view(frames, lanes, R) if self.frame_major else self.x.view(lanes, frames, R)
            blockwise(lambda f0, f1: c3_input(torch, lane_lo, lanes, f0, f1, layout, self.dev, R), torch, xv, lanes, frames, layout, R)
            self.y = torch.empty(lanes * frames, dtype=torch.float32, device=self.dev)
        elif self.family == "lockin":
            self.cfgs = _abi.LockinI32()
            self.cfgs.order, self.cfgs.cascade = 2, 2
            for c in range(2):
                self.cfgs.k[c][0], self.cfgs.k[c][1] = lockin_k()
            assert self.fn["lockin_state_words"](C.byref(self.cfgs)) == cfg["state_words"]
            self.x = torch.empty(lanes * frames, dtype=torch.int32, device=self.dev)
            xv = self.x.view(frames, lanes) if self.frame_major else self.x.view(lanes, frames)
            blockwise(lambda f0, f1: c4_input(torch, lane_lo, lanes, f0, f1, layout, self.dev), torch, xv, lanes, frames, layout)
            self.y = torch.empty(lanes * frames * 2, dtype=torch.int32, device=self.dev)
            self.state_init = {1: c4_steps(torch, lane_lo, lanes, self.dev)}  # Accu {state, step}: row 1 = step
        elif cfg["dtype"] == "i32":
            sos = (C.c_double * 6)(*lowpass_sos(F0))
            rec = _abi.BiquadI32()
            call("biquad_i32_from_sos", sos, FRAC, C.byref(rec))
            self.cfgs = (_abi.BiquadI32 * 1)(rec)
            self.x_host = c2_input_host(frames, lanes, layout, rank)
            self.x = torch.from_numpy(self.x_host).to(self.dev).reshape(-1)
            if rank != 0:
                self.x_host = None
        else:
            sos = (C.c_double * 6)(*lowpass_sos(F0))
            rec = _abi.BiquadF32()
            call("biquad_f32_from_sos_f64", sos, C.byref(rec))
            self.cfgs = (_abi.BiquadF32 * 1)(rec)
            self.x = torch.empty(lanes * frames, dtype=torch.float32, device=self.dev)
            xv = self.x.view(frames, lanes) if self.frame_major else self.x.view(lanes, frames)
            blockwise(lambda f0, f1: c5_input(torch, lane_lo, lanes, f0, f1, layout, self.dev), torch, xv, lanes, frames, layout)
        self.inplace = bool(inplace) and self.family == "biquad"
        self.x0 = None
        if self.inplace:
            # SplitInplace::inplace (dsp-process/src/process.rs:135-142): one buffer, y == x.  Every pass overwrites its input, so a
            # pristine copy restores it before each timed launch and before the integrity step (restore outside the event pair)
            self.y = self.x
            self.x0 = self.x.clone()
        elif self.family == "biquad":
            self.y = torch.empty_like(self.x)  # a plain second allocation: no placement tuning
        self.state = torch.zeros((cfg["state_words"], lanes), dtype=torch.int32, device=self.dev)
        self.reset_state()
        self.stream = torch.cuda.Stream(device=self.dev)
        self.entry = cfg["entry"]
        tail = (C.c_void_p(self.state.data_ptr()), C.c_void_p(self.x.data_ptr()), C.c_void_p(self.y.data_ptr()), lanes, frames,
                self.layout, C.c_void_p(self.stream.cuda_stream))
        if self.family == "biquad":
            self._args = (C.cast(self.cfgs, C.c_void_p), n_sections) + tail
        else:
            self._args = (C.byref(self.cfgs),) + tail
        self.sync()

    def reset_state(self):
        """The state at the start of a stream: zero, except the rows a configuration initialises (C4: the Accu steps)."""
        self.state.zero_()
        for row, val in (self.state_init or {}).items():
            self.state[row] = val

    def step(self):
        self.call(self.entry, *self._args)

    def restore(self):
        """In place only: x back to the pristine input, on the launch stream."""
        if self.x0 is not None:
            with self.torch.cuda.stream(self.stream):
                self.x.copy_(self.x0, non_blocking=True)

    def sync(self):
        self.stream.synchronize()
        self.torch.cuda.synchronize()

    def timed_steps(self, k: int):
        """The timed region: k back-to-back launches between ONE pair of HIP events recorded on the launch stream itself;
        returns a function that yields [average ms per launch] once the stream has been synchronised.  (Round 2 recorded an
        event pair around EVERY launch inside the timed region: the records cost 2.5-3.4 % of the step —
        tools/exp_bench_gap.py: 0.3555 ms per step with them, 0.3458 without, kernel median 0.3457 — so `value` read lower
        than the engine's back-to-back rate.  The per-launch figures now come from probe_steps(), outside the timed region.)"""
        if self.inplace:
            # y == x: every launch works on the restored input; one event pair per launch, the restores between the pairs
            ev = [(self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)) for _ in range(k)]
            for a, b in ev:
                self.restore()
                a.record(self.stream)
                self.step()
                b.record(self.stream)
            return lambda: [sum(a.elapsed_time(b) for a, b in ev) / max(k, 1)]
        a, b = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        a.record(self.stream)
        for _ in range(k):
            self.step()
        b.record(self.stream)
        return lambda: [a.elapsed_time(b) / max(k, 1)]

    def probe_steps(self, k: int):
        """k further launches (untimed), each between its own event pair: per-launch durations in ms."""
        ev = [(self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)) for _ in range(k)]
        for a, b in ev:
            self.restore()
            a.record(self.stream)
            self.step()
            b.record(self.stream)
        self.sync()
        return [a.elapsed_time(b) for a, b in ev]

    def copy_rate(self, k: int = 20):
        """GB/s (read + written bytes) of `idsp_device_copy` over this configuration's own x -> y footprint, same run, same
        buffers, same stream: what a plain copy reaches on this box ("also report vs measured copy-kernel bandwidth
        on the box").  y is clobbered: call before verify().  None in place (one buffer) or where y is not x's size."""
        if self.inplace or self.x.numel() * self.x.element_size() != self.y.numel() * self.y.element_size():
            return None
        nbytes = self.x.numel() * self.x.element_size()
        args = (C.c_void_p(self.y.data_ptr()), C.c_void_p(self.x.data_ptr()), nbytes, C.c_void_p(self.stream.cuda_stream))
        for _ in range(3):
            self.call("device_copy", *args)
        a, b = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        a.record(self.stream)
        for _ in range(k):
            self.call("device_copy", *args)
        b.record(self.stream)
        self.sync()
        return 2.0 * nbytes * k / (a.elapsed_time(b) * 1e-3) / 1e9

    def verify(self, sample_lanes: int = 0):
        """Zero the state, run one step, return [sum of y's words, sum of the state words] as wrapping i64 —
        a function of (configuration, input) only, whatever was timed before.  With `sample_lanes`, also the
        y sum over the first `sample_lanes` lanes (what a bounded CPU sample can be compared with)."""
        from idsp_amd.sharding import checksum_i64

        self.sync()
        self.reset_state()
        self.restore()
        self.sync()
        self.step()
        self.sync()
        out = [checksum_i64(self.y), checksum_i64(self.state)]
        if sample_lanes:
            yv = self.y.view(self.frames, self.lanes)[:, :sample_lanes] if self.frame_major else self.y.view(self.lanes, self.frames)[:sample_lanes]
            out.append(checksum_i64(yv.contiguous()))
        return out

    def kernel_name(self) -> str:
        return self.fn["last_kernel"]().decode(errors="replace")

    def reduce_device(self, backend: str):
        return self.dev if backend == "nccl" else "cpu"

    def free(self):
        self.x = self.y = self.state = self.x0 = None
        self.torch.cuda.empty_cache()


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
    # barrier, synchronize, THEN the clock; after the K steps synchronize, read the clock, THEN the closing barrier: the
    # interval holds this rank's K steps only (the MAX over the ranks is taken by the caller), not the rendezvous
    if dist:
        dist.barrier()
    engine.sync()
    t0 = time.perf_counter()
    durations = engine.timed_steps(steps)
    engine.sync()
    elapsed = time.perf_counter() - t0
    if dist:
        dist.barrier()
    engine.sync()
    return elapsed, durations(), done


def host_cpu_budget(affinity_cpus: int):
    """(threads worth using, CPU quota of this container in cores or None).  The GPU boxes expose all host cores to
    sched_getaffinity (256) but cap the container with a cgroup CPU quota (cpu.max = "1600000 100000" = 16 cores):
    threads beyond the quota are throttled, not run — 256 threads measured SLOWER than 16 (profiles/r03_cpu_scaling_probe.txt)."""
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2
            q, period = f.read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:  # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = float(f.read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    threads = affinity_cpus if quota is None else max(1, min(affinity_cpus, int(quota + 0.5)))
    return threads, quota


def host_cpu_model():
    """(model name, logical CPUs of the host) from /proc/cpuinfo — SURVEY 8(d): "report nproc, CPU model"."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, os.cpu_count()


def cpu_baseline(cfg_name: str, cfg: dict, x_host=None, layout: str = "frame", seconds_budget: float = 14.0):
    """Time the CPU oracle (kind "port") on the host cores of this box.

    C2: the FULL 65536 x 4096 tensor the GPU was timed on (`x_host`); C5: the first C5_CPU_LANES lanes of it.  The
    oracle's `idsp_ref_biquad_mt_reps` cuts the lanes into one contiguous block per thread (pinned POSIX
    threads inside the C library; every thread repeats its block `reps` times with the state carried, so
    thread start-up is paid once per measurement), in FRAME_MAJOR — the reference's `[X; N]` frames-outer
    loop, dsp-process/src/compose.rs:468-476 — and in LANE_MAJOR — its serial lane loop over contiguous
    slices, compose.rs:478-494 — with all allowed cores and with one thread.  The first pass (zero state)
    also yields the oracle's checksums of the tensor for the `integrity` comparison."""
    import numpy as np

    import oracle  # cpu_baseline leg only
    from idsp_amd import _abi

    try:
        lib = oracle.load(native=True)  # -march=native, built on this host
        flavour = "-O3 -march=native"
    except Exception:
        lib = oracle.load()
        flavour = "-O3"
    mt = lib.idsp_ref_biquad_mt_reps
    mt.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t,
                   C.c_int, C.c_int, C.c_int]
    mt.restype = C.c_int
    lib.idsp_ref_host_cpus.restype = C.c_int
    affinity_cpus = max(1, int(lib.idsp_ref_host_cpus()))
    cores, quota = host_cpu_budget(affinity_cpus)  # threads of the all-core legs = what the container may actually use
    sos = (C.c_double * 6)(*lowpass_sos(F0))
    if cfg["dtype"] == "i32":
        rec = _abi.BiquadI32()
        lib.idsp_ref_biquad_i32_from_sos(sos, FRAC, C.byref(rec))
    else:
        rec = _abi.BiquadF32()
        lib.idsp_ref_biquad_f32_from_sos_f64(sos, C.byref(rec))
    frames = cfg["frames"]
    if cfg_name == "c2":
        if x_host is None:
            x_host = c2_input_host(frames, cfg["lanes"], layout, 0)
        lanes = x_host.shape[1] if layout == "frame" else x_host.shape[0]
        frames = x_host.size // lanes
        native = np.ascontiguousarray(x_host)
    else:
        lanes = C5_CPU_LANES
        native = c5_input_host(0, lanes, frames, layout)
    kind, words = cfg["oracle_kind"], cfg["state_words"]
    n = lanes * frames
    # the same samples in the other layout (a transpose of the tensor, not a new draw)
    fm = native if layout == "frame" else np.ascontiguousarray(native.reshape(lanes, frames).T)
    lm = native if layout == "lane" else np.ascontiguousarray(native.reshape(frames, lanes).T)
    y = np.zeros(n, dtype=native.dtype)  # touched: no page faults inside the timed passes

    def run(x, lay, threads, reps):
        st = np.zeros((words, lanes), dtype=np.uint32)
        t0 = time.perf_counter()
        rc = mt(kind, C.byref(rec), 1, st.ctypes.data, x.ctypes.data, y.ctypes.data, lanes, frames, lay, threads, reps)
        dt = time.perf_counter() - t0
        assert rc == 0
        return dt, st

    def checksum(a):
        return wrap64(int(a.reshape(-1).view(np.int32).sum(dtype=np.int64)))

    def rate(x, lay, threads, budget):
        dt1, _ = run(x, lay, threads, 1)  # calibration (and warm caches / page tables)
        reps = max(1, min(4096, int(budget / max(dt1, 1e-4))))
        dt, _ = run(x, lay, threads, reps)
        return reps * n / dt / 1e6, reps

    # oracle checksums of the GPU's tensor in the GPU's layout (zero state, one pass)
    _, st0 = run(native, 0 if layout == "frame" else 1, cores, 1)
    oracle_sums = [checksum(y), checksum(st0)]
    leg = seconds_budget / 4.0
    res = {}
    for name, x, lay in (("frame", fm, 0), ("lane", lm, 1)):
        all_rate, all_reps = rate(x, lay, cores, leg)
        one_rate, one_reps = rate(x, lay, 1, leg)
        res[name] = {"all_cores": round(all_rate, 1), "one_thread": round(one_rate, 1),
                     "parallel_efficiency": round(all_rate / (one_rate * cores), 3), "passes": [all_reps, one_reps]}
    best_layout = max(res, key=lambda k: res[k]["all_cores"])
    best = res[best_layout]
    cpu_model, nproc = host_cpu_model()
    return {
        "value": best["all_cores"], "unit": "Msamples/s", "cores": cores, "kind": "port",
        "cpu_model": cpu_model, "nproc": nproc,
        "host_cpus": affinity_cpus, "cgroup_cpu_quota": quota,
        "single_thread_value": max(res[k]["one_thread"] for k in res),
        "layout_of_value": best_layout, "by_layout": res,
        "parallel_efficiency": best["parallel_efficiency"],
        "dram_gbs_at_value": round(best["all_cores"] * cfg["bytes_per_sample"] / 1e3, 1),
        "note": "`cores` = threads of the all-core legs = min(CPUs in the affinity mask, cgroup CPU quota): threads beyond the quota are "
                "throttled.  parallel_efficiency = all-core rate / (one-thread rate x cores); dram_gbs_at_value = 8 B per sample at "
                "that rate (x and y are first-touched by one thread)",
        "sample": f"{lanes} of {cfg['lanes']} lanes x {frames} samples ({'the full tensor the GPU ran' if lanes == cfg['lanes'] else 'a lane prefix of it'}), "
                  f"C oracle {flavour}, pinned pthreads over contiguous lane blocks, ~{seconds_budget:.0f} s in 4 legs; "
                  "reference is Rust (no toolchain here)",
        "oracle_checksums": oracle_sums, "oracle_checksum_lanes": lanes,
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


def committed_issue(config: str, kernel: str):
    """The second roof of the kernels that are not memory-bound (C3, C4), from the committed SQ / TCC passes of this same command
    (profiles/bench_<config>_traffic.json, key "issue", written by tools/make_traffic_json.py): how busy the busiest issue unit
    is — SQ_ACTIVE_INST_{VALU, LDS} x 4 / (SIMDs or CUs) / kernel cycles — and the clock the L2 ran at (TCC_BUSY / duration: these
    launches are power-bound in clock, profiles/NOTES.md round 6).  From file, like `traffic`; None when the kernel differs."""
    try:
        with open(os.path.join(ROOT, "profiles", f"bench_{config}_traffic.json")) as f:
            tj = json.load(f)
        if tj.get("kernel_prefix") and not kernel.startswith(tj["kernel_prefix"]):
            return None
        return tj.get("issue")
    except (OSError, KeyError, ValueError):
        return None


def report(cfg_name, cfg, args, world, lanes_rank, frames, elapsed, kern_ms, untimed, kernel, total_lanes, inplace=False):
    samples_all = total_lanes * frames * cfg.get("samples_per_frame", 1)  # (input) samples per step over all ranks
    alg_bytes = algorithmic_bytes(cfg, lanes_rank, frames)
    med = statistics.median(kern_ms) if kern_ms else 0.0
    achieved = alg_bytes / (med * 1e-3) / 1e9 if med > 0 else 0.0
    file_key = cfg_name + ("_inplace" if inplace else "" if args.layout == "frame" else "_lane")
    traffic, traffic_src = committed_traffic(file_key, kernel) if not args.lanes else (None, None)
    issue = committed_issue(file_key, kernel) if not args.lanes else None
    ms_step = elapsed / max(args.steps, 1) * 1e3
    line = {
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
            "buffers": ("in place, y == x (SplitInplace::inplace); x restored from a pristine copy before every launch, outside the launch's event pair"
                        if inplace else "two separate plain allocations"),
            "untimed_steps": untimed, "settle_ms": args.settle_ms,
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "kernel": kernel, "kernel_ms": round(med, 4),
            "kernel_ms_stat": "average launch duration: one HIP-event pair on the launch stream around the K timed launches, / K (rank 0)",
            # the same bytes over the wall-clock step time (`ms_per_step`, what `value` is built from): launch gaps included
            "frac_of_ms_per_step": round(alg_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_step > 0 else None,
            "algorithmic_bytes": alg_bytes,
        },
    }
    if issue:
        line["roofline"]["issue"] = issue
    bd = by_direction(cfg_name, cfg, lanes_rank, frames, med)
    if bd:
        line["roofline"]["by_direction"] = bd
    return line


def summary_of(line: dict, head: str) -> dict:
    """{config: [wall-clock ms per step, its roofline fraction, HIP-event ms per launch, its roofline fraction (= roofline.frac),
    integrity match]} for the head line and every sub-object — every fraction beside the time it was computed from; compact and the
    LAST key of the line."""
    def entry(o):
        integ = o.get("integrity") or {}
        r = o["roofline"]
        return [o["ms_per_step"], r.get("frac_of_ms_per_step"), r["kernel_ms"], r["frac"], integ.get("match")]

    out = {head: entry(line)}
    if isinstance(line.get("perf_guard"), dict):
        out["perf_guard_ok"] = line["perf_guard"].get("ok")
    for name in ("c3", "c4", "c5"):
        if name in line:
            out[name] = entry(line[name])
    for name, o in (line.get("lane_major") or {}).items():
        out[name + "_lm"] = entry(o)
    for name, o in (line.get("inplace") or {}).items():
        out[name + "_inplace"] = entry(o)
    return out


# ---------------------------------------------------------------------------------------- ranks

def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_command(n: int, argv, script: str | None = None):
    """The launcher line of the bench contract, on a free local port."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script or os.path.abspath(__file__), *argv]


def spawn(n: int, argv, script: str | None = None, **popen_kw) -> int:
    """Start n ranks of `script` (default: this file) and wait for them; rank 0 prints the JSON line.  The ranks' stdout is
    filtered: JSON lines go to stdout, anything else a library prints there (gloo's connection banner) goes to stderr, so that
    stdout of `bench.py --gpus N` is the ONE line of the contract."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    if popen_kw:
        return subprocess.call(spawn_command(n, argv, script), env=env, **popen_kw)
    proc = subprocess.Popen(spawn_command(n, argv, script), env=env, stdout=subprocess.PIPE, text=True, errors="replace")
    for ln in proc.stdout:
        (sys.stdout if ln.startswith("{") else sys.stderr).write(ln)
        sys.stdout.flush()
    return proc.wait()


def gather_ints(values, dist, device):
    """[[values of rank 0], [values of rank 1], …] (signed 64-bit integers)."""
    import torch

    t = torch.tensor([wrap64(int(v)) for v in values], dtype=torch.int64, device=device)
    if not dist:
        return [t.tolist()]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def run_config(cfg_name, args, engine_factory, dist, rank, world, local, backend, steps, warmup, settle_ms, lanes_override,
               frames_override, inplace=False):
    """One configuration on this rank: build the engine, the timed region, the reductions and the integrity
    step.  Returns (line or None on ranks > 0, engine) — the caller frees the engine."""
    import torch

    from idsp_amd.sharding import allreduce_checksum

    cfg = CONFIGS[cfg_name]
    frames = frames_override or cfg["frames"]
    lane_lo, lanes_rank = job_shard(cfg, rank, world, lanes_override or None)
    total_lanes = (lanes_override or cfg["lanes"]) * (world if cfg["scaling"] == "weak" else 1)  # replicas count as lanes of their own
    if inplace:
        engine = engine_factory(cfg_name, cfg, lane_lo, lanes_rank, frames, args.layout, rank, local, inplace=True)
    else:
        engine = engine_factory(cfg_name, cfg, lane_lo, lanes_rank, frames, args.layout, rank, local)
    rdev = engine.reduce_device(backend)
    elapsed, kern_ms, untimed = run_timed(engine, steps, warmup, settle_ms, dist)
    if inplace and kern_ms:
        # y == x: the wall-clock interval holds the K restores of x as well; the step time of an in-place pass is the sum of the
        # K launches' own event-pair durations (timed_steps), which is what `value` and `ms_per_step` are built from here
        elapsed = kern_ms[0] * steps * 1e-3
    per_launch = engine.probe_steps(min(steps, 50)) if hasattr(engine, "probe_steps") else list(kern_ms)  # outside the timed region
    t = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
    if dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed_max = float(t.item())
    med_us = int(round(statistics.median(per_launch) * 1e3)) if per_launch else 0
    copy_gbs = engine.copy_rate() if (rank == 0 and hasattr(engine, "copy_rate")) else None
    sample = C5_CPU_LANES if cfg_name == "c5" and world == 1 and lanes_rank > C5_CPU_LANES else 0
    sums = engine.verify(sample)
    y_sum_all = allreduce_checksum(sums[0], device=rdev)  # the 8-byte checksum all-reduce
    per_rank = gather_ints([sums[0], sums[1], med_us, lane_lo, lanes_rank], dist, rdev)
    if rank != 0:
        return None, engine
    a = argparse.Namespace(**{**vars(args), "steps": steps, "warmup": warmup, "settle_ms": settle_ms,
                              "lanes": lanes_override, "frames": frames_override})
    line = report(cfg_name, cfg, a, world, lanes_rank, frames, elapsed_max, kern_ms, untimed, engine.kernel_name(), total_lanes, inplace)
    if copy_gbs:
        line["roofline"]["copy_gbs"] = round(copy_gbs, 1)
        line["roofline"]["copy_note"] = ("idsp_device_copy (16 B per thread, nontemporal, a chunk per workgroup) x -> y of this configuration, 20 launches "
                                         "between one event pair, same run / buffers / stream; read + written bytes")
        line["roofline"]["frac_of_copy"] = round(line["roofline"]["achieved"] / copy_gbs, 4)
    if per_launch:
        line["roofline"]["per_launch_ms"] = {"median": round(statistics.median(per_launch), 4), "min": round(min(per_launch), 4),
                                             "launches": len(per_launch), "note": "separate event pair per launch, after the timed region"}
    overridden = bool(lanes_override or frames_override)
    exp = [expected_for(cfg_name, args.layout, r, pr[3], pr[4], overridden) for r, pr in enumerate(per_rank)]
    match = [None if e is None else (e[0] == pr[0] and e[1] == pr[1]) for e, pr in zip(exp, per_rank)]
    assert wrap64(sum(pr[0] for pr in per_rank)) == y_sum_all, "checksum all-reduce disagrees with the gathered sums"
    line["ranks"] = {
        "kernel_ms_median": [round(pr[2] / 1e3, 4) for pr in per_rank],
        "lanes": [pr[4] for pr in per_rank], "first_lane": [pr[3] for pr in per_rank],
    }
    line["integrity"] = {
        "method": "state zeroed, one more step, wrapping i64 sums of the 32-bit words of y and of the written-back state; "
                  "y sums all-reduced (SUM) over the ranks",
        "checksum_y_allreduce": y_sum_all, "checksum_state_sum": wrap64(sum(pr[1] for pr in per_rank)),
        "per_rank": [[pr[0], pr[1]] for pr in per_rank],
        "expected_source": "tests/golden/bench_checksums.json (CPU oracle on the same inputs)" if any(e is not None for e in exp) else None,
        "per_rank_match": match,
        "match": (all(match) if all(m is not None for m in match) else None),
    }
    if sample:
        line["integrity"]["sample_lanes"] = sample
        line["integrity"]["sample_checksum_y"] = sums[2]
    return line, engine


def rank_main(args, engine_factory=HipEngine):
    """One rank of the bench (the only rank when WORLD_SIZE is unset)."""
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local, ndev = engine_factory.device_setup(local)
    shared = world > ndev
    # RCCL ("nccl" on ROCm) is the backend, one rank per GPU.  With fewer GPUs than ranks (a 1-GPU box) two
    # ranks cannot join one RCCL communicator from the same device: the ranks then share devices and meet over
    # gloo, which exercises the control flow only.  IDSP_BENCH_BACKEND overrides the choice.
    backend = os.environ.get("IDSP_BENCH_BACKEND") or ("gloo" if shared else "nccl")
    dist = None
    rccl_ranks = 1
    # IDSP_BENCH_FORCE_DIST=1: join a process group even as the only rank (under torchrun with one process) — the way to run the
    # RCCL code path (init with device_id, device-tensor reductions, barriers) on a one-GPU box
    if world > 1 or os.environ.get("IDSP_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        one = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", local) if backend == "nccl" else "cpu")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        rccl_ranks = int(one.item())
        assert rccl_ranks == dist.get_world_size() == world

    line, engine = run_config(args.config, args, engine_factory, dist, rank, world, local, backend, args.steps, args.warmup,
                              args.settle_ms, args.lanes, args.frames, inplace=bool(getattr(args, "inplace", False)))
    x_host = getattr(engine, "x_host", None)
    engine.free()
    sub = None
    if args.config == "c2" and not args.no_c5 and (args.c5_lanes or not (args.lanes or args.frames)):
        # the strong-scaling job beside the weak-scaling headline: 2^20 / N lanes per rank
        sub, e5 = run_config("c5", args, engine_factory, dist, rank, world, local, backend, min(args.steps, 20),
                             args.warmup, min(args.settle_ms, 100.0), args.c5_lanes, args.c5_frames)
        e5.free()
    subs = {}
    if args.config == "c2" and not (args.lanes or args.frames):
        # the other single-GPU configurations of BASELINE.json beside the headline: own timed region, roofline and integrity
        families = getattr(engine_factory, "FAMILIES", ("biquad",))
        for name, skip in (("c3", args.no_c3), ("c4", args.no_c4)):
            if skip or CONFIGS[name]["family"] not in families:
                continue
            subs[name], e = run_config(name, args, engine_factory, dist, rank, world, local, backend, min(args.steps, 20),
                                       args.warmup, min(args.settle_ms, 100.0), 0, 0)
            e.free()
    lm_subs = {}
    if args.config == "c2" and args.layout == "frame" and not getattr(args, "no_lane_major", False) and not (args.lanes or args.frames):
        # LaneMajor (`[lane][frame]`, the reference's native lane view, dsp-process/src/view.rs:176-196) of C2, C3 and C4: the same
        # tensors transposed, own timed region, roofline and integrity each
        families = getattr(engine_factory, "FAMILIES", ("biquad",))
        lm_args = argparse.Namespace(**{**vars(args), "layout": "lane"})
        for name, skip in (("c2", False), ("c3", args.no_c3), ("c4", args.no_c4)):
            if skip or CONFIGS[name].get("family", "biquad") not in families:
                continue
            lm_subs[name], e = run_config(name, lm_args, engine_factory, dist, rank, world, local, backend, min(args.steps, 20),
                                          args.warmup, min(args.settle_ms, 100.0), 0, 0)
            e.free()
    ip_subs = {}
    if (args.config == "c2" and args.layout == "frame" and not getattr(args, "no_inplace", False) and getattr(engine_factory, "INPLACE", False)
            and not (args.lanes or args.frames)):
        # the reference's own benchmarked mode: SplitInplace::inplace (dsp-process/src/process.rs:135-142; "slice inplace",
        # tests/embedded/README.md) — C2 and (unless --no-c5) C5 with y == x, same inputs, so the checksums on file hold
        for name, skip in (("c2", False), ("c5", args.no_c5)):
            if skip:
                continue
            ip_subs[name], e = run_config(name, args, engine_factory, dist, rank, world, local, backend, min(args.steps, 20),
                                          args.warmup, min(args.settle_ms, 100.0), 0, 0, inplace=True)
            e.free()
    if rank == 0:
        # ranks that met in the process group's SUM all-reduce; `rccl_ranks` only when that group IS an RCCL communicator
        line["group_ranks"] = rccl_ranks
        line["rccl_ranks"] = rccl_ranks if (backend == "nccl" and dist) else None
        line["backend"] = "nccl (RCCL)" if backend == "nccl" else backend
        line["ranks_share_device"] = shared
        line["launcher"] = os.environ.get("IDSP_BENCH_LAUNCHER", "torchrun (external)" if world > 1 else "single process")
        if world != args.gpus:
            line["config"]["note"] = f"--gpus {args.gpus} but WORLD_SIZE={world}: the process group's size is what ran"
        if sub is not None:
            keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "config",
                    "roofline", "ranks", "integrity")
            line["c5"] = {k: sub[k] for k in keep}
        for name, sl in subs.items():
            keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "config",
                    "roofline", "ranks", "integrity")
            line[name] = {k: sl[k] for k in keep}
        if lm_subs:
            keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "integrity")
            line["lane_major"] = {name: {k: sl[k] for k in keep} for name, sl in lm_subs.items()}
        if ip_subs:
            keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "integrity")
            line["inplace"] = {name: {k: sl[k] for k in keep} for name, sl in ip_subs.items()}
        if world == 1 and not args.no_cpu and CONFIGS[args.config].get("family", "biquad") == "biquad":
            cb = cpu_baseline(args.config, CONFIGS[args.config], x_host, args.layout)
            integ = line["integrity"]
            if cb["oracle_checksum_lanes"] == line["config"]["lanes_per_gpu"]:
                integ["oracle_live"] = cb["oracle_checksums"]
                integ["oracle_live_match"] = cb["oracle_checksums"] == integ["per_rank"][0]
            elif integ.get("sample_lanes") == cb["oracle_checksum_lanes"]:
                integ["oracle_live_sample_y"] = cb["oracle_checksums"][0]
                integ["oracle_live_match"] = cb["oracle_checksums"][0] == integ["sample_checksum_y"]
            line["cpu_baseline"] = cb
        else:
            line["cpu_baseline"] = None
        if world == 1 and args.config == "c2" and not getattr(args, "no_guard", False) and not (args.lanes or args.frames) and getattr(engine_factory, "GUARD", False):
            # performance guard (tools/perf_guard.py): the shapes whose rate rides on the paced sweep schedule, against committed floors
            from tools import perf_guard

            have = {"c2": line["roofline"]["frac"]}
            if "c5" in line:
                have["c5"] = line["c5"]["roofline"]["frac"]
            if "c2" in (line.get("inplace") or {}):
                have["c2_inplace"] = line["inplace"]["c2"]["roofline"]["frac"]
            try:
                line["perf_guard"] = perf_guard.run(have)
            except Exception as e:  # noqa: BLE001  (the guard must never cost the line)
                line["perf_guard"] = {"ok": None, "error": repr(e)}
        head_copy = line["roofline"].get("copy_gbs")
        if head_copy:
            # every other configuration against the SAME yardstick: the plain copy of the head line's footprint, measured in this run
            for o in [line.get(k) for k in ("c3", "c4", "c5")] + list((line.get("lane_major") or {}).values()) + list((line.get("inplace") or {}).values()):
                if o and "frac_of_copy" not in o["roofline"]:
                    o["roofline"]["frac_of_head_copy"] = round(o["roofline"]["achieved"] / head_copy, 4)
        line["summary"] = summary_of(line, args.config)  # LAST key: the stored tail of a long line still carries every config
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def parse_args(argv=None):
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
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 strong-scaling sub-object of the default run")
    ap.add_argument("--no-lane-major", action="store_true", help="skip the LaneMajor sub-objects (C2, C3, C4) of the default run")
    ap.add_argument("--inplace", action="store_true", help="run the configuration itself in place (y == x; profiling runs of the in-place mode)")
    ap.add_argument("--no-inplace", action="store_true", help="skip the in-place (y == x) sub-objects (C2, C5) of the default run")
    ap.add_argument("--no-guard", action="store_true", help="skip the perf_guard object (tools/perf_guard.py) of the default run")
    ap.add_argument("--no-c3", action="store_true", help="skip the C3 (HbfDec /16) sub-object of the default run")
    ap.add_argument("--no-c4", action="store_true", help="skip the C4 (lock-in) sub-object of the default run")
    ap.add_argument("--c5-lanes", type=int, default=0, help="total lanes of the C5 sub-object (diagnostics; default 2^20)")
    ap.add_argument("--c5-frames", type=int, default=0, help="samples per lane of the C5 sub-object (diagnostics; default 4096)")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: start the N ranks ourselves, exactly as the contract's torchrun line would
        os.environ["IDSP_BENCH_LAUNCHER"] = "self-spawned torch.distributed.run"
        raise SystemExit(spawn(args.gpus, sys.argv[1:] if argv is None else list(argv)))
    rank_main(args)


if __name__ == "__main__":
    main()
