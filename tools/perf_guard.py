#!/usr/bin/env python3
"""Performance guard (GPU): the shapes whose rate rides on a schedule tuned at the memory ceiling — the paced dense-sweep kernel
(idsp_amd/csrc/fm_sweep.h: the same `s_sleep 4` measures 0.754 behind a run-time switch and 0.800 as a compile-time constant) —
timed through the C ABI against committed floors (profiles/perf_guard_floors.json, fractions of the 8 TB/s HBM peak), with the
toolchain and firmware strings of the box recorded beside them.  A compiler or firmware update that moves the timing edge shows
up as `ok: false` in bench.py's line (`perf_guard`) instead of as a silent 5 %.

    python tools/perf_guard.py            # all shapes, 30 launches each, one JSON object
bench.py passes the fractions it has already measured (C2, C2 in place, C5) and this module times the rest.

Shapes (FrameMajor x 4096 frames): C2 = i32 DF1 65536 lanes; C2 in place; C5 = f32 DF2T 2^20 lanes; C5's 8-GPU shard = f32 DF2T
131072 lanes; i32 DF1 at 131072 / 100000 / 32768 lanes (full blocks two per workgroup; narrow blocks; several frames per segment) and at
16384 lanes (the compute + mover pair kernel).
Reference loop nest replaced: dsp-process/src/compose.rs:468-494 over process.rs:122-141."""
from __future__ import annotations

import ctypes as C
import glob
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK = 8.0e12
FLOORS_PATH = os.path.join(ROOT, "profiles", "perf_guard_floors.json")
# name -> (entry, dtype name, state words, lanes, frames, in place)
SHAPES = {
    "c2": ("biquad_i32_df1", "int32", 4, 65536, 4096, False),
    "c2_inplace": ("biquad_i32_df1", "int32", 4, 65536, 4096, True),
    "c5": ("biquad_f32_df2t", "float32", 2, 1 << 20, 4096, False),
    "c5_shard8": ("biquad_f32_df2t", "float32", 2, 131072, 4096, False),
    "i32_131072": ("biquad_i32_df1", "int32", 4, 131072, 4096, False),
    "i32_100000": ("biquad_i32_df1", "int32", 4, 100000, 4096, False),
    "i32_32768": ("biquad_i32_df1", "int32", 4, 32768, 4096, False),
    "i32_16384": ("biquad_i32_df1", "int32", 4, 16384, 4096, False),  # the compute + mover pair kernel (lane_stream.h)
}


def lowpass_sos(f0: float):
    w0 = math.pi * f0
    fcos, fsin = math.cos(w0), math.sin(w0)
    alpha = 0.5 * fsin * math.sqrt(2.0)
    b = 0.5 * (1.0 - fcos)
    return [b, 2.0 * b, b, 1.0 + alpha, -2.0 * fcos, 1.0 - alpha]


def toolchain():
    """What the timing edge depends on besides the silicon: compiler, runtime, kernel driver, firmware."""
    out = {}
    try:
        txt = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, timeout=20).stdout
        out["hipcc"] = " | ".join(line.strip() for line in txt.splitlines()[:2])
    except (OSError, subprocess.SubprocessError):
        out["hipcc"] = None
    try:
        import torch

        out["torch_hip"] = torch.version.hip
    except Exception:  # noqa: BLE001
        out["torch_hip"] = None
    try:
        with open("/sys/module/amdgpu/version") as f:
            out["amdgpu_driver"] = f.read().strip()
    except OSError:
        out["amdgpu_driver"] = None
    fw = {}
    for path in sorted(glob.glob("/sys/class/drm/card*/device/fw_version/*"))[:64]:
        name = os.path.basename(path)
        if name.split("_fw_version")[0] in ("mec", "mec2", "sdma", "smc", "rlc", "imu", "mes", "psp_sos") and name not in fw:
            try:
                with open(path) as f:
                    fw[name] = f.read().strip()
            except OSError:
                pass
    out["firmware"] = fw or None
    return out


def measure(name: str, launches: int = 30):
    """fraction of the HBM peak of one shape: algorithmic bytes (8 B per sample + 2 x state) / average launch time between ONE event pair"""
    import torch

    from idsp_amd import _abi
    from idsp_amd._lib import call

    entry, dt, words, lanes, frames, inplace = SHAPES[name]
    dev = torch.device("cuda", torch.cuda.current_device())
    sos = (C.c_double * 6)(*lowpass_sos(0.01))
    if dt == "int32":
        rec = _abi.BiquadI32()
        call("biquad_i32_from_sos", sos, 30, C.byref(rec))
        x = torch.randint(-(1 << 24), 1 << 24, (lanes * frames,), dtype=torch.int32, device=dev)
    else:
        rec = _abi.BiquadF32()
        call("biquad_f32_from_sos_f64", sos, C.byref(rec))
        x = torch.randn(lanes * frames, dtype=torch.float32, device=dev)
    y = x if inplace else torch.empty_like(x)
    st = torch.zeros((words, lanes), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    args = (C.byref(rec), 1, C.c_void_p(st.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), lanes, frames, _abi.FRAME_MAJOR,
            C.c_void_p(stream.cuda_stream))
    import time

    t0 = time.perf_counter()
    while True:  # warm up by time, not by count: the first launches after an idle gap run at a lower clock (bench.py settles for 250 ms too)
        for _ in range(8):
            call(entry, *args)
        torch.cuda.synchronize()
        if time.perf_counter() - t0 > 0.25:
            break
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(launches):
        call(entry, *args)
    b.record(stream)
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / launches
    alg = 8 * lanes * frames + 2 * 4 * words * lanes
    del x, y, st
    torch.cuda.empty_cache()
    return round(alg / (ms * 1e-3) / HBM_PEAK, 4)


def run(measured: dict | None = None, launches: int = 30):
    """{"ok", "worst": [shape, fraction, floor], "fractions", "floors", "toolchain", "floors_toolchain"}; `measured`: fractions the caller
    already has (same definition) — the rest are timed here."""
    with open(FLOORS_PATH) as f:
        ref = json.load(f)
    floors = ref["floors"]
    fr = dict(measured or {})
    for name in SHAPES:
        if name in floors and name not in fr:
            fr[name] = measure(name, launches)
    margin = {k: round(fr[k] - floors[k], 4) for k in floors if k in fr}
    worst = min(margin, key=margin.get)
    tc = toolchain()
    return {
        "ok": all(v >= 0 for v in margin.values()), "worst": [worst, fr[worst], floors[worst]],
        "fractions": {k: fr[k] for k in floors if k in fr}, "floors": floors,
        "toolchain": tc, "toolchain_of_floors": ref.get("toolchain"),
        "toolchain_changed": (ref.get("toolchain") or {}).get("hipcc") != tc.get("hipcc") or (ref.get("toolchain") or {}).get("firmware") != tc.get("firmware"),
        "note": "fractions of the 8 TB/s HBM peak, 30 launches between one event pair through the C ABI; floors = profiles/perf_guard_floors.json "
                "(set below the slowest of the boxes seen: box-to-box spread is 3-5 %); ok = every shape at or above its floor",
    }


if __name__ == "__main__":
    print(json.dumps(run()))
