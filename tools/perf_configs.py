#!/usr/bin/env python3
"""Per-kernel roofline survey over the BASELINE.json configs (C2..C5 shapes) on
one GPU.  Development tool (bench.py is the contract benchmark): prints one
JSON line per kernel with algorithmic GB/s against the 8 TB/s HBM peak.

  python tools/perf_configs.py [--only c2,c3,...] [--iters 10]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from idsp_amd import _abi  # noqa: E402
from idsp_amd._lib import call  # noqa: E402

PEAK = 8000.0
dev = torch.device("cuda:0")


def lowpass_sos(f0):
    w0 = math.tau * f0
    fsin, fcos = math.sin(w0), math.cos(w0)
    alpha = 0.5 * fsin * math.sqrt(2.0)
    b = 0.5 * (1.0 - fcos)
    return [b, 2.0 * b, b, 1.0 + alpha, -2.0 * fcos, 1.0 - alpha]


def timeit(fn, iters, warm=3, warm_ms=30.0):
    stream = torch.cuda.current_stream()
    # The first launches after an idle gap (allocation, input generation) run at a lower clock: the first line of a
    # process read 0.49 ms where the steady state is 0.37 ms.  Warm up by time, not by count.
    import time
    t0 = time.perf_counter()
    n = 0
    while n < warm or (time.perf_counter() - t0) * 1e3 < warm_ms:
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(stream)
        fn()
        b.record(stream)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def report(name, units, unit_name, alg_bytes, med, mn, **extra):
    print(json.dumps({
        "kernel": name, "ms_median": round(med, 4), "ms_min": round(mn, 4),
        f"G{unit_name}/s": round(units / (med * 1e-3) / 1e9, 1),
        "GB/s": round(alg_bytes / (med * 1e-3) / 1e9, 1), "frac_hbm_peak": round(alg_bytes / (med * 1e-3) / 1e9 / PEAK, 4),
        **extra}), flush=True)


def p(t):
    return C.c_void_p(t.data_ptr())


def out_like(x):
    """Output buffer of x's shape: a plain second allocation, like bench.py's.  IDSP_PERF_ARENA=1 carves x and y out of one
    allocation with y 48 KiB past x's size instead (round 1's placement: the offset y - x moved a ring-of-8 streaming
    kernel by up to 15 %).  Returns (x, y)."""
    if not os.environ.get("IDSP_PERF_ARENA"):
        return x, torch.empty_like(x)
    n, pad = x.numel(), (48 << 10) // x.element_size()
    arena = torch.empty(2 * n + pad, dtype=x.dtype, device=x.device)
    arena[:n].copy_(x)
    return arena[:n], arena[n + pad:]


def sptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def biquad(op, dtype, words, lanes, frames, layout, n_sections, iters, tag):
    if dtype == torch.int32:
        q = _abi.BiquadI32()
        call("biquad_i32_from_sos", (C.c_double * 6)(*lowpass_sos(0.01)), 30, C.byref(q))
        if "clamp" in op:
            cfg = (_abi.BiquadClampI32 * n_sections)()
            for c in cfg:
                c.ba[:] = list(q.ba)
                c.frac, c.u, c.min, c.max = 30, 3, -(1 << 30), 1 << 30
        else:
            cfg = (_abi.BiquadI32 * n_sections)(*([q] * n_sections))
        x = torch.randint(-(1 << 24), 1 << 24, (lanes * frames,), dtype=torch.int32, device=dev)
    else:
        q = _abi.BiquadF32()
        call("biquad_f32_from_sos_f64", (C.c_double * 6)(*lowpass_sos(0.01)), C.byref(q))
        if "clamp" in op:
            cfg = (_abi.BiquadClampF32 * n_sections)()
            for c in cfg:
                c.ba[:] = list(q.ba)
                c.u, c.min, c.max = 0.01, -10.0, 10.0
        else:
            cfg = (_abi.BiquadF32 * n_sections)(*([q] * n_sections))
        x = torch.randn(lanes * frames, dtype=torch.float32, device=dev)
    x, y = out_like(x)
    st = torch.zeros((words * n_sections if "cascade" not in op else 2 + 2 * n_sections, lanes), dtype=torch.int32, device=dev)

    def run():
        call(op, C.cast(cfg, C.c_void_p), n_sections, p(st), p(x), p(y), lanes, frames, layout, sptr())

    med, mn = timeit(run, iters)
    # rate in section-samples is what compares across kernels; "GB/s" prices ONE pass of 8 B/sample per launch of up to 8 sections
    passes = 1 if "cascade" in op else (n_sections + 7) // 8
    report(f"{tag}:{op} x{n_sections} {'LM' if layout else 'FM'} {lanes}x{frames}", lanes * frames, "sample",
           8 * lanes * frames * passes, med, mn)


_hbf_lib = None


def hbf_call(name, *args):
    """idsp_hbf_{dec,int}_f32 from IDSP_HBF_LIB if set (a small library holding only the half-band objects: timing
    variants of tools/exp_hbf_ring.sh), else from the engine."""
    global _hbf_lib
    path = os.environ.get("IDSP_HBF_LIB")
    if not path:
        return call(name, *args)
    if _hbf_lib is None:
        _hbf_lib = C.CDLL(path)
    fn = getattr(_hbf_lib, "idsp_" + name)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
    rc = fn(*args)
    assert rc == 0, rc
    return rc


def hbf(kind, stages, lanes, frames_low, layout, iters, tag):
    cfg = _abi.HbfCascadeF32()
    call(f"hbf_{kind}_cascade", 0, stages, C.byref(cfg))
    R = 1 << stages
    words = call(f"hbf_{kind}_state_words", C.byref(cfg))
    hi = torch.randn(lanes * frames_low * R, dtype=torch.float32, device=dev)
    if os.environ.get("IDSP_PERF_ZERO"):  # diagnostic only: all-zero input draws less power (the clock it buys is not a valid result)
        hi.zero_()
    lo = torch.randn(lanes * frames_low, dtype=torch.float32, device=dev)
    st = torch.zeros((words, lanes), dtype=torch.int32, device=dev)
    x, y = (hi, lo) if kind == "dec" else (lo, hi)

    def run():
        hbf_call(f"hbf_{kind}_f32", C.byref(cfg), p(st), p(x), p(y), lanes, frames_low, layout, sptr())

    med, mn = timeit(run, iters)
    n_hi = lanes * frames_low * R
    report(f"{tag}:hbf_{kind} /{R} {'LM' if layout else 'FM'} {lanes}x{frames_low * R}", n_hi, "hi-rate-sample",
           4 * n_hi + 4 * lanes * frames_low, med, mn)


def lockin(order, cascade, lanes, frames, layout, iters, tag, out="iq"):
    """out: "iq" (Complex<i32>), "arg" / "norm_sqr" (polar read-out fused), "iq+atan2" (two passes, for comparison)"""
    cfg = _abi.LockinI32()
    cfg.order, cfg.cascade = order, cascade
    k = math.pi * (1 << 31) * 1e-3  # f0 = 1e-3 fn (src/lowpass.rs:31-38)
    for c in range(cascade):
        if order == 1:
            cfg.k[c][0] = int(k)
        else:
            cfg.k[c][0], cfg.k[c][1] = int(k * k / (1 << 32)), -int(k * math.sqrt(2.0))
    words = call("lockin_state_words", C.byref(cfg))
    x = torch.randint(-(1 << 28), 1 << 28, (lanes * frames,), dtype=torch.int32, device=dev)
    y = torch.empty(lanes * frames * 2, dtype=torch.int32, device=dev)
    st = torch.zeros((words, lanes), dtype=torch.int32, device=dev)
    st[1] = torch.randint(-(1 << 31), (1 << 31) - 1, (lanes,), dtype=torch.int64, device=dev).to(torch.int32)

    a = torch.empty(lanes * frames, dtype=torch.int32, device=dev)
    entry = {"iq": "lockin_i32_process", "iq+atan2": "lockin_i32_process", "arg": "lockin_i32_arg", "norm_sqr": "lockin_i32_norm_sqr"}[out]
    # algorithmic bytes per sample: 4 in + 8 (Complex<i32> or i64) or 4 (arg) out; the two-pass form is
    # priced at the fused form's bytes so that the GB/s figures compare the same job
    nbytes = {"iq": 12, "norm_sqr": 12, "arg": 8, "iq+atan2": 8}[out]

    def run():
        call(entry, C.byref(cfg), p(st), p(x), p(y), lanes, frames, layout, sptr())
        if out == "iq+atan2":
            call("atan2_i32", p(y), p(a), lanes * frames, sptr())

    med, mn = timeit(run, iters)
    report(f"{tag}:lockin{'' if out == 'iq' else '->' + out} Lowpass<{order}>x{cascade} {'LM' if layout else 'FM'} {lanes}x{frames}",
           lanes * frames, "sample", nbytes * lanes * frames, med, mn)


def dds(lanes, frames, layout, iters, tag):
    st = torch.zeros((2, lanes), dtype=torch.int32, device=dev)
    st[1] = torch.randint(-(1 << 31), (1 << 31) - 1, (lanes,), dtype=torch.int64, device=dev).to(torch.int32)
    y = torch.empty(lanes * frames * 2, dtype=torch.int32, device=dev)

    def run():
        call("dds_i32", p(st), p(y), lanes, frames, layout, sptr())

    med, mn = timeit(run, iters)
    report(f"{tag}:dds {'LM' if layout else 'FM'} {lanes}x{frames}", lanes * frames, "sample", 8 * lanes * frames, med, mn)


def biquad_bylane(op, dtype, words, cv, lanes, frames, layout, n_sections, iters, tag):
    """`ByLane<[Biquad; lanes]>`: a bank of lowpasses with per-lane corner frequencies (coefficient planes in HBM)."""
    f0 = torch.rand(n_sections, 1, lanes, dtype=torch.float64) * 0.2 + 0.005
    w0 = 2 * math.pi * f0
    alpha = 0.5 * torch.sin(w0) * math.sqrt(2.0)
    b = 0.5 * (1 - torch.cos(w0))
    a0 = 1 + alpha
    ba = torch.cat([b / a0, 2 * b / a0, b / a0, 2 * torch.cos(w0) / a0, -(1 - alpha) / a0], 1)
    if dtype == torch.int32:
        coef = torch.clamp(torch.round(ba * float(1 << 30)), -(1 << 31), (1 << 31) - 1).to(torch.int32)
        extra = torch.tensor([3, -(1 << 30), 1 << 30], dtype=torch.int32)
        x = torch.randint(-(1 << 24), 1 << 24, (lanes * frames,), dtype=torch.int32, device=dev)
    else:
        coef = ba.to(dtype)
        extra = torch.tensor([0.01, -10.0, 10.0], dtype=dtype)
        x = torch.randn(lanes * frames, dtype=dtype, device=dev)
    if cv == 8:
        coef = torch.cat([coef, extra.view(1, 3, 1).expand(n_sections, 3, lanes)], 1)
    coef = coef.contiguous().to(dev)
    x, y = out_like(x)
    st = torch.zeros((words * n_sections, lanes), dtype=torch.int32, device=dev)
    pre = (p(coef), 30) if dtype == torch.int32 else (p(coef),)

    def run():
        call(op + "_bylane", *pre, n_sections, p(st), p(x), p(y), lanes, frames, layout, sptr())

    med, mn = timeit(run, iters)
    esz = x.element_size()
    passes = (n_sections + 3) // 4  # FrameMajor from 40960 lanes: three or four sections per pass on the two-wave kernel (else two)
    report(f"{tag}:{op}_bylane x{n_sections} {'LM' if layout else 'FM'} {lanes}x{frames}", lanes * frames, "sample",
           2 * esz * lanes * frames * passes + coef.numel() * esz, med, mn)


def cic(kind, dtype, order, rate, lanes, frames, layout, iters, tag):
    cfg = _abi.Cic(order, 1, rate)
    R = rate + 1
    hi = torch.randint(-(1 << 20), 1 << 20, (lanes * frames * R,), dtype=dtype, device=dev)
    lo = torch.randint(-(1 << 20), 1 << 20, (lanes * frames,), dtype=dtype, device=dev)
    esz = hi.element_size()
    st = torch.zeros((call("cic_state_words", C.byref(cfg), esz * 8), lanes), dtype=torch.int32, device=dev)
    x, y = (hi, lo) if kind == "dec" else (lo, hi)
    name = f"cic_{kind}_{'i64' if esz == 8 else 'i32'}"

    def run():
        call(name, C.byref(cfg), p(st), p(x), p(y), lanes, frames, layout, sptr())

    med, mn = timeit(run, iters)
    report(f"{tag}:{name} N={order} R={R} {'LM' if layout else 'FM'} {lanes}x{frames}", lanes * frames * R, "sample",
           (R + 1) * esz * lanes * frames, med, mn)


def wdf(lanes, frames, layout, iters, tag):
    """three 2nd-order + one 1st-order section: the 7th-order branch of the reference's embedded bench"""
    secs = (_abi.Wdf * 4)()
    for d, (m, g) in zip(secs, [(0xAD, [-0.9, 0.9]), (0xAD, [-0.6, 0.7]), (0xAD, [-0.7, 0.6]), (0xA, [0.8])]):
        call("wdf_quantize", len(g), m, (C.c_double * len(g))(*g), C.byref(d))
    x = torch.randint(-(1 << 24), 1 << 24, (lanes * frames,), dtype=torch.int32, device=dev)
    x, y = out_like(x)
    st = torch.zeros((7, lanes), dtype=torch.int32, device=dev)

    def run():
        call("wdf_i32", C.cast(secs, C.c_void_p), 4, p(st), p(x), p(y), lanes, frames, layout, sptr())

    med, mn = timeit(run, iters)
    report(f"{tag}:wdf 7th order {'LM' if layout else 'FM'} {lanes}x{frames}", lanes * frames, "sample", 8 * lanes * frames, med, mn)


def cossin(n, iters, tag):
    ph = torch.randint(-(1 << 31), (1 << 31) - 1, (n,), dtype=torch.int64, device=dev).to(torch.int32)
    out = torch.empty(2 * n, dtype=torch.int32, device=dev)

    def run():
        call("cossin_i32", p(ph), p(out), n, sptr())

    med, mn = timeit(run, iters)
    report(f"{tag}:cossin {n}", n, "phase", 12 * n, med, mn)


def atan2(n, iters, tag):
    xy = torch.randint(-(1 << 31), (1 << 31) - 1, (2 * n,), dtype=torch.int64, device=dev).to(torch.int32)
    out = torch.empty(n, dtype=torch.int32, device=dev)

    def run():
        call("atan2_i32", p(xy), p(out), n, sptr())

    med, mn = timeit(run, iters)
    report(f"{tag}:atan2 {n}", n, "angle", 12 * n, med, mn)


def fm_disc(lanes, frames, layout, iters, tag):
    cfg = _abi.FmDisc()
    q = _abi.BiquadI32()
    call("biquad_i32_from_sos", (C.c_double * 6)(*lowpass_sos(0.02)), 30, C.byref(q))
    cfg.carrier, cfg.deemph = 0x19341234, q
    x = torch.randint(-(1 << 31), (1 << 31) - 1, (2 * lanes * frames,), dtype=torch.int64, device=dev).to(torch.int32)
    y = torch.empty(lanes * frames, dtype=torch.int32, device=dev)
    st = torch.zeros((7, lanes), dtype=torch.int32, device=dev)

    def run():
        call("fm_disc_i32", C.byref(cfg), p(st), p(x), p(y), lanes, frames, layout, sptr())

    med, mn = timeit(run, iters)
    report(f"{tag}:fm_disc {'LM' if layout else 'FM'} {lanes}x{frames}", lanes * frames, "sample", 12 * lanes * frames, med, mn)


def lockin_generic(form, n, lanes, frames, layout, iters, tag):
    """`Lockin<C>` with biquad arms: form "phase" (idsp_lockin_i32_biquad_process), "lo" (i32, external LO), "lo_f32"
    (the ddc_lockin graph); algorithmic bytes per sample: 4 in + 8 out (+ 8 of LO)"""
    f32 = form == "lo_f32"
    if f32:
        q = _abi.BiquadF32()
        call("biquad_f32_from_sos_f64", (C.c_double * 6)(*lowpass_sos(0.01)), C.byref(q))
        cfg = (_abi.BiquadF32 * n)(*([q] * n))
        x = torch.randn(lanes * frames, dtype=torch.float32, device=dev)
        lo = torch.randn(2 * lanes * frames, dtype=torch.float32, device=dev)
        y = torch.empty(2 * lanes * frames, dtype=torch.float32, device=dev)
    else:
        q = _abi.BiquadI32()
        call("biquad_i32_from_sos", (C.c_double * 6)(*lowpass_sos(0.01)), 30, C.byref(q))
        cfg = (_abi.BiquadI32 * n)(*([q] * n))
        x = torch.randint(-(1 << 28), 1 << 28, (lanes * frames,), dtype=torch.int32, device=dev)
        lo = torch.randint(-(1 << 31), (1 << 31) - 1, (2 * lanes * frames,), dtype=torch.int64, device=dev).to(torch.int32)
        y = torch.empty(2 * lanes * frames, dtype=torch.int32, device=dev)
    words = call("lockin_biquad_state_words", n, 1 if form == "phase" else 0)
    st = torch.zeros((words, lanes), dtype=torch.int32, device=dev)
    if form == "phase":
        st[1] = torch.randint(-(1 << 31), (1 << 31) - 1, (lanes,), dtype=torch.int64, device=dev).to(torch.int32)

        def run():
            call("lockin_i32_biquad_process", C.cast(cfg, C.c_void_p), n, p(st), p(x), p(y), lanes, frames, layout, sptr())
    else:
        entry = "lockin_f32_biquad_lo_process" if f32 else "lockin_i32_biquad_lo_process"

        def run():
            call(entry, C.cast(cfg, C.c_void_p), n, p(st), p(x), p(lo), p(y), lanes, frames, layout, sptr())

    med, mn = timeit(run, iters)
    report(f"{tag}:lockin<biquad x{n}> {form} {'LM' if layout else 'FM'} {lanes}x{frames}", lanes * frames, "sample",
           (12 if form == "phase" else 20) * lanes * frames, med, mn)


def copy_ref(nbytes, iters):
    a = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
    b = torch.empty_like(a)
    med, mn = timeit(lambda: b.copy_(a), iters)
    report(f"ref:torch copy {nbytes >> 20} MiB", nbytes // 4, "word", 2 * nbytes, med, mn)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    sel = set(a.only.split(",")) if a.only else None
    it = a.iters
    FM, LM = 0, 1

    def want(k):
        return sel is None or k in sel

    if want("ref"):
        copy_ref(1 << 30, it)
    if sel == {"lmstride"}:  # probe: does the LaneMajor row pitch (frames * 4 bytes) alias HBM channels?
        for fr in (4096, 4160, 4224, 4352, 5120, 8192):
            biquad("biquad_i32_df1", torch.int32, 4, 65536, fr, LM, 1, it, "lmstride")
        for fr in (4096, 5120):
            hbf("dec", 4, 16384, fr, LM, it, "lmstride")
            hbf("int", 4, 16384, fr, LM, it, "lmstride")
            cic("dec", torch.int32, 3, 15, 16384, fr, LM, it, "lmstride")
            cic("int", torch.int32, 3, 15, 16384, fr // 4, LM, it, "lmstride")
            lockin(2, 2, 32768, fr, LM, it, "lmstride")
    if want("c2"):
        for layout in (FM, LM):
            biquad("biquad_i32_df1", torch.int32, 4, 65536, 4096, layout, 1, it, "C2")
    if want("ragged"):
        # lane counts that are not multiples of 256 (reference: any N in `Lanes<C>`, dsp-process/src/compose.rs:468): 65000 and
        # 100000 are multiples of 4 (LDS-DMA kernel with a ragged last block), 65537 is not (its last lane on a second stream); their
        # 256-multiples beside them
        for lanes in (65536, 65000, 65537, 65532, 65540, 69632, 73728, 73731, 99840, 100000, 100001, 131072, 131076, 131073, 147456, 32768, 32769, 32771, 16384, 16385, 8193, 40001):
            biquad("biquad_i32_df1", torch.int32, 4, lanes, 4096, FM, 1, it, "ragged")
        for lanes in (65000, 65537, 100000):
            biquad("biquad_i32_df1", torch.int32, 4, lanes, 4096, LM, 1, it, "ragged")
        # the f32 sections at the small lane counts (round 6: where the i32 DF1 sits on one wave's chain of five v_mad_i64_i32)
        for lanes in (8192, 16384, 32768):
            biquad("biquad_f32_df1", torch.float32, 4, lanes, 4096, FM, 1, it, "ragged")
            biquad("biquad_f32_df2t", torch.float32, 2, lanes, 4096, FM, 1, it, "ragged")
    if sel and "lanesweep" in sel:  # where does the one-workgroup-per-block launch lose to the persistent grid? (IDSP_DIAG=1 IDSP_LDS_GRID=0 forces the former)
        for blocks in (256, 288, 320, 352, 384, 416, 448, 480, 512, 576, 640, 768):
            biquad("biquad_i32_df1", torch.int32, 4, blocks * 256, 4096, FM, 1, it, "lanesweep")
    if want("i32var"):
        for op, w in (("biquad_i32_df1_clamp", 4), ("biquad_i32_dither", 5), ("biquad_i32_dither_clamp", 5),
                      ("biquad_i32_wide", 6), ("biquad_i32_wide_clamp", 6)):
            biquad(op, torch.int32, w, 65536, 4096, FM, 1, it, "C2v")
        biquad("biquad_i32_df1", torch.int32, 4, 65536, 4096, FM, 4, it, "C2v")
        biquad("cascade_i32_df1", torch.int32, 4, 65536, 4096, FM, 4, it, "C2v")
        biquad("cascade_i32_df1", torch.int32, 4, 65536, 4096, FM, 8, it, "C2v")
    if want("multi"):  # chains of 4 .. 8 sections (two-wave kernel on FrameMajor; IDSP_DIAG=1 IDSP_NO_DUO=1 for the single-wave kernels)
        for n in (4, 5, 6, 8):
            biquad("biquad_i32_df1", torch.int32, 4, 65536, 4096, FM, n, it, "multi")
        biquad("biquad_i32_df1_clamp", torch.int32, 4, 65536, 4096, FM, 4, it, "multi")
        biquad("biquad_i32_wide", torch.int32, 6, 65536, 4096, FM, 4, it, "multi")
        for n in (4, 8):
            biquad("biquad_f32_df2t", torch.float32, 2, 65536, 4096, FM, n, it, "multi")
            biquad("biquad_f32_df1", torch.float32, 4, 65536, 4096, FM, n, it, "multi")
        for n in (5, 8):
            biquad("cascade_i32_df1", torch.int32, 4, 65536, 4096, FM, n, it, "multi")
            biquad("cascade_f32_df1", torch.float32, 4, 65536, 4096, FM, n, it, "multi")
        biquad("biquad_i32_df1", torch.int32, 4, 131072, 4096, FM, 8, it, "multi")
        biquad("biquad_i32_df1", torch.int32, 4, 49152, 4096, FM, 4, it, "multi")
        biquad_bylane("biquad_i32_df1", torch.int32, 4, 5, 65536, 4096, FM, 4, it, "multi")
        biquad_bylane("biquad_f32_df2t", torch.float32, 2, 5, 65536, 4096, FM, 4, it, "multi")
        biquad_bylane("biquad_i32_df1_clamp", torch.int32, 4, 8, 65536, 4096, FM, 3, it, "multi")
    if want("bylane"):
        for layout in (FM, LM):
            biquad_bylane("biquad_i32_df1", torch.int32, 4, 5, 65536, 4096, layout, 1, it, "C2b")
        biquad_bylane("biquad_i32_df1_clamp", torch.int32, 4, 8, 65536, 4096, FM, 1, it, "C2b")
        biquad_bylane("biquad_i32_df1", torch.int32, 4, 5, 65536, 4096, FM, 2, it, "C2b")
        biquad_bylane("biquad_i32_wide_clamp", torch.int32, 6, 8, 65536, 4096, FM, 1, it, "C2b")
        biquad_bylane("biquad_f32_df2t", torch.float32, 2, 5, 65536, 4096, FM, 1, it, "C2b")
        biquad_bylane("biquad_f32_df1_clamp", torch.float32, 4, 8, 65536, 4096, FM, 1, it, "C2b")
        biquad_bylane("biquad_f32_df2t", torch.float32, 2, 5, 1 << 20, 4096, FM, 1, max(3, it // 3), "C5b")
    if want("f32"):
        for op, w in (("biquad_f32_df1", 4), ("biquad_f32_df2t", 2), ("biquad_f32_df1_clamp", 4), ("biquad_f32_df2t_clamp", 2)):
            for layout in (FM, LM):
                biquad(op, torch.float32, w, 65536, 4096, layout, 1, it, "f32")
    if want("c5"):
        # one GPU's shard of C5 at 8 GPUs (2^17 lanes) and the whole C5 on one GPU (2^20 lanes)
        biquad("biquad_f32_df2t", torch.float32, 2, 1 << 17, 4096, FM, 1, it, "C5/8")
        biquad("biquad_f32_df2t", torch.float32, 2, 1 << 20, 4096, FM, 1, max(3, it // 3), "C5")
        biquad("biquad_f32_df2t", torch.float32, 2, 1 << 20, 4096, LM, 1, max(3, it // 3), "C5")
    if sel and "c5sweep" in sel:  # lane counts beyond one resident wave of workgroups (IDSP_DIAG=1 IDSP_LDS_GRID=... per process)
        for lg in (16, 17, 18, 19, 20):
            biquad("biquad_f32_df2t", torch.float32, 2, 1 << lg, 4096, FM, 1, max(3, it >> max(0, lg - 17)), "C5s")
        for lg in (17, 20):
            biquad("biquad_i32_df1", torch.int32, 4, 1 << lg, 4096, FM, 1, max(3, it >> max(0, lg - 17)), "C5s")
            biquad("biquad_i32_df1_clamp", torch.int32, 4, 1 << lg, 4096, FM, 1, max(3, it >> max(0, lg - 17)), "C5s")
    if want("c3dec"):  # the decimator alone (profiling passes)
        hbf("dec", 4, 16384, 4096, LM, it, "C3")
        hbf("dec", 4, 16384, 4096, FM, it, "C3")
    if want("c3"):
        hbf("dec", 4, 16384, 4096, LM, it, "C3")
        hbf("dec", 4, 16384, 4096, FM, max(3, it // 3), "C3")
        hbf("int", 4, 16384, 4096, LM, it, "C3i")
        hbf("int", 4, 16384, 4096, FM, max(3, it // 3), "C3i")
    if want("hbfvar"):
        for s in (1, 2, 3, 5):
            hbf("dec", s, 16384, 65536 >> s, LM, it, "hbf")
            hbf("int", s, 16384, 65536 >> s, LM, it, "hbf")
    if want("c4small"):  # lock-in below one workgroup per CU and a quarter of C4's frames: which multi-wave form per [Lowpass<N>; K]
        for order, cascade in ((1, 1), (2, 1), (1, 2), (2, 2), (2, 4)):
            for lanes in (4096, 16384):
                lockin(order, cascade, lanes, 4096, FM, it, "C4s")
        lockin(2, 2, 16384, 4096, FM, it, "C4s", "arg")
        lockin(2, 1, 16384, 4096, FM, it, "C4s", "arg")
        lockin(2, 1, 32768, 4096, FM, it, "C4s", "arg")
        lockin(1, 1, 32768, 4096, FM, it, "C4s", "arg")
        lockin(2, 4, 32768, 4096, FM, it, "C4s", "arg")
        lockin(2, 4, 32768, 4096, FM, it, "C4s")
    if want("nw"):
        biquad("normal_i32_df1", torch.int32, 4, 65536, 4096, FM, 1, it, "nw")
        biquad("normal_f32_df1", torch.float32, 4, 65536, 4096, FM, 1, it, "nw")
        wdf(65536, 4096, FM, it, "nw")
        wdf(65536, 4096, LM, it, "nw")
    if want("cic"):
        for layout in (FM, LM):
            cic("dec", torch.int32, 3, 15, 16384, 4096, layout, it, "cic")
            cic("int", torch.int32, 3, 15, 16384, 4096, layout, it, "cic")
        cic("dec", torch.int64, 3, 15, 16384, 2048, FM, it, "cic")
        cic("dec", torch.int64, 3, 15, 16384, 2048, LM, it, "cic")
        cic("dec", torch.int32, 3, 63, 16384, 1024, FM, it, "cic")
        cic("dec", torch.int32, 3, 15, 65536, 1024, FM, it, "cic")
    if want("c4"):
        for layout in (FM, LM):
            lockin(2, 2, 32768, 4096, layout, it, "C4")
        lockin(1, 1, 32768, 4096, FM, it, "C4v")
        lockin(2, 2, 65536, 4096, FM, it, "C4v")
        for lanes in (32768, 65536):
            for out in ("arg", "iq+atan2", "norm_sqr"):
                lockin(2, 2, lanes, 4096, FM, it, "C4p", out)
        lockin(2, 2, 32768, 4096, LM, it, "C4p", "arg")
        dds(32768, 4096, FM, it, "dds")
        dds(65536, 4096, FM, it, "dds")
        cossin(1 << 27, it, "cossin")
        atan2(1 << 27, it, "atan2")
    if sel and "readouts" in sel:  # SURVEY 8(f) surfaces far below the HBM roof: one shape per kernel (issue-roof passes, tools/issue_roof.py)
        lockin(2, 2, 32768, 4096, FM, it, "f", "arg")
        lockin(2, 2, 32768, 4096, FM, it, "f", "iq+atan2")
        lockin(2, 2, 4096, 4096, FM, it, "f")
        fm_disc(65536, 4096, LM, it, "f")
        fm_disc(65536, 4096, FM, it, "f")
        wdf(65536, 4096, FM, it, "f")
        biquad("cascade_i32_df1", torch.int32, 4, 65536, 4096, FM, 8, it, "f")
        biquad("biquad_i32_wide", torch.int32, 6, 65536, 4096, FM, 4, it, "f")
        hbf("int", 4, 16384, 4096, FM, max(3, it // 3), "f")
    if want("lockinc"):  # `Lockin<C>` with biquad arms at the C4 shape (thread-per-lane stream kernels)
        for layout in (FM, LM):
            lockin_generic("phase", 1, 32768, 4096, layout, it, "C4g")
            lockin_generic("lo", 1, 32768, 4096, layout, it, "C4g")
            lockin_generic("lo_f32", 1, 32768, 4096, layout, it, "C4g")
        lockin_generic("phase", 2, 32768, 4096, FM, it, "C4g")
        lockin_generic("lo_f32", 2, 32768, 4096, FM, it, "C4g")
        lockin_generic("phase", 1, 65536, 4096, FM, it, "C4g")
    if sel and "ew" in sel:  # the element-wise entries alone
        cossin(1 << 27, it, "cossin")
        atan2(1 << 27, it, "atan2")
    if sel and "fmlm" in sel:  # the LaneMajor fm_disc kernel alone, one shape (profiling passes)
        fm_disc(65536, 4096, LM, it, "fm")
    if want("c4") or want("fm"):
        fm_disc(65536, 4096, FM, it, "fm")
        fm_disc(65536, 4096, LM, it, "fm")
        if sel and "fm" in sel:
            for lanes in (8192, 16384, 32768, 131072):
                fm_disc(lanes, 4096, FM, it, "fm")
            for lanes in (16384, 32768, 131072):
                fm_disc(lanes, 4096, LM, it, "fm")


if __name__ == "__main__":
    main()
