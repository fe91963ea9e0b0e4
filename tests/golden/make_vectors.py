#!/usr/bin/env python3
"""Generate tests/golden/vectors.npz — seeded input/expected-output vectors for every
operator of the hot path, computed by the executable spec model (oracle/spec.py, the
reference-shaped restatement; the reference itself is Rust and cannot run here).

Small by construction (a few lanes x <= 256 samples, adversarial values mixed in).  The C
oracle must reproduce them (tests/test_golden_vectors.py, CPU) and so must the HIP path
(-m gpu).  Regenerate with:  python tests/golden/make_vectors.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import spec  # noqa: E402

I32_MIN, I32_MAX = -(1 << 31), (1 << 31) - 1
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors.npz")


def adv_i32(rng, n):
    x = rng.integers(I32_MIN, I32_MAX, size=n, dtype=np.int64, endpoint=True)
    x[:: 7] = rng.choice([I32_MIN, I32_MAX, 0, 1, -1], size=len(x[::7]))
    return x.astype(np.int32)


def adv_f32(rng, n):
    x = rng.standard_normal(n).astype(np.float32)
    x[::9] = rng.choice(np.array([0.0, -0.0, 1e-41, 1e30, -1.0], np.float32), size=len(x[::9]))
    return x


def main():
    rng = np.random.default_rng(20260928)
    v = {}
    lanes, frames = 3, 160
    # ---- i32 biquads: per lane independent streams, LANE_MAJOR [lanes, frames]
    for frac in (30, 13):
        ba = spec.biquad_i32_from_sos(spec.filter_lowpass(0.05, 0.9), frac)
        sat = adv_i32(rng, 5).tolist()  # arbitrary bits: wrapping
        u, lo, hi = 12345, -(1 << 29), (1 << 29) + 7
        x = np.stack([adv_i32(rng, frames) for _ in range(lanes)])
        for name, coeffs in (("lp", ba), ("raw", sat)):
            key = f"i32_f{frac}_{name}"
            v[key + "_ba"] = np.array(coeffs, np.int64)
            v[key + "_x"] = x
            outs = {k: np.zeros_like(x) for k in ("df1", "df1c", "dit", "ditc", "wide", "widec")}
            for l in range(lanes):
                s1, s1c = spec.DirectForm1(), spec.DirectForm1()
                sd, sdc = spec.DirectForm1Dither(), spec.DirectForm1Dither()
                sw, swc = spec.DirectForm1Wide(), spec.DirectForm1Wide()
                for f in range(frames):
                    xv = int(x[l, f])
                    outs["df1"][l, f] = spec.biquad_i32_df1(coeffs, frac, s1, xv)
                    outs["df1c"][l, f] = spec.biquad_i32_df1_clamp(coeffs, frac, u, lo, hi, s1c, xv)
                    outs["dit"][l, f] = spec.biquad_i32_dither(coeffs, frac, sd, xv)
                    outs["ditc"][l, f] = spec.biquad_i32_dither_clamp(coeffs, frac, u, lo, hi, sdc, xv)
                    outs["wide"][l, f] = spec.biquad_i32_wide(coeffs, frac, sw, xv)
                    outs["widec"][l, f] = spec.biquad_i32_wide_clamp(coeffs, frac, u, lo, hi, swc, xv)
            for k, a in outs.items():
                v[f"{key}_{k}"] = a
    v["i32_clamp"] = np.array([12345, -(1 << 29), (1 << 29) + 7], np.int64)
    # ---- f32 biquads
    baf = np.array(spec.ba_from_sos_f64(spec.filter_lowpass(0.07, 1.3)), np.float32)
    xf = np.stack([adv_f32(rng, frames) for _ in range(lanes)])
    v["f32_ba"], v["f32_x"] = baf, xf
    v["f32_clamp"] = np.array([0.03125, -0.75, 0.875], np.float32)
    o = {k: np.zeros_like(xf) for k in ("df1", "df1c", "df2t", "df2tc")}
    uf, lof, hif = (float(t) for t in v["f32_clamp"])
    with np.errstate(all="ignore"):
        for l in range(lanes):
            a, b = spec.DirectForm1(np.float32(0)), spec.DirectForm1(np.float32(0))
            c, d = [np.float32(0)] * 2, [np.float32(0)] * 2
            for f in range(frames):
                o["df1"][l, f] = spec.biquad_f32_df1(baf, a, xf[l, f])
                o["df1c"][l, f] = spec.biquad_f32_df1_clamp(baf, uf, lof, hif, b, xf[l, f])
                o["df2t"][l, f] = spec.biquad_f32_df2t(baf, c, xf[l, f])
                o["df2tc"][l, f] = spec.biquad_f32_df2t_clamp(baf, uf, lof, hif, d, xf[l, f])
    for k, a in o.items():
        v["f32_" + k] = a
    # ---- half-band cascades (built-in tap sets), one lane each
    for ts, name in ((0, "HBF_TAPS"), (1, "HBF_TAPS_98")):
        table = getattr(spec, name)
        for stages in (1, 4, 5):
            R = 1 << stages
            nout = 40
            x = adv_f32(rng, nout * R) * np.float32(0.5)
            seq = [table[t] for t in range(stages - 1, -1, -1)]
            v[f"hbfdec_{ts}_{stages}_x"] = x
            v[f"hbfdec_{ts}_{stages}_y"] = np.array(spec.hbf_dec_cascade_block(seq, spec.hbf_dec_states(seq), x), np.float32)
            xi = adv_f32(rng, 24)
            seq = [table[t] for t in range(stages)]
            v[f"hbfint_{ts}_{stages}_x"] = xi
            v[f"hbfint_{ts}_{stages}_y"] = np.array(spec.hbf_int_cascade_block(seq, spec.hbf_int_states(seq), xi), np.float32)
    # ---- same-rate FIR, all four symmetry types
    taps = (rng.standard_normal(6) * 0.25).astype(np.float32)
    xs = adv_f32(rng, 100)
    v["fir_taps"], v["fir_x"] = taps, xs
    for kind, (odd, sym) in enumerate(((True, True), (False, True), (True, False), (False, False))):
        hist = [np.float32(0)] * (2 * len(taps) - 1 + int(odd))
        v[f"fir_{kind}_y"] = np.array(spec.fir_sym(taps.tolist(), odd, sym, hist, xs), np.float32)
    # ---- cossin, lowpass, lock-in
    ph = np.concatenate([adv_i32(rng, 500), (np.arange(-64, 64, dtype=np.int64) << 25).astype(np.int32)])
    v["cossin_phase"] = ph
    v["cossin_out"] = np.array([spec.cossin(int(p)) for p in ph], np.int32)
    for order, ks in ((1, [[1 << 27], [123456789]]), (2, [[70000, -(1 << 26)], [1 << 12, -300000000]])):
        xl = adv_i32(rng, 200)
        st = [[0] * order for _ in ks]
        v[f"lowpass{order}_k"] = np.array(ks, np.int64)
        v[f"lowpass{order}_x"] = xl
        v[f"lowpass{order}_y"] = np.array([spec.lowpass_cascade(ks, st, int(t)) for t in xl], np.int32)
        acc = spec.Accu(-77, 0x3C6EF372)
        siq = [[[0] * order for _ in ks] for _ in range(2)]
        v[f"lockin{order}_accu"] = np.array([-77, 0x3C6EF372], np.int64)
        v[f"lockin{order}_y"] = np.array([spec.lockin(ks, siq, int(t), acc.next()) for t in xl], np.int32)
    np.savez_compressed(OUT, **v)
    print(f"wrote {OUT}: {len(v)} arrays, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
