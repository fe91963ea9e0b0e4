"""The reference's own known-answer tests for the hot path, written once against
a back end (`tests/_backends.py`).  `test_oracle_kat.py` runs them on the CPU
oracle (pinning it), `test_gpu_reference_kat.py` replays them through the HIP
path.  Data comes from tests/golden/ref_kat.json (each entry cites the
reference file:line that asserts it)."""
from __future__ import annotations

import ctypes as C
import json
import math
import os

import numpy as np

from idsp_amd import _abi
from tests import _harness as H

FM, LM = H.FM, H.LM
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kat.json")))
INF = {"inf": math.inf, "-inf": -math.inf}


def _f(v):
    return INF.get(v, v) if isinstance(v, str) else v


def _f32bits(vals):
    return np.asarray(vals, dtype=np.float32).view(np.uint32)


def _sos_builder(be, entry):
    """coefficients::Filter::{lowpass,highpass} (f64) -> Biquad<Q32<F>> through the
    back end's own ingestion helper (idsp[_ref]_biquad_i32_from_sos)."""
    w0 = math.tau * entry["critical_frequency"]
    q = 1.0 / math.sqrt(2.0)  # Shape::default(), src/iir/coefficients.rs:19-22
    fsin, fcos = math.sin(w0), math.cos(w0)
    alpha = 0.5 * fsin * (1.0 / q)
    if entry["builder"] == "lowpass":  # src/iir/coefficients.rs:276-283
        b = entry["gain"] * 0.5 * (1.0 - fcos)
        sos = [b, 2.0 * b, b, 1.0 + alpha, -2.0 * fcos, 1.0 - alpha]
    else:  # src/iir/coefficients.rs:328-335
        b = entry["gain"] * 0.5 * (1.0 + fcos)
        sos = [b, -2.0 * b, b, 1.0 + alpha, -2.0 * fcos, 1.0 - alpha]
    out = _abi.BiquadI32()
    assert be.helper("biquad_i32_from_sos", (C.c_double * 6)(*sos), entry["frac"], C.byref(out)) == 0
    return out


def case_biquad_i32_filter(be):
    for e in KAT["biquad_i32_filter"]:
        cfg = _sos_builder(be, e)
        for layout in (FM, LM):
            x = np.array(e["x"], dtype=np.int32)  # `iir.inplace(&mut DirectForm1::default(), &mut xy)`
            st = np.zeros((4, 1), dtype=np.uint32)
            rc, y = be.stream("biquad_i32_df1", (_abi.BiquadI32 * 1)(cfg), 1, st, x, 1, len(x), layout, inplace=True)
            assert rc == 0
            assert y.tolist() == e["y"], e["cite"]


def case_biquad_f32_df1_state(be):
    e = KAT["biquad_f32_df1_state"]
    st = _f32bits(e["state_x"] + e["state_y"]).reshape(4, 1).copy()
    rc, y = be.stream("biquad_f32_df1", H.biquad_f32([e["ba"]]), 1, st, np.array([e["x0"]], np.float32), 1, 1, FM)
    assert rc == 0 and y[0] == e["y0"]
    assert st.view(np.float32)[:, 0].tolist() == e["after_x"] + e["after_y"]


def case_biquad_f32_simple(be):
    for e in KAT["biquad_f32_simple"]:
        st = _f32bits([0.0, 0.0] + e["state_y"]).reshape(4, 1).copy()
        rc, y = be.stream("biquad_f32_df1", H.biquad_f32([e["ba"]]), 1, st, np.array([e["x0"]], np.float32), 1, 1, LM)
        assert rc == 0 and y[0] == e["y0"], e["cite"]


def case_biquad_f32_clamp(be):
    for e in KAT["biquad_f32_clamp"]:
        # BiquadClamp::<f32>::default(): zero coefficients, u/min/max as given
        cfg = H.biquad_clamp_f32([([0.0] * 5, _f(e["u"]), _f(e["min"]), _f(e["max"]))])
        st = np.zeros((4, 1), dtype=np.uint32)
        rc, y = be.stream("biquad_f32_df1_clamp", cfg, 1, st, np.array([e["x0"]], np.float32), 1, 1, FM)
        assert rc == 0 and y[0] == e["y0"], e["cite"]


def case_biquad_f32_df2t_identity(be):
    for e in KAT["biquad_f32_df2t_identity"]:
        st = np.zeros((2, 1), dtype=np.uint32)
        x = np.array([e["x0"]], np.float32)
        if e["clamp"]:
            cfg = H.biquad_clamp_f32([([1.0, 0, 0, 0, 0], 0.0, -math.inf, math.inf)])
            rc, y = be.stream("biquad_f32_df2t_clamp", cfg, 1, st, x, 1, 1, FM)
        else:
            rc, y = be.stream("biquad_f32_df2t", H.biquad_f32([[1.0, 0, 0, 0, 0]]), 1, st, x, 1, 1, FM)
        assert rc == 0 and y[0] == e["y0"], e["cite"]


def case_biquad_i32_dither(be):
    e = KAT["biquad_i32_dither"]
    st = np.array(e["state_x"] + e["state_y"] + [e["e"]], dtype=np.uint32).reshape(5, 1)
    rc, y = be.stream("biquad_i32_dither", H.biquad_i32([(e["ba"], e["frac"])]), 1, st, np.array([e["x0"]], np.int32), 1, 1, FM)
    assert rc == 0 and y[0] == e["y0"]
    assert st[:, 0].tolist() == e["after_x"] + e["after_y"] + [e["after_e"]]


def _f32_from_sos(be, sos):
    out = _abi.BiquadF32()
    # `Biquad::from([[0.7, -0.4, 0.1], [1.0, -0.2, 0.05]])` infers f32 from the state type
    assert be.helper("biquad_f32_from_sos", (C.c_float * 6)(*sos), C.byref(out)) == 0
    return list(out.ba)


def case_biquad_f32_df1_vs_df2t(be):
    e = KAT["biquad_f32_df1_vs_df2t"]
    ba = _f32_from_sos(be, e["sos"])
    x = np.array(e["x"], np.float32)
    _, y1 = be.stream("biquad_f32_df1", H.biquad_f32([ba]), 1, np.zeros((4, 1), np.uint32), x, 1, len(x), LM)
    _, y2 = be.stream("biquad_f32_df2t", H.biquad_f32([ba]), 1, np.zeros((2, 1), np.uint32), x, 1, len(x), LM)
    assert np.all(np.abs(y1 - y2) < e["tol"])


def case_biquad_f32_cascade_vs_repeated(be):
    e = KAT["biquad_f32_cascade_vs_repeated"]
    ba = _f32_from_sos(be, e["sos"])
    n = e["sections"]
    x = np.array(e["x"], np.float32)
    _, yc = be.stream("cascade_f32_df1", H.biquad_f32([ba] * n), n, np.zeros((2 + 2 * n, 1), np.uint32), x, 1, len(x), LM)
    _, yr = be.stream("biquad_f32_df1", H.biquad_f32([ba] * n), n, np.zeros((4 * n, 1), np.uint32), x, 1, len(x), LM)
    assert np.all(np.abs(yc - yr) < e["tol"])
    assert np.array_equal(yc.view(np.uint32), yr.view(np.uint32))  # same expression tree: bit-identical


def case_hbf_dec_single(be):
    e = KAT["hbf_dec_single"]
    cfg = H.hbf_cfg([e["taps"]])
    st = np.zeros((1, 1), dtype=np.uint32)  # 3M-2 = 1 word
    x = np.array(e["x"], np.float32)
    # `h.block(&[], &mut [])` first (src/hbf.rs:549): an empty block is a no-op
    rc, _ = be.cfgcall("hbf_dec_f32", cfg, st, np.zeros(0, np.float32), (0,), np.float32, 1, 0, LM)
    assert rc == 0
    for layout in (LM, FM):
        st[:] = 0
        rc, y = be.cfgcall("hbf_dec_f32", cfg, st, x, (4,), np.float32, 1, 4, layout)
        assert rc == 0 and y.tolist() == e["y"]


def _cascade(be, kind, tap_set, stages):
    cfg = _abi.HbfCascadeF32()
    assert be.helper(f"hbf_{kind}_cascade", tap_set, stages, C.byref(cfg)) == 0
    return cfg


def case_hbf_response_length(be):
    e = KAT["hbf_response_length"]
    for d, n in e["dec"].items():
        assert be.helper("hbf_dec_response_length", C.byref(_cascade(be, "dec", e["tap_set"], int(d)))) == n
    for d, n in e["int"].items():
        assert be.helper("hbf_int_response_length", C.byref(_cascade(be, "int", e["tap_set"], int(d)))) == n


def case_hbf_dec_response(be):
    """src/hbf.rs:577-595: 100 random frames, then 64 zero frames: y[n-1] != 0, y[n] == 0."""
    cfg = _cascade(be, "dec", 0, 4)
    words = be.helper("hbf_dec_state_words", C.byref(cfg))
    assert words == 118
    st = np.zeros((words, 1), dtype=np.uint32)
    rng = np.random.default_rng(7)
    x = rng.random(100 * 16, dtype=np.float32)
    rc, _ = be.cfgcall("hbf_dec_f32", cfg, st, x, (100,), np.float32, 1, 100, LM)
    assert rc == 0
    rc, y = be.cfgcall("hbf_dec_f32", cfg, st, np.zeros(64 * 16, np.float32), (64,), np.float32, 1, 64, LM)
    n = be.helper("hbf_dec_response_length", C.byref(cfg))
    assert rc == 0 and n == 57
    assert y[n - 1] != 0.0
    assert y[n] == 0.0 and not np.any(y[n:])


def case_hbf_int_response_and_spectrum(be):
    """src/hbf.rs:598-633: impulse response length and spectral mask of the x16 interpolator."""
    e = KAT["hbf_spectrum"]
    R = e["depth"]
    cfg = _cascade(be, "int", 0, R)
    r = be.helper("hbf_int_response_length", C.byref(cfg))
    nin = (r >> R) + 1
    x = np.zeros(nin, np.float32)
    x[0] = 1.0
    st = np.zeros((be.helper("hbf_int_state_words", C.byref(cfg)), 1), dtype=np.uint32)
    rc, y = be.cfgcall("hbf_int_f32", cfg, st, x, (nin << R,), np.float32, 1, nin, LM)
    assert rc == 0
    assert y[r] != 0.0
    assert not np.any(y[r + 1:])
    z = np.zeros(e["fft_len"], dtype=np.float64)
    z[: y.size] = y.astype(np.float64) / (1 << R)
    p = 10.0 * np.log10(np.abs(np.fft.fft(z)) ** 2)
    f = p.size / (1 << R)
    p_pass = np.max(np.abs(p[: int(math.floor(f * e["passband"]))]))
    assert p_pass < e["max_passband_ripple_db"], p_pass
    p_stop = np.max(p[int(math.ceil(f * (1.0 - e["passband"]))): p.size // 2])
    assert p_stop < e["max_stopband_db"], p_stop


def case_cossin_bounds(be):
    """src/cossin.rs:131-196 over 2^20 phases."""
    e = KAT["cossin_bounds"]
    depth = e["phase_depth"]
    amp = float(1 << 31) - 0.85 * float(1 << 15)
    phase = (np.arange(1 << depth, dtype=np.int64) << (32 - depth)).astype(np.uint32).view(np.int32)
    rc, cs = be.cossin(phase)
    assert rc == 0
    have = cs.astype(np.float64) / amp
    rad = 2.0 * np.pi * phase.astype(np.float64) / float(1 << 32)
    want = np.stack([np.cos(rad), np.sin(rad)], axis=1)
    err = have - want
    assert abs(math.fsum(have[:, 0])) < e["sum_cos"]
    assert abs(math.fsum(have[:, 1])) < e["sum_sin"]
    assert abs(math.fsum(have[:, 0] * want[:, 0] - have[:, 1] * want[:, 1])) < e["demod_re"]
    assert abs(math.fsum(have[:, 1] * want[:, 0] + have[:, 0] * want[:, 1])) < e["demod_im"]
    assert abs(math.fsum(err[:, 0])) < e["sum_err"] and abs(math.fsum(err[:, 1])) < e["sum_err"]
    assert np.sqrt(np.mean(err ** 2, axis=0)).max() < e["rms"]
    assert np.abs(err).max() < e["max"]
    s = KAT["survey_checksums"]
    rc, c0 = be.cossin(np.array([0], np.int32))
    assert c0[0].tolist() == s["cossin_0"]


def case_cossin_spur(be):
    """src/cossin.rs:199-230: complex DDS at bin k, first spur pair at (M +- 1)k, -120.4 dBc."""
    e = KAT["cossin_spur"]
    n, k = 1 << e["dds_log2"], e["k"]
    st = np.zeros((2, 1), dtype=np.uint32)
    st[1, 0] = np.uint32(k << (32 - e["dds_log2"]))
    rc, out = be.dds(st, 1, n, LM)
    assert rc == 0
    amp = float(1 << 31) - 0.85 * float(1 << 15)
    z = (out[0::2] + 1j * out[1::2]) / amp
    power = np.abs(np.fft.fft(z)) ** 2
    m = 8 * 128
    lo, hi = n - ((m - 1) * k) % n, ((m + 1) * k) % n
    for b in (lo, hi):
        assert abs(10 * np.log10(power[b] / power[k]) - e["spur_dbc"]) < e["tol_db"]
    rest = power.copy()
    rest[k] = 0
    assert int(np.argmax(rest)) in (lo, hi)


def case_atan2_zero_axis(be):
    e = KAT["atan2_zero_axis"]
    xy = np.array([[c["x"], c["y"]] for c in e["cases"]], np.int32)
    rc, out = be.atan2(xy)
    assert rc == 0 and out.tolist() == [c["want"] for c in e["cases"]]


def case_atan2_absolute_error(be):
    """src/atan2.rs:117-153: 323 x 323 grid against f64 atan2."""
    e = KAT["atan2_absolute_error"]
    n, scale = e["n"], float(1 << 31)
    vals = [int(scale * (-1.0 + 2.0 * i / n)) for i in range(n)]
    assert -(1 << 31) in vals
    vals += [(1 << 31) - 1, 0]
    v = np.array(vals, np.int64)
    xs, ys = np.repeat(v, v.size), np.tile(v, v.size)
    rc, out = be.atan2(np.stack([xs, ys], 1).astype(np.int32))
    assert rc == 0
    want = np.arctan2(ys.astype(np.float64), xs.astype(np.float64))
    err = np.abs(out.astype(np.float64) * (math.pi / scale) - want)
    assert err.max() < e["abs_err"]
    assert math.sqrt(float((err * err).sum())) / v.size < e["rms_err"]
    assert not (err > e["rel_threshold"]).any()  # rel_err stays 0 < 1e-15


def case_atan2_small(be):
    """src/atan2.rs:155-175: small equal inputs and small vectors near the origin."""
    e = KAT["atan2_small"]
    scale = math.pi / float(1 << 31)
    v = np.arange(*e["equal_range"], dtype=np.int32)
    rc, out = be.atan2(np.stack([v, v], 1))
    assert rc == 0 and np.abs(out * scale - math.pi / 4).max() < e["abs_err"]
    pts = np.array([(x, y) for x in range(*e["near_origin_x"]) for y in range(x + 1)], np.int32)
    rc, out = be.atan2(pts)
    want = np.arctan2(pts[:, 1].astype(np.float64), pts[:, 0].astype(np.float64))
    assert rc == 0 and np.abs(out * scale - want).max() < e["abs_err"]


def _cic(be, kind, cfg, x, lanes, frames, layout, state=None, dtype=np.int64):
    """kind 'dec' / 'int'; returns (y, state)."""
    R = cfg.rate + 1
    bits = 64 if dtype == np.int64 else 32
    words = be.helper("cic_state_words", C.byref(cfg), bits)
    st = np.zeros((words, lanes), np.uint32) if state is None else state
    suffix = "i64" if dtype == np.int64 else "i32"
    n_out = lanes * frames * (1 if kind == "dec" else R)
    rc, y = be.cfgcall(f"cic_{kind}_{suffix}", cfg, st, np.asarray(x, dtype), (n_out,), dtype, lanes, frames, layout)
    assert rc == 0
    return y, st


def case_cic_identity_and_unit_rate(be):
    """src/cic.rs:223-240,286-306"""
    e = KAT["cic"]["identity"]
    cfg = _abi.Cic(e["order"], e["comb_delay"], e["rate"])
    rng = np.random.default_rng(12)
    x = rng.integers(-(1 << 62), 1 << 62, size=200, dtype=np.int64)
    y, st = _cic(be, "dec", cfg, x, 1, 200, LM)
    assert np.array_equal(y, x)
    assert int(st[0, 0]) | (int(st[1, 0]) << 32) == int(x[-1]) & 0xFFFFFFFFFFFFFFFF  # get_decimate() == zoh
    y, _ = _cic(be, "int", cfg, x >> 3, 1, 200, LM)
    assert np.array_equal(y, x >> 3)
    u = KAT["cic"]["unit_rate"]
    cfg = _abi.Cic(u["order"], u["comb_delay"], u["rate"])
    assert be.helper("cic_gain_log2", C.byref(cfg)) == u["gain_log2"] and be.helper("cic_gain", C.byref(cfg)) == u["gain"]
    x = np.concatenate([rng.integers(-(1 << 31), 1 << 31, size=5), np.zeros(100, np.int64)]).astype(np.int64)
    y, st = _cic(be, "dec", cfg, x, 1, x.size, LM)
    assert np.all(y[5 + u["order"] * u["comb_delay"]:] == 0)  # FIR of length N * (M - 1) + 1 <= 9 flushed


def case_cic_step_response(be):
    """src/cic.rs:242-262 with the chunked interpolator: one Some(x) per tick, R outputs per input."""
    e = KAT["cic"]["step"]
    for rate in e["rates"]:
        cfg = _abi.Cic(e["order"], e["comb_delay"], rate)
        shift = be.helper("cic_gain_log2", C.byref(cfg))
        gain = be.helper("cic_gain", C.byref(cfg))
        assert shift < 32 and gain <= 1 << shift
        resp = be.helper("cic_response_length", C.byref(cfg))
        assert resp == rate * e["order"]
        R = rate + 1
        chunks = -(-2 * resp // R) + 1
        st, y_last = None, 0
        for xv in e["x"]:
            y, st = _cic(be, "int", cfg, np.full(chunks, xv, np.int64), 1, chunks, LM, state=st)
            want = xv * gain
            for i, v in enumerate(y.tolist()):
                if i < resp:
                    if want > y_last:
                        assert y_last <= v < want
                    elif want < y_last:
                        assert want <= v - 1 < y_last
                    else:
                        assert v == want
                else:
                    assert v == want
            y_last = int(y[-1])


def case_cic_modular_composition(be):
    """src/cic.rs:348-383: Cic == Integrator^N -> Downsample -> Comb^N (and the interpolating dual)."""
    from oracle import spec

    xd = [v * 3 - 7 for v in range(-31, 65)]
    xi = [v * v - 3 * v + 2 for v in range(-12, 20)]
    for n, r, m in KAT["cic"]["modular"]["cases"]:
        cfg = _abi.Cic(n, m, r - 1)
        frames = len(xd) // r
        y, _ = _cic(be, "dec", cfg, xd[:frames * r], 1, frames, LM)
        assert y.tolist() == spec.cic_modular_decimator(n, r, m, [xd[i * r:(i + 1) * r] for i in range(frames)])
        y, _ = _cic(be, "int", cfg, xi, 1, len(xi), LM)
        assert y.reshape(len(xi), r).tolist() == spec.cic_modular_interpolator(n, r, m, xi)


def fm_disc_fixture():
    """examples/fm_disc.rs:57-75 (`fm_signal`) and :99-108 (`lowpass_reference`) in f32 like the example."""
    e = KAT["fm_disc"]
    f32 = np.float32
    phase, phases, msg = 0, [], []
    tau = f32(2 * math.pi)
    for i in range(e["n"]):
        m = f32(np.sin(tau * f32(e["message_freq"]) * f32(i)))
        inc = e["carrier"] + int(f32(e["deviation"]) * m)  # `carrier as i32 + (deviation as f32 * msg) as i32`
        phase = (phase + inc + (1 << 31)) % (1 << 32) - (1 << 31)
        phases.append(phase)
        msg.append(m)
    return e, np.array(phases, np.int32), np.array(msg, f32)


def case_fm_disc_tracks_known_modulation(be):
    """examples/fm_disc.rs:148-157: corr > 0.999, 0.95 < gain < 1.05, rms < 5e-4."""
    from oracle import spec_coeff as S

    e, phases, msg = fm_disc_fixture()
    rc, xs = be.cossin(phases)  # `Complex::new(Q32::from_bits(re), Q32::from_bits(im))`
    assert rc == 0
    # `Filter::default().critical_frequency(cutoff).lowpass()` in f32 -> Biquad<Q32<30>> / Biquad<f32>
    sos = S.filter_build(np.float32, S.LOWPASS, np.float32(math.tau) * np.float32(e["cutoff"]), 1.0, 1.0, S.SHAPE_Q,
                         np.float32(1.0) / np.float32(math.sqrt(2.0)))
    qba = S.normalize(np.float32, sos, S.Out("i32", e["frac"]))
    fba = S.normalize(np.float32, sos, S.Out("f32"))
    cfg = _abi.FmDisc()
    cfg.carrier = e["carrier"]
    cfg.deemph.ba[:] = qba
    cfg.deemph.frac = e["frac"]
    st = np.zeros((_abi.FM_DISC_STATE_WORDS, 1), np.uint32)
    rc, y = be.cfgcall("fm_disc_i32", cfg, st, xs.reshape(-1), (e["n"],), np.int32, 1, e["n"], LM)
    assert rc == 0 and y[0] == 0 and st[0, 0] == 1
    scale = np.float32(math.tau) / np.float32(4294967296.0)
    yf = y.astype(np.float32) * scale
    # lowpass_reference: Biquad<f32> DF1 over deviation * scale * msg
    from oracle import spec

    d1 = spec.DirectForm1()
    m = np.array([spec.biquad_f32_df1(fba, d1, np.float32(e["deviation"]) * scale * v) for v in msg], np.float32)
    yy, mm = yf[e["skip"]:].astype(np.float64), m[e["skip"]:].astype(np.float64)
    gain = float((yy * mm).sum() / (mm * mm).sum())
    rms = float(np.sqrt(((yy - gain * mm) ** 2).sum()) / yy.size)
    corr = float((yy * mm).sum() / (np.sqrt((yy * yy).sum()) * np.sqrt((mm * mm).sum())))
    assert corr > e["corr_min"] and e["gain"][0] < gain < e["gain"][1] and rms < e["rms_max"], (corr, gain, rms)


def case_accu(be):
    e = KAT["accu"]
    st = np.array([[e["state"]], [e["step"]]], dtype=np.int64).astype(np.uint32)
    rc, out = be.dds(st, 1, 2, FM)
    assert rc == 0
    o = H.oracle()
    want = [o.cossin(p) for p in e["next"]]
    assert out.reshape(2, 2).tolist() == [list(w) for w in want]
    assert np.int32(st[0, 0]) == e["next"][-1]


def case_lane_views(be):
    e = KAT["lane_major_lanes"]
    cfg = H.biquad_clamp_f32([([1.0, 0, 0, 0, 0], float(e["offset"]), -math.inf, math.inf)])
    x = np.array(e["x"], np.float32)
    st = np.zeros((4, e["lanes"]), dtype=np.uint32)
    rc, y = be.stream("biquad_f32_df1_clamp", cfg, 1, st, x, e["lanes"], e["frames"], LM)
    assert rc == 0 and y.tolist() == [float(v) for v in e["y"]]
    f = KAT["frame_major_fallback"]
    x = np.array(f["frames_x"], np.float32)
    st = np.zeros((4, 2), dtype=np.uint32)
    rc, y = be.stream("biquad_f32_df1_clamp", cfg, 1, st, x, 2, 2, FM)
    assert rc == 0 and y.tolist() == [[float(v) for v in r] for r in f["frames_y"]]


ALL_CASES = [v for k, v in sorted(globals().items()) if k.startswith("case_")]
