"""Host-side coefficient front-end of the C ABI (`idsp_filter_build`, `idsp_pid_build_*`,
`idsp_pid_build_clamp_*`, `idsp_config_*_build_*`): the reference's own tests
(tests/golden/ref_kat.json) replayed on the product library AND on the independent Python
restatement (oracle/spec_coeff.py), then product == restatement on randomised parameters.
Pure host code: runs without a GPU."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from idsp_amd import _abi
from idsp_amd._lib import load
from oracle import spec
from oracle import spec_coeff as S

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kat.json")))
TYPES = {n: i for i, n in enumerate(_abi.FILTER_TYPES)}
EPS32 = float(np.finfo(np.float32).eps)
ACT = {"I2": 0, "I": 1, "P": 2, "D": 3, "D2": 4}


@pytest.fixture(scope="module")
def fn():
    return load()[0]


def err(fn):
    return fn["last_error"]().decode()


# ------------------------------------------------------------------ thin callers of the product
def p_filter(fn, typ, w0, gain=1.0, shelf=1.0, kind=S.SHAPE_Q, shape=1 / math.sqrt(2.0), f32=False, validate=0):
    f = _abi.Filter(w0, gain, shelf, shape, kind, int(f32))
    ba = (C.c_double * 6)()
    rc = fn["filter_build"](C.byref(f), typ, validate, ba)
    return rc, list(ba)


def p_builder(order, gain, limit, f32):
    b = _abi.PidBuilder()
    b.order, b.f32 = order, int(f32)
    b.gain[:] = gain
    b.limit[:] = limit
    return b


def p_pid(fn, kind, order, gain, limit, period, f32=False, frac=0, validate=0):
    b = p_builder(order, gain, limit, f32)
    if kind == "i32":
        out = (C.c_int32 * 5)()
        rc = fn["pid_build_i32"](C.byref(b), period, validate, frac, out)
    else:
        out = ((C.c_float if kind == "f32" else C.c_double) * 5)()
        rc = fn["pid_build_" + kind](C.byref(b), period, validate, out)
    return rc, list(out)


REC = {"i32": _abi.BiquadClampI32, "f32": _abi.BiquadClampF32, "f64": _abi.BiquadClampF64}


def p_clamp(fn, name, kind, cfg, units, frac=0, validate=0):
    out = REC[kind]()
    u = _abi.Units(*units)
    args = (C.byref(cfg), C.byref(u), validate) + ((frac,) if kind == "i32" else ()) + (C.byref(out),)
    rc = fn[f"{name}_{kind}"](*args)
    return rc, (list(out.ba), out.u, out.min, out.max)


def same(kind, have, want, ulps=0):
    """Bitwise (ulps == 0) or ULP-bounded equality of product vs restatement values."""
    if kind == "i32":
        return int(have) == int(want)
    dt = np.float32 if kind == "f32" else np.float64
    a, b = dt(have), dt(want)
    if np.isnan(a) or np.isnan(b):
        return bool(np.isnan(a) and np.isnan(b))
    if ulps == 0 or not (np.isfinite(a) and np.isfinite(b)):
        return a == b and np.signbit(a) == np.signbit(b)
    it = np.int32 if kind == "f32" else np.int64
    return abs(int(a.view(it)) - int(b.view(it))) <= ulps or abs(float(a) - float(b)) <= ulps * float(np.finfo(dt).eps) * max(abs(float(a)), abs(float(b)))


def close32(have, want, rel, scale=1.0):
    """f32 trigonometric paths: non-finite values must agree exactly, finite ones within rel * max|want|."""
    fin = [abs(float(w)) / scale for w in want if math.isfinite(float(w))]
    tol = rel * max([1e-30] + fin)
    for h, w in zip(have, want):
        h, w = float(h), float(w)
        if not (math.isfinite(h) and math.isfinite(w)):
            if not (h == w or (math.isnan(h) and math.isnan(w))):
                return False
        elif abs(h - w) / scale > tol:
            return False
    return True


# ------------------------------------------------------------------ reference KATs
def _mask_filter(case):
    kind, shape = S.SHAPE_Q, 1 / math.sqrt(2.0)
    if "bandwidth" in case:
        kind, shape = S.SHAPE_BANDWIDTH, case["bandwidth"]
    if "q" in case:
        kind, shape = S.SHAPE_Q, case["q"]
    gain = math.pow(10.0, case.get("gain_db", 0.0) / 20.0)    # Filter::gain_db, coefficients.rs:157-159
    shelf = math.pow(10.0, case.get("shelf_db", 0.0) / 20.0)  # Filter::shelf_db, :177-179
    return TYPES[case["type"]], math.tau * case["f0"], gain, shelf, kind, shape


def _check_mask(ba, mask, what):
    for f, tol in mask:
        h = S.freqz(ba[:3], ba[3:], f)
        g = 10.0 * math.log10(h.real * h.real + h.imag * h.imag) if h != 0 else -math.inf
        if "gain_db" in tol:
            assert abs(g - tol["gain_db"][0]) <= tol["gain_db"][1], (what, f, g)
        else:
            assert g <= tol["below_db"], (what, f, g)


@pytest.mark.parametrize("case", KAT["filter_masks"]["cases"], ids=lambda c: c["type"])
@pytest.mark.parametrize("impl", ["product", "restatement"])
def test_reference_transfer_masks(fn, case, impl):
    """src/iir/coefficients.rs:695-845 via check_transfer (:674-692)."""
    typ, w0, gain, shelf, kind, shape = _mask_filter(case)
    if impl == "product":
        rc, ba = p_filter(fn, typ, w0, gain, shelf, kind, shape, validate=1)
        assert rc == 0, err(fn)
        q = _abi.BiquadI32()
        assert fn["biquad_i32_from_sos"]((C.c_double * 6)(*ba), 30, C.byref(q)) == 0
        qba = list(q.ba)
    else:
        S.filter_validate(np.float64, w0, gain, shelf, kind, shape)
        ba = [float(v) for v in S.filter_build(np.float64, typ, w0, gain, shelf, kind, shape)]
        qba = S.normalize(np.float64, [np.float64(v) for v in ba], S.Out("i32", 30))
    _check_mask(ba, case["mask"], "f64")
    bai = [c * 2.0 ** -30 for c in qba]  # Q32<30> -> f64 (num_traits_impl.rs:48-57)
    _check_mask([bai[0], bai[1], bai[2], 1.0, -bai[3], -bai[4]], case["mask"], "Q30")


@pytest.mark.parametrize("impl", ["product", "restatement"])
def test_reference_q30_doctests(fn, impl):
    """src/iir/coefficients.rs:289-300,316-326: Filter -> Biquad<Q32<30>> -> DF1 in place."""
    e = KAT["filter_q30_doctests"]
    for typ, key in ((S.LOWPASS, "lowpass_y"), (S.HIGHPASS, "highpass_y")):
        if impl == "product":
            rc, ba = p_filter(fn, typ, math.tau * e["f0"], e["gain"])
            q = _abi.BiquadI32()
            assert rc == 0 and fn["biquad_i32_from_sos"]((C.c_double * 6)(*ba), e["frac"], C.byref(q)) == 0
            qba = list(q.ba)
        else:
            sos = S.filter_build(np.float64, typ, math.tau * e["f0"], e["gain"], 1.0, S.SHAPE_Q, 1 / math.sqrt(2.0))
            qba = S.normalize(np.float64, sos, S.Out("i32", e["frac"]))
        st = spec.DirectForm1()
        assert [spec.biquad_i32_df1(qba, e["frac"], st, v) for v in e["x"]] == e[key]


def _gl(d, default):
    v = [default] * 5
    for k, x in d.items():
        v[ACT[k]] = x
    return v


@pytest.mark.parametrize("impl", ["product", "restatement"])
def test_reference_pid_kat(fn, impl):
    """src/iir/pid.rs:574-590"""
    e = KAT["pid"]
    gain, limit = _gl(e["gain"], 0.0), _gl(e["limit"], math.inf)
    if impl == "product":
        rc, ba = p_pid(fn, "f32", S.ORDER_I, gain, limit, e["period"], validate=1)
        assert rc == 0, err(fn)
    else:
        ba = S.builder_build(np.float64, S.ORDER_I, gain, limit, e["period"], S.Out("f32"))
    for have, want in zip(ba, e["want"]):
        assert abs(np.float32(have) / np.float32(want) - np.float32(1.0)) < e["rel_tol_eps"] * EPS32, (ba, e["want"])


def _df1_f32(ba, xs):
    st = spec.DirectForm1()
    return [spec.biquad_f32_df1([np.float32(v) for v in ba], st, np.float32(x)) for x in xs]


@pytest.mark.parametrize("impl", ["product", "restatement"])
def test_reference_pid_doctests(fn, impl):
    e = KAT["pid_doctests"]

    def build(order, gain, limit, period):
        if impl == "product":
            rc, ba = p_pid(fn, "f32", order, gain, limit, period)
            assert rc == 0
            return ba
        return S.builder_build(np.float64, order, gain, limit, period, S.Out("f32"))

    inf5 = [math.inf] * 5
    d = e["i_gain"]  # pid.rs:104-112
    ba = build(S.ORDER_I, _gl({"I": d["ki"]}, 0.0), inf5, d["tau"])
    y0 = _df1_f32(ba, [d["x0"]])[0]
    assert abs(np.float32(y0) / np.float32(d["x0"] * d["tau"] * d["ki"]) - np.float32(1)) < d["tol_eps"] * EPS32
    d = e["i_limit"]  # pid.rs:144-158
    ba = build(S.ORDER_I, _gl({"I": d["ki"]}, 0.0), _gl({"I": d["limit"]}, math.inf), d["period"])
    y = _df1_f32(ba, [d["x0"]] * (d["n"] + 1))[-1]
    assert abs(float(y) / (d["x0"] * d["limit"]) - 1.0) < d["tol"]
    d = e["order_p"]  # pid.rs:251-255
    ba = build(S.ORDER_P, _gl({"P": d["kp"]}, 0.0), inf5, d["period"])
    assert [float(v) for v in ba] == d["want"]
    d = e["units"]  # pid.rs:606-619
    ba = build(S.ORDER_I, _gl({"I": d["ki"]}, 0.0), inf5, d["tau"])
    for i, y in enumerate(_df1_f32(ba, [1.0] * d["n"]), start=1):
        want = np.float32(i) * np.float32(d["tau"]) * np.float32(d["ki"])
        assert abs(np.float32(y) / want - np.float32(1)) < d["tol_eps"] * EPS32


def test_reference_config_and_offset_kats(fn):
    e = KAT["biquad_config"]  # config.rs:188-200
    c = _abi.BaConfig()
    c.ba[:] = [0, 0, 0, 1, 0, 0]
    c.offset, c.min, c.max, c.f32 = 0.0, e["min"], e["max"], 1
    rc, _ = p_clamp(fn, "config_ba_build", "f32", c, (1, 1, 1), validate=1)
    assert rc == _abi.IDSP_EINVERTED and err(fn) == e["error"]
    with pytest.raises(S.BuildError, match="output_limits"):
        S.config_ba_build(np.float32, [0, 0, 0, 1, 0, 0], 0.0, e["min"], e["max"], (1, 1, 1), S.Out("f32"), validate=True)
    # unchecked build of the same config still produces a record, like `BiquadConfig::build`
    rc, (ba, u, mn, mx) = p_clamp(fn, "config_ba_build", "f32", c, (1, 1, 1), validate=0)
    assert rc == 0 and (mn, mx) == (1.0, 0.0)
    # biquad.rs:236-255: proportional(3) with setpoint -2 -> u = 6 (set_input_offset through Pid::build)
    o = KAT["biquad_clamp_offset"]
    p = _abi.Pid()
    p.builder = p_builder(S.ORDER_P, _gl({"P": o["k"]}, 0.0), [math.inf] * 5, False)
    p.setpoint, p.min, p.max = -o["input_offset"], -math.inf, math.inf
    rc, (ba, u, mn, mx) = p_clamp(fn, "pid_build_clamp", "f64", p, (1, 1, 1), validate=1)
    assert rc == 0 and ba == [o["k"], 0, 0, 0, 0] and u == o["u"] and (mn, mx) == (-math.inf, math.inf)


# ------------------------------------------------------------------ product == restatement
def _rand_filter(rng):
    kind = int(rng.integers(0, 3))
    shape = float({0: rng.uniform(0.2, 12.0), 1: rng.uniform(0.1, 4.0), 2: rng.uniform(0.3, 1.0)}[kind])
    return (float(rng.uniform(1e-4, math.pi)), float(10 ** rng.uniform(-2, 2)), float(10 ** rng.uniform(-1.5, 1.5)), kind, shape)


@pytest.mark.parametrize("typ", range(9), ids=_abi.FILTER_TYPES)
def test_filter_product_equals_restatement(fn, typ):
    rng = np.random.default_rng(100 + typ)
    for _ in range(300):
        w0, gain, shelf, kind, shape = _rand_filter(rng)
        rc, ba = p_filter(fn, typ, w0, gain, shelf, kind, shape, validate=1)
        assert rc == 0, err(fn)
        want = S.filter_build(np.float64, typ, w0, gain, shelf, kind, shape)
        assert all(same("f64", h, w) for h, w in zip(ba, want)), (typ, ba, want)  # f64: bit-identical
        rc, ba = p_filter(fn, typ, w0, gain, shelf, kind, shape, f32=True)
        want = S.filter_build(np.float32, typ, w0, gain, shelf, kind, shape)
        # f32: sinf/cosf/sinhf vs rounded f64 libm may differ in the last place and cancel in 1 - cos
        assert rc == 0 and close32(ba, want, 1e-5), (typ, ba, want)


def _rand_pid(rng):
    order = int(rng.integers(0, 3))
    sign = float(rng.choice([-1.0, 1.0]))
    gain = [sign * float(10 ** rng.uniform(-6, 3)) if rng.integers(0, 3) else 0.0 for _ in range(5)]
    limit = [sign * float(10 ** rng.uniform(-2, 4)) if rng.integers(0, 2) else math.inf for _ in range(5)]
    return order, gain, limit, float(10 ** rng.uniform(-4, 1))


@pytest.mark.parametrize("f32", [False, True], ids=["T=f64", "T=f32"])
@pytest.mark.parametrize("kind", ["i32", "f32", "f64"])
def test_pid_builder_product_equals_restatement(fn, kind, f32):
    T = np.float32 if f32 else np.float64
    rng = np.random.default_rng(7 + 2 * ["i32", "f32", "f64"].index(kind) + int(f32))
    for _ in range(400):
        order, gain, limit, period = _rand_pid(rng)
        frac = int(rng.integers(8, 31))
        rc, ba = p_pid(fn, kind, order, gain, limit, period, f32=f32, frac=frac, validate=1)
        assert rc == 0, err(fn)
        want = S.builder_build(T, order, gain, limit, period, S.Out(kind, frac))
        assert all(same(kind, h, w) for h, w in zip(ba, want)), (kind, order, gain, limit, period, ba, want)


@pytest.mark.parametrize("f32", [False, True], ids=["T=f64", "T=f32"])
@pytest.mark.parametrize("kind", ["i32", "f32", "f64"])
def test_pid_clamp_product_equals_restatement(fn, kind, f32):
    T = np.float32 if f32 else np.float64
    rng = np.random.default_rng(31 + 2 * ["i32", "f32", "f64"].index(kind) + int(f32))
    for _ in range(300):
        order, gain, limit, t = _rand_pid(rng)
        if rng.integers(0, 8) == 0:
            limit[int(rng.integers(0, 5))] = math.nan  # "json null": treated as +inf by Pid::build only
        units = (t, float(10 ** rng.uniform(-3, 3)), float(10 ** rng.uniform(-3, 3)))
        setpoint = float(rng.standard_normal() * 100)
        mn, mx = sorted((rng.standard_normal(2) * 1e4).tolist())
        frac = int(rng.integers(8, 31))
        p = _abi.Pid()
        p.builder = p_builder(order, gain, limit, f32)
        p.setpoint, p.min, p.max = setpoint, mn, mx
        rc, (ba, u, lo, hi) = p_clamp(fn, "pid_build_clamp", kind, p, units, frac=frac)
        wba, wu, wlo, whi = S.pid_build_clamp(T, order, gain, limit, setpoint, mn, mx, units, S.Out(kind, frac))
        assert rc == 0
        assert all(same(kind, h, w) for h, w in zip(ba + [u, lo, hi], wba + [wu, wlo, whi])), (kind, ba, wba, u, wu)


@pytest.mark.parametrize("f32", [False, True], ids=["T=f64", "T=f32"])
@pytest.mark.parametrize("kind", ["i32", "f32", "f64"])
def test_config_ba_and_filter_product_equals_restatement(fn, kind, f32):
    T = np.float32 if f32 else np.float64
    rng = np.random.default_rng(57 + 2 * ["i32", "f32", "f64"].index(kind) + int(f32))
    for _ in range(300):
        units = (float(10 ** rng.uniform(-3, 0)), float(10 ** rng.uniform(-2, 2)), float(10 ** rng.uniform(-2, 2)))
        offset = float(rng.standard_normal() * 10)
        mn, mx = sorted((rng.standard_normal(2) * 1e3).tolist())
        frac = int(rng.integers(8, 31))
        ba6 = (rng.standard_normal(6) * [1, 2, 1, 1, 2, 1]).tolist()
        ba6[3] = float(rng.uniform(0.5, 2.0))
        c = _abi.BaConfig()
        c.ba[:] = ba6
        c.offset, c.min, c.max, c.f32 = offset, mn, mx, int(f32)
        rc, (ba, u, lo, hi) = p_clamp(fn, "config_ba_build", kind, c, units, frac=frac, validate=1)
        wba, wu, wlo, whi = S.config_ba_build(T, ba6, offset, mn, mx, units, S.Out(kind, frac), validate=True)
        assert rc == 0, err(fn)
        assert all(same(kind, h, w) for h, w in zip(ba + [u, lo, hi], wba + [wu, wlo, whi]))
        # Filter arm: f64 bit-identical; f32 within the libm caveat
        typ = int(rng.integers(0, 9))
        skind = int(rng.integers(0, 3))
        shape = float({0: rng.uniform(0.3, 8.0), 1: rng.uniform(0.2, 3.0), 2: rng.uniform(0.3, 1.0)}[skind])
        freq = float(rng.uniform(1e-3, 0.49)) / units[0]
        gdb, sdb = float(rng.uniform(-30, 30)), float(rng.uniform(-20, 20))
        fc = _abi.FilterConfig(typ, skind, freq, gdb, sdb, shape, offset, mn, mx, int(f32))
        rc, (ba, u, lo, hi) = p_clamp(fn, "config_filter_build", kind, fc, units, frac=frac, validate=1)
        wba, wu, wlo, whi = S.config_filter_build(T, typ, freq, gdb, sdb, skind, shape, offset, mn, mx, units,
                                                  S.Out(kind, frac), validate=True)
        assert rc == 0, err(fn)
        if not f32:
            assert all(same(kind, h, w) for h, w in zip(ba + [u, lo, hi], wba + [wu, wlo, whi])), (typ, ba, wba)
        else:
            assert close32(ba, wba, 2e-5, float(1 << frac) if kind == "i32" else 1.0), (typ, ba, wba)
            assert all(same(kind, h, w) for h, w in zip([u, lo, hi], [wu, wlo, whi]))


# ------------------------------------------------------------------ validation paths
def test_filter_validation_errors_match_reference_text(fn):
    nan, inf = math.nan, math.inf
    cases = [  # (w0, gain, shelf, kind, shape) -> status
        ((nan, 1, 1, 0, 1), _abi.IDSP_ENONFINITE), ((inf, 1, 1, 0, 1), _abi.IDSP_ENONFINITE),
        ((-0.1, 1, 1, 0, 1), _abi.IDSP_EOUTOFRANGE), ((3.2, 1, 1, 0, 1), _abi.IDSP_EOUTOFRANGE),
        ((1, 0, 1, 0, 1), _abi.IDSP_ENONPOSITIVE), ((1, nan, 1, 0, 1), _abi.IDSP_ENONPOSITIVE),
        ((1, 1, -1, 0, 1), _abi.IDSP_ENONPOSITIVE), ((1, 1, inf, 0, 1), _abi.IDSP_ENONPOSITIVE),
        ((1, 1, 1, 0, nan), _abi.IDSP_ENONFINITE), ((1, 1, 1, 0, 0), _abi.IDSP_ENONPOSITIVE),
        ((1, 1, 1, 1, inf), _abi.IDSP_ENONFINITE), ((1, 1, 1, 1, -2.0), 0),
        ((1, 1, 1, 2, nan), _abi.IDSP_ENONFINITE), ((1, 1, 1, 2, -1), _abi.IDSP_ENONPOSITIVE),
    ]
    for (w0, gain, shelf, kind, shape), status in cases:
        for f32 in (False, True):
            rc, _ = p_filter(fn, S.LOWPASS, w0, gain, shelf, kind, shape, f32=f32, validate=1)
            assert rc == status, (w0, gain, shelf, kind, shape, err(fn))
            try:
                S.filter_validate(np.float32 if f32 else np.float64, w0, gain, shelf, kind, shape)
                assert status == 0
            except S.BuildError as e:
                assert status != 0 and err(fn) == e.text
            # the unchecked build never refuses (coefficients may be NaN/inf like the reference)
            assert p_filter(fn, S.LOWPASS, w0, gain, shelf, kind, shape, f32=f32, validate=0)[0] == 0
    assert p_filter(fn, 9, 1.0)[0] == _abi.IDSP_EINVAL and p_filter(fn, 0, 1.0, kind=3)[0] == _abi.IDSP_EINVAL


def test_pid_validation_errors_match_reference_text(fn):
    nan, inf = math.nan, math.inf
    z5, i5 = [0.0] * 5, [inf] * 5

    def v(order, gain, limit, period, **kw):
        rc, _ = p_pid(fn, "f64", order, gain, limit, period, validate=1)
        try:
            S.builder_validate(np.float64, order, gain, limit, period)
            assert rc == 0
        except S.BuildError as e:
            assert rc < 0 and err(fn) == e.text, (rc, err(fn), e.text)
        return rc

    assert v(1, z5, i5, 1.0) == 0
    assert v(1, z5, i5, nan) == _abi.IDSP_ENONFINITE and v(1, z5, i5, inf) == _abi.IDSP_ENONFINITE
    assert v(1, z5, i5, 0.0) == _abi.IDSP_ENONPOSITIVE and v(1, z5, i5, -1.0) == _abi.IDSP_ENONPOSITIVE
    assert v(1, [0, nan, 0, 0, 0], i5, 1.0) == _abi.IDSP_ENONFINITE
    assert v(1, z5, [inf, nan, inf, inf, inf], 1.0) == _abi.IDSP_ENONFINITE
    assert v(1, [0, inf, 0, 0, 0], i5, 1.0) == 0  # infinite gains pass validation (pid.rs:204-208)
    assert v(1, z5, [inf, 0.0, inf, inf, inf], 1.0) == _abi.IDSP_ENONPOSITIVE
    assert v(1, [0, 1.0, 0, 0, 0], [inf, -5.0, inf, inf, inf], 1.0) == _abi.IDSP_ESIGN
    assert v(1, [0, -1.0, 0, 0, 0], [inf, -5.0, inf, inf, inf], 1.0) == 0
    assert v(1, [0, 0, 1.0, 0, 0], [inf, inf, -5.0, inf, inf], 1.0) == 0  # P limit is ignored
    assert p_pid(fn, "f64", 3, z5, i5, 1.0)[0] == _abi.IDSP_EINVAL
    assert p_pid(fn, "i32", 1, z5, i5, 1.0, frac=32)[0] == _abi.IDSP_EINVAL
    # Pid::validate (pid.rs:497-518) and the config checks (config.rs:309-344)
    p = _abi.Pid()
    p.builder = p_builder(1, z5, i5, False)
    p.setpoint, p.min, p.max = 0.0, 1.0, -1.0
    assert p_clamp(fn, "pid_build_clamp", "f64", p, (1, 1, 1), validate=1)[0] == _abi.IDSP_EINVERTED
    p.min, p.max = -1.0, 1.0
    for units, status, text in [((0, 1, 1), _abi.IDSP_ENONPOSITIVE, "parameter `t` must be positive"),
                                ((1, nan, 1), _abi.IDSP_ENONFINITE, "parameter `x` must be finite"),
                                ((1, 1, -2), _abi.IDSP_ENONPOSITIVE, "parameter `y` must be positive")]:
        assert p_clamp(fn, "pid_build_clamp", "f64", p, units, validate=1)[0] == status and err(fn) == text
    c = _abi.BaConfig()
    c.ba[:] = [1, 0, 0, 1, 0, 0]
    c.offset, c.min, c.max = nan, -1.0, 1.0
    assert p_clamp(fn, "config_ba_build", "f64", c, (1, 1, 1), validate=1)[0] == _abi.IDSP_ENONFINITE
    assert err(fn) == "parameter `offset` must be finite"
    c.offset = 0.0
    assert p_clamp(fn, "config_ba_build", "f64", c, (0, 1, 1), validate=1)[0] == 0  # Ba does not check t (config.rs:391)
    c.ba[4] = inf
    assert p_clamp(fn, "config_ba_build", "f64", c, (1, 1, 1), validate=1)[0] == _abi.IDSP_ENONFINITE and err(fn) == "parameter `ba` must be finite"
    fc = _abi.FilterConfig(0, 0, 0.6, 0.0, 0.0, 0.7, 0.0, -1.0, 1.0, 0)
    assert p_clamp(fn, "config_filter_build", "f64", fc, (1, 1, 1), validate=1)[0] == _abi.IDSP_EOUTOFRANGE  # f0 > 0.5
    assert p_clamp(fn, "config_filter_build", "f64", fc, (0, 1, 1), validate=1)[0] == _abi.IDSP_ENONPOSITIVE  # t checked here
