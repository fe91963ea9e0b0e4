"""Host-side mirror (idsp_amd.process): argument checking that needs no GPU
(the reference's debug_assert / panic conditions surface as ValueError)."""
import pytest
import torch

import idsp_amd as ia


def test_view_from_flat_length_check():
    flat = torch.zeros(6, dtype=torch.int32)
    v = ia.View(flat, ia.LaneMajor, 2)
    assert v.frames == 3
    with pytest.raises(ValueError):  # view.rs:182 assert_eq!(flat.len(), frames * L)
        ia.View(flat, ia.LaneMajor, 4)
    with pytest.raises(ValueError):
        ia.View(flat, ia.LaneMajor, 2, frames=4)


def test_biquad_constructors_follow_reference_constants():
    b = ia.Biquad.identity(frac=30)
    assert b.ba == [1 << 30, 0, 0, 0, 0] and b.frac == 30
    assert ia.Biquad.hold().ba == [0.0, 0.0, 0.0, 1.0, 0.0]
    assert ia.Biquad.proportional(3.0).forward_gain() == 3.0  # biquad.rs:219-226
    c = ia.BiquadClamp(ia.Biquad.identity())
    assert c.u == 0.0 and c.min == float("-inf") and c.max == float("inf")  # num.rs:33-52
    ci = ia.BiquadClamp(ia.Biquad.identity(frac=29))
    assert (ci.u, ci.min, ci.max) == (0, -(1 << 31), (1 << 31) - 1)


def test_from_sos_quantisation_matches_reference_kat():
    import math

    w0 = math.tau * 0.1
    alpha = 0.5 * math.sin(w0) * math.sqrt(2.0)
    b = 1000.0 * 0.5 * (1.0 - math.cos(w0))
    q = ia.Biquad.from_sos([b, 2 * b, b, 1 + alpha, -2 * math.cos(w0), 1 - alpha], frac=30)
    # saturated feed-forward taps, src/iir/coefficients.rs:289-300
    assert q.ba == [2147483647, 2147483647, 2147483647, 1227265970, -443242341]


def test_cpu_tensors_are_rejected_not_silently_processed():
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(ValueError):
        ia.Split(ia.Biquad.identity(frac=30), ia.DirectForm1).lanes(4, device="cpu")
    with pytest.raises(ValueError):
        ia.cossin(torch.zeros(4, dtype=torch.int32))


def test_lowpass_order_limits():
    with pytest.raises(NotImplementedError):  # lowpass.rs:75 unimplemented!()
        ia.Lowpass([1, 2, 3])


# ---- coefficient front-end mirror (idsp_amd.coefficients): host-only, written like the reference's doctests
def test_filter_builder_doctests():
    """src/iir/coefficients.rs:289-300,316-326 (`Filter::default().critical_frequency(0.1).gain(1000.0)`)."""
    from idsp_amd import coefficients as co

    lp = co.Filter().critical_frequency(0.1).gain(1000.0).build_biquad(co.Type.Lowpass, frac=30)
    assert lp.ba == [2147483647, 2147483647, 2147483647, 1227265970, -443242341] and lp.frac == 30
    ba = co.Filter().frequency(1000.0, 48e3).q(5.0).gain_db(3.0).bandpass()  # :318-327 (prints only)
    assert ba[0][1] == 0.0 and ba[0][0] == -ba[0][2] and ba[1][0] > 1.0
    ba = co.Filter().frequency(1000.0, 48e3).shelf_slope(2.0).shelf_db(20.0).lowshelf()  # :385-394
    # slope 2 with a 20 dB shelf has no real solution: qi = sqrt(negative) -> the doctest prints NaNs
    assert all(v != v for v in (ba[0][0], ba[0][2], ba[1][0], ba[1][2])) and ba[0][1] == ba[0][1]
    ba = co.Filter().frequency(1000.0, 48e3).shelf_slope(1.0).shelf_db(20.0).lowshelf()
    assert all(v == v for v in ba[0] + ba[1]) and ba[0][0] > ba[1][0]
    with pytest.raises(ia.IdspError, match="parameter `frequency` is out of range"):
        co.Filter().critical_frequency(0.6).try_build(co.Type.Lowpass)
    assert co.Filter().critical_frequency(0.6).build(co.Type.Lowpass)  # unchecked build never refuses


def test_pid_builder_doctests():
    """src/iir/pid.rs:28-38,251-255,574-590."""
    from idsp_amd.coefficients import Action, Builder, Order

    b = (Builder().gain(Action.I, 1e-3).gain(Action.P, 1.0).gain(Action.D, 1e2)
         .limit(Action.I, 1e3).limit(Action.D, 1e1).build(1.0))
    want = [9.181909, -18.272726, 9.090908, 1.9090908, -0.9090908]
    assert all(abs(h / w - 1.0) < 2 * 1.1920929e-07 for h, w in zip(b.ba, want))
    i = Builder().gain(Action.P, 3.0).order(Order.P).build(1.0)
    assert i.ba == ia.Biquad.proportional(3.0).ba
    q = Builder(f32=True).ki(1e-5).kp(1e-2).kd(1e0).limit_i(1e1).limit_d(1e-1).build(1.0, frac=29)  # pid.rs:592-603
    assert q.frac == 29 and all(isinstance(v, int) for v in q.ba)
    with pytest.raises(ia.IdspError, match="incompatible sign"):
        Builder().ki(1.0).limit_i(-1.0).try_build(1.0)


def test_biquad_config_variants():
    """src/iir/config.rs:176-207 and the four build arms (:355-387)."""
    from idsp_amd import coefficients as co

    cfg = co.BiquadConfig.from_tag("Filter")
    assert cfg.as_ref() == "Filter"
    with pytest.raises(ValueError):
        co.BiquadConfig.from_tag("Unknown")
    with pytest.raises(ia.IdspError, match="range `output_limits` is inverted"):
        co.BiquadConfig.Ba(co.BaConfig(min=1.0, max=0.0, f32=True)).try_build(co.Units())
    raw = ia.BiquadClamp(ia.Biquad.identity())
    assert co.BiquadConfig.Raw(raw).try_build(co.Units(0.0, 0.0, 0.0)) is raw  # Raw never validates units
    pid = co.Pid().kp(-2.0).ki(-10.0).limit_i(-50.0).setpoint(0.5).output_limits(-1.0, 1.0)
    c = co.BiquadConfig.Pid(pid).try_build(co.Units(t=1e-3, x=2.0, y=4.0), f64=True)
    assert (c.min, c.max) == (-0.25, 0.25) and c.coeff.ba[3] > 0.99 and c.u == -0.25 * c.coeff.forward_gain()
    f = co.BiquadConfig.Filter(co.FilterConfig(co.Type.Notch, frequency=50.0, shape=co.Shape.Q(10.0), offset=1.0))
    c = f.try_build(co.Units(t=1e-3), frac=30)
    assert c.u == 1 and c.coeff.ba[0] == c.coeff.ba[2] and c.coeff.ba[1] == -c.coeff.ba[3]
