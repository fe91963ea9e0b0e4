"""The host-side mirror driving the HIP engine, written like the reference's
own doctests/tests (names and call shapes of dsp-process / idsp)."""
import math

import numpy as np
import pytest
import torch

import idsp_amd as ia

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(gpu):
    return gpu


def dev(a, dtype):
    return torch.tensor(a, dtype=dtype, device="cuda")


def test_lowpass_doctest_inplace():
    """src/iir/coefficients.rs:289-300 through `Split::new(iir, DirectForm1).lanes(1)`."""
    w0 = math.tau * 0.1
    alpha = 0.5 * math.sin(w0) * math.sqrt(2.0)
    b = 1000.0 * 0.5 * (1.0 - math.cos(w0))
    iir = ia.Biquad.from_sos([b, 2 * b, b, 1 + alpha, -2 * math.cos(w0), 1 - alpha], frac=30)
    xy = dev([[3], [-4], [5], [7], [-3], [2]], torch.int32)
    ia.Split(iir, ia.DirectForm1).lanes(1).inplace(xy)
    assert xy.flatten().tolist() == [5, 3, 9, 25, 42, 49]


def test_lane_major_lanes_view():
    """dsp-process/src/lib.rs:146-154 with a per-sample offset stage."""
    p = ia.Split(ia.BiquadClamp(ia.Biquad.identity(), u=3.0), ia.DirectForm1).lanes(2)
    x = dev([1, 2, 3, 10, 20, 30], torch.float32)
    y = torch.zeros(6, dtype=torch.float32, device="cuda")
    p.process_view(ia.View(x, ia.LaneMajor, 2), ia.ViewMut(y, ia.LaneMajor, 2))
    assert y.tolist() == [4, 5, 6, 13, 23, 33]
    with pytest.raises(ValueError):  # frames mismatch: debug_assert_eq!(x.frames(), y.frames())
        p.process_view(ia.View(x, ia.LaneMajor, 2), ia.ViewMut(y[:4], ia.LaneMajor, 2))


def test_frame_major_block_and_state_continuity():
    b = ia.Biquad.from_sos([0.2, 0.3, 0.1, 1.0, -0.5, 0.2])
    x = torch.randn(100, 7, device="cuda")
    whole = ia.Split(b, ia.DirectForm2Transposed).lanes(7)
    y = torch.empty_like(x)
    whole.block(x, y)
    halves = ia.Split(b, ia.DirectForm2Transposed).lanes(7)
    y2 = torch.empty_like(x)
    halves.block(x[:37].contiguous(), y2[:37])
    halves.block(x[37:].contiguous(), y2[37:])
    assert torch.equal(y, y2) and torch.equal(whole.state, halves.state)


def test_cascade_equals_serial_sections():
    """src/iir/biquad.rs:685-699."""
    stage = ia.Biquad.from_sos([0.5, 0.25, 0.125, 1.0, -0.1, 0.02], f32_math=True)
    x = dev([[-0.75], [0.5], [0.0], [0.25], [-0.125], [1.0], [-0.5], [0.375]], torch.float32)
    yc, yr = torch.empty_like(x), torch.empty_like(x)
    ia.Split(ia.Cascade([stage] * 3)).lanes(1).block(x, yc)
    ia.Split([stage] * 3, ia.DirectForm1).lanes(1).block(x, yr)
    assert torch.equal(yc, yr)


def test_hbf_dec_kat_and_cascade_gain():
    """src/hbf.rs:548-555, then DC gain 2 per stage on HBF_DEC_CASCADE /16."""
    h = ia.HbfDecCascade(taps=[[0.5]]).lanes(1)
    y = torch.zeros(4, device="cuda")
    h.process_view(ia.View(torch.ones(8, device="cuda"), ia.LaneMajor, 1, width=2), ia.ViewMut(y, ia.LaneMajor, 1))
    assert y.tolist() == [1.5, 2.0, 2.0, 2.0]
    d = ia.HbfDecCascade(4).lanes(3)
    assert d.response_length() == 57 and d.state.shape == (118, 3)
    x = torch.ones(3 * 200 * 16, device="cuda")
    y = torch.zeros(3 * 200, device="cuda")
    d.process_view(ia.View(x, ia.LaneMajor, 3, width=16), ia.ViewMut(y, ia.LaneMajor, 3))
    assert abs(y[-1].item() - 16.0) < 1e-4


def test_cossin_pyfunction_shape():
    """src/py.rs:10-28: i32[N] -> i32[N, 2]."""
    out = ia.cossin(dev([0, 1 << 30, -(1 << 30)], torch.int32))
    assert out.shape == (3, 2)
    assert out[0].tolist() == [2147454703, -1898]
    assert abs(out[1, 0].item()) < 1 << 17 and out[1, 1].item() > (1 << 31) - (1 << 16)


def test_atan2_pyfunction_shape():
    """src/py.rs:30-47: i32[N, 2] rows [x, y] -> i32[N]; src/atan2.rs:177-183 values."""
    out = ia.atan2(dev([[1, 0], [0, 1], [-1, 0], [0, -1]], torch.int32))
    assert out.tolist() == [0, 0x3FFFFFFF, 0x7FFFFFFF, -0x40000000]
    with pytest.raises(ValueError):
        ia.atan2(dev([1, 2, 3], torch.int32))


def test_sos_pyfunction_matches_oracle():
    """src/py.rs:49-73 `sos(sos, xy)` (Q29, slice composition) on one stream."""
    import ctypes as C

    from idsp_amd import _abi
    from tests import _harness as H

    rows = [[0.1, 0.2, 0.1, 1.0, -1.2, 0.5], [0.3, -0.1, 0.05, 1.0, -0.3, 0.1]]
    rng = np.random.default_rng(0)
    x = rng.integers(-(1 << 26), 1 << 26, size=1000, dtype=np.int32)
    xy = torch.from_numpy(x.copy()).cuda()
    ia.sos(rows, xy)
    o = H.oracle()
    cfg = (_abi.BiquadI32 * 2)()
    for c, r in zip(cfg, rows):
        o.fn["biquad_i32_from_sos"]((C.c_double * 6)(*r), 29, C.byref(c))
    st = np.zeros((8, 1), np.uint32)
    y = x.copy()
    assert o.stream("biquad_i32_df1", cfg, 2, st, y, y, 1, 1000, H.LM) == 0
    assert np.array_equal(xy.cpu().numpy(), y)


def test_lockin_recovers_dc_iq():
    """Shape of examples/ddc_lockin.rs:100-110 on the integer lock-in: a tone at the
    LO frequency demodulates to a DC I/Q vector; the Q32<32> mixer halves the
    full-scale LO (cossin amplitude 2^31 read as Q32<32> = 0.5), so |IQ| = A/4."""
    n, lanes = 16384, 4
    f = 0.173
    step = int(round(f * (1 << 32)))
    phi = 0.37
    t = np.arange(1, n + 1)
    amp = 1 << 28
    x = np.round(amp * np.cos(2 * np.pi * ((t * step) % (1 << 32)) / (1 << 32) - phi)).astype(np.int32)
    k = math.pi * (1 << 31) * 0.004
    lp = ia.Lowpass([int(k * k / (1 << 32)), -int(k * math.sqrt(2.0))])
    p = ia.Lockin([lp, lp]).lanes(lanes, step=step)
    xd = torch.from_numpy(np.repeat(x[:, None], lanes, axis=1).copy()).cuda()
    y = torch.empty((n, lanes, 2), dtype=torch.int32, device="cuda")
    p.block(xd, y)
    iq = y[12288:].double().mean(dim=0).cpu().numpy() / amp
    assert np.allclose(iq[:, 0], 0.25 * math.cos(phi), atol=3e-3)
    assert np.allclose(iq[:, 1], 0.25 * math.sin(phi), atol=3e-3)
    # polar read-out fused into the same pass: phase = phi (1 << 31 == pi), |IQ|^2 = (A/4)^2
    for layout_lm in (False, True):
        pa = ia.Lockin([lp, lp], output="arg").lanes(lanes, step=step)
        pp = ia.Lockin([lp, lp], output="norm_sqr").lanes(lanes, step=step)
        if layout_lm:
            xl = xd.t().contiguous()
            arg = torch.empty((lanes, n), dtype=torch.int32, device="cuda")
            pw = torch.empty((lanes, n), dtype=torch.int64, device="cuda")
            pa.process_view(ia.View(xl, ia.LaneMajor, lanes), ia.ViewMut(arg, ia.LaneMajor, lanes))
            pp.process_view(ia.View(xl, ia.LaneMajor, lanes), ia.ViewMut(pw, ia.LaneMajor, lanes))
            arg, pw = arg.t(), pw.t()
        else:
            arg = torch.empty((n, lanes), dtype=torch.int32, device="cuda")
            pw = torch.empty((n, lanes), dtype=torch.int64, device="cuda")
            pa.block(xd, arg)
            pp.block(xd, pw)
        assert torch.equal(pa.state, p.state) and torch.equal(pp.state, p.state)
        z = y.to(torch.int64)
        assert torch.equal(pw, z[..., 0] * z[..., 0] + z[..., 1] * z[..., 1])
        if layout_lm:
            assert torch.equal(arg, arg_fm)  # LaneMajor multi-wave kernel == FrameMajor one
        arg_fm = arg
        ph = arg[12288:].double().mean(dim=0).cpu().numpy() * math.pi / (1 << 31)
        assert np.allclose(ph, phi, atol=2e-2)


def test_f64_biquad_and_fir_through_the_mirror():
    """`Biquad<f64>` lanes and a same-rate `EvenSymmetric` FIR."""
    b = ia.Biquad.from_sos([0.2, 0.3, 0.1, 1.0, -0.5, 0.2], f64=True)
    x = torch.randn(50, 3, device="cuda", dtype=torch.float64)
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    ia.Split(b, ia.DirectForm1).lanes(3).block(x, y1)
    ia.Split(b, ia.DirectForm2Transposed).lanes(3).block(x, y2)
    assert torch.allclose(y1, y2, atol=1e-12)
    xi = torch.zeros(10, 1, device="cuda")
    xi[0] = 1.0
    yi = torch.empty_like(xi)
    ia.FirSym("EvenSymmetric", [0.25, 0.5]).lanes(1).block(xi, yi)
    assert yi.flatten().tolist()[:5] == [0.25, 0.5, 0.5, 0.25, 0.0]  # taps mirror around the window centre


def test_by_lane_filter_bank():
    """`Split::new(ByLane([c0, c1, ..]), states)` (dsp-process/src/compose.rs:363-390): a bank of
    different lowpasses, one per lane, equals each filter run alone; `set_lane` swaps one in place."""
    from idsp_amd import coefficients as co

    f0s = [0.01, 0.05, 0.1, 0.2, 0.3]
    bank = [co.Filter().critical_frequency(f0).build_biquad(co.Type.Lowpass, frac=30) for f0 in f0s]
    x = torch.randint(-(1 << 20), 1 << 20, (300, len(bank)), dtype=torch.int32, device="cuda")
    y = torch.empty_like(x)
    bl = ia.ByLane(bank, ia.DirectForm1)
    bl.block(x, y)
    for i, b in enumerate(bank):
        yi = torch.empty((300, 1), dtype=torch.int32, device="cuda")
        ia.Split(b, ia.DirectForm1).lanes(1).block(x[:, i:i + 1].contiguous(), yi)
        assert torch.equal(y[:, i], yi[:, 0])
    bl.reset()
    bl.set_lane(2, bank[0])
    bl.block(x, y)
    y0 = torch.empty((300, 1), dtype=torch.int32, device="cuda")
    ia.Split(bank[0], ia.DirectForm1).lanes(1).block(x[:, 2:3].contiguous(), y0)
    assert torch.equal(y[:, 2], y0[:, 0])
    # a PID bank with clamps: per-lane BiquadClamp<f32> on DirectForm2Transposed, lane-major views
    pids = [co.Pid().kp(1.0 + k).ki(10.0 * k).output_limits(-0.5, 0.5).build(co.Units(t=1e-3)) for k in range(4)]
    pl = ia.ByLane(pids, ia.DirectForm2Transposed)
    xs = torch.randn(4 * 64, device="cuda")
    ys = torch.empty_like(xs)
    pl.process_view(ia.View(xs, ia.LaneMajor, 4), ia.ViewMut(ys, ia.LaneMajor, 4))
    assert ys.abs().max().item() <= 0.5
    for k, c in enumerate(pids):
        one = torch.empty(64, device="cuda")
        ia.Split(c, ia.DirectForm2Transposed).lanes(1).process_view(
            ia.View(xs[64 * k:64 * (k + 1)].contiguous(), ia.LaneMajor, 1), ia.ViewMut(one, ia.LaneMajor, 1))
        assert torch.equal(ys[64 * k:64 * (k + 1)], one)
    with pytest.raises(ValueError):
        ia.ByLane([bank[0], pids[0]], ia.DirectForm1)


def test_cic_reference_properties():
    """src/cic.rs:223-240 (rate 0 identity), :286-306 (unit rate), :242-262 (step response)."""
    id_ = ia.Cic(3, 0).decimate().lanes(2)
    x = torch.tensor([[5, -7], [1 << 40, -(1 << 50)], [3, 4]], dtype=torch.int64, device="cuda")
    y = torch.zeros(3, 2, dtype=torch.int64, device="cuda")
    id_.block(x.reshape(3, 2, 1), y)
    assert torch.equal(x, y)
    unit = ia.Cic(3, 0, comb_delay=3)
    assert unit.gain_log2() == 6 and unit.gain() == 27 and unit.order() == 3 and unit.comb_delay() == 3
    up = ia.Cic(3, 3, dtype=torch.int32)
    assert up.gain() == 64 and up.response_length() == 9
    p = up.interpolate().lanes(1)
    xs = torch.full((8,), 10, dtype=torch.int32, device="cuda")
    ys = torch.zeros(32, dtype=torch.int32, device="cuda")
    p.process_view(ia.View(xs, ia.LaneMajor, 1), ia.ViewMut(ys, ia.LaneMajor, 1, width=4))
    assert ys[9:].tolist() == [640] * 23 and ys[:9].tolist() == sorted(ys[:9].tolist())
    # decimating what was interpolated returns gain^2-scaled samples once settled
    d = ia.Cic(3, 3, dtype=torch.int32).decimate().lanes(1)
    yd = torch.zeros(8, dtype=torch.int32, device="cuda")
    d.process_view(ia.View(ys, ia.LaneMajor, 1, width=4), ia.ViewMut(yd, ia.LaneMajor, 1))
    assert yd[-1].item() == 10 * 64 * 64
    with pytest.raises(ValueError):
        ia.Cic(0, 3)


def test_normal_and_wdf_lanes():
    """`Normal::from` (src/iir/normal.rs:62-76) and `Wdf` (src/iir/wdf.rs) through the mirror."""
    nf = ia.Normal.from_ba([[0.2, 0.4, 0.2], [1.0, -1.2, 0.52]])
    assert abs(nf.ba[3] ** 2 + nf.ba[4] ** 2 - 0.52) < 1e-12
    with pytest.raises(ia.IdspError):
        ia.Normal.from_ba([[1, 0, 0], [1.0, -3.0, 1.0]])  # real poles: assert!(pq >= 0.0)
    p = nf.lanes(2)
    x = torch.zeros(50, 2, dtype=torch.float64, device="cuda")
    x[0] = 1.0
    y = torch.empty_like(x)
    p.block(x, y)
    assert torch.equal(y[:, 0], y[:, 1]) and y[1, 0].item() != 0.0 and abs(y[-1, 0].item()) < 1e-3
    delay = ia.Wdf.default(1, 0x1).lanes(1)
    xi = dev([[1], [2], [3], [4]], torch.int32)
    yi = torch.empty_like(xi)
    delay.block(xi, yi)
    assert yi.flatten().tolist() == [0, 1, 2, 3]
    assert ia.Wdf.quantize(0xA, [0.3]) is None and ia.Wdf.quantize(0xAD, [-0.9, 0.9]).cfg.n == 2
    chain = ia.Wdf.chain([ia.Wdf.quantize(0xAD, [-0.9, 0.9]), ia.Wdf.quantize(0xA, [0.8])]).lanes(3)
    assert chain.state.shape == (3, 3)
    xs = (torch.randn(4000, 3, device="cuda") * (1 << 20)).to(torch.int32)
    ys = torch.empty_like(xs)
    chain.block(xs, ys)
    ex, ey = (xs.double() ** 2).sum().item(), (ys.double() ** 2).sum().item()
    assert abs(ey / ex - 1.0) < 5e-3  # allpass


def test_fm_disc_receiver_core():
    """examples/fm_disc.rs: a constant-frequency carrier demodulates to (frequency - carrier) after the deemphasis settles."""
    from idsp_amd import coefficients as co

    carrier, offset, n = 0x19341234, 0x00200000, 600
    ph = (torch.arange(1, n + 1, dtype=torch.int64) * (carrier + offset)) & 0xFFFFFFFF
    ph = torch.where(ph >= (1 << 31), ph - (1 << 32), ph).to(torch.int32).cuda()
    x = ia.cossin(ph)  # [n, 2] = Complex<Q32<32>> bits
    deemph = co.Filter(f32=True).critical_frequency(0.02).build_biquad(co.Type.Lowpass, frac=30)
    rx = ia.FmDisc(carrier, deemph).lanes(1)
    y = torch.empty(n, dtype=torch.int32, device="cuda")
    rx.process_view(ia.View(x, ia.LaneMajor, 1, width=2), ia.ViewMut(y, ia.LaneMajor, 1))
    assert y[0].item() == 0 and abs(y[-1].item() / offset - 1.0) < 2e-3


def test_lockin_with_biquad_arms_and_external_lo_recover_dc_iq():
    """`Lockin<C>` beyond lowpass arms (src/lockin.rs:16-27) through the host mirror: the integer lock-in with a Q30 biquad
    arm on a phase accumulator, and the f32 `mix -> Biquad<f32>.lanes()` graph of examples/ddc_lockin.rs:35-42 with its
    own acceptance bounds (:100-111)."""
    n, lanes, f, phi = 16384, 4, 0.173, 0.37
    step = int(round(f * (1 << 32)))
    t = np.arange(1, n + 1)
    amp = 1 << 28
    x = np.round(amp * np.cos(2 * np.pi * ((t * step) % (1 << 32)) / (1 << 32) - phi)).astype(np.int32)
    w0 = math.tau * 0.002
    alpha = 0.5 * math.sin(w0) * math.sqrt(2.0)
    b = 0.5 * (1.0 - math.cos(w0))
    sos = [b, 2 * b, b, 1 + alpha, -2 * math.cos(w0), 1 - alpha]
    p = ia.Lockin([ia.Biquad.from_sos(sos, frac=30)]).lanes(lanes, step=step)
    xd = torch.from_numpy(np.repeat(x[:, None], lanes, axis=1).copy()).cuda()
    y = torch.empty((n, lanes, 2), dtype=torch.int32, device="cuda")
    p.block(xd, y)
    iq = y[12288:].double().mean(dim=0).cpu().numpy() / amp
    assert np.allclose(iq[:, 0], 0.25 * math.cos(phi), atol=3e-3) and np.allclose(iq[:, 1], 0.25 * math.sin(phi), atol=3e-3)
    # f32 graph with the LO as an input
    ph = (np.float32(math.tau * f) * np.arange(n, dtype=np.float32))
    xf = np.cos(ph + np.float32(phi)).astype(np.float32)
    lo = np.stack([np.cos(ph), -np.sin(ph)], axis=1).astype(np.float32)
    q = ia.LockinLo([ia.Biquad.from_sos(sos)], lanes)
    xfd = torch.from_numpy(np.repeat(xf[:, None], lanes, axis=1).copy()).cuda()
    lod = torch.from_numpy(np.repeat(lo[:, None, :], lanes, axis=1).copy()).cuda()
    yf = torch.empty((n, lanes, 2), dtype=torch.float32, device="cuda")
    q.process(xfd, lod, yf)
    m = yf[12288:].double().mean(dim=0).cpu().numpy()
    assert np.allclose(m[:, 0], 0.5 * math.cos(phi), atol=3e-3) and np.allclose(m[:, 1], 0.5 * math.sin(phi), atol=3e-3)
