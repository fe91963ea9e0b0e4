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
