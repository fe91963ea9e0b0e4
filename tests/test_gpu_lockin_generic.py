"""`Lockin<C>` with biquad arms / external LO (src/lockin.rs:16-39, include/idsp_hip.h `idsp_lockin_*_biquad*`, `*_lo_*`) on
HIP against the CPU oracle: bit-exact outputs and written-back state, both layouts, ragged lane and frame counts, two
consecutive calls on one state; and examples/ddc_lockin.rs:100-111 replayed through the f32 entry on the GPU."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import _harness as H
from tests import _lockin_generic_cases as G

pytestmark = pytest.mark.gpu
FM, LM = H.FM, H.LM
DEV = "cuda:0"
SHAPES = [(1, 1), (3, 5), (64, 33), (65, 128), (200, 257), (1024, 64), (4099, 19)]


def dev(a):
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).to(DEV)


def run_pair(name, cfg, n, st0, x, lo, ydtype, lanes, frames, layout, lo_form):
    """Run oracle and HIP on the same inputs for two consecutive calls; assert equal outputs and state."""
    o, e = H.oracle(), H.engine()
    so, sg = st0.copy(), dev(st0)
    for rep in range(2):
        yo = np.empty(lanes * frames * 2, ydtype)
        yg = torch.full((lanes * frames * 2,), -77, dtype=torch.float32 if ydtype == np.float32 else torch.int32, device=DEV)
        xd = dev(x[rep])
        if lo_form:
            lod = dev(lo[rep])
            rco = G.call_lo(o, name, cfg, n, so, x[rep], lo[rep], yo, lanes, frames, layout, False)
            rcg = G.call_lo(e, name, cfg, n, sg, xd, lod, yg, lanes, frames, layout, True)
        else:
            rco = o.stream(name, cfg, n, so, x[rep], yo, lanes, frames, layout)
            rcg = e.stream(name, cfg, n, sg, xd, yg, lanes, frames, layout)
        torch.cuda.synchronize()
        assert rco == 0 and rcg == 0, e.err()
        got = yg.cpu().numpy()
        assert np.array_equal(got.view(np.uint32), yo.view(np.uint32)), (name, lanes, frames, layout, rep, e.last_kernel())
        assert np.array_equal(sg.cpu().numpy().view(np.uint32), so), (name, lanes, frames, layout, rep)


@pytest.mark.parametrize("layout", [FM, LM])
@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_phase_form_biquad_arms(n, layout):
    rng = np.random.default_rng(10 * n + layout)
    arr, _ = G.sections_i32(n, rng)
    for lanes, frames in SHAPES:
        st = np.zeros((2 + 8 * n, lanes), np.uint32)
        st[:2] = rng.integers(0, 1 << 32, (2, lanes), dtype=np.uint64).astype(np.uint32)
        st[2:] = rng.integers(-(1 << 20), 1 << 20, (8 * n, lanes)).astype(np.int32).view(np.uint32)
        x = [rng.integers(-(1 << 28), 1 << 28, lanes * frames, dtype=np.int32) for _ in range(2)]
        run_pair("lockin_i32_biquad_process", arr, n, st, x, None, np.int32, lanes, frames, layout, False)


@pytest.mark.parametrize("layout", [FM, LM])
@pytest.mark.parametrize("order,cascade", [(1, 1), (2, 2), (1, 4), (2, 3)])
def test_external_lo_lowpass_arms(order, cascade, layout):
    rng = np.random.default_rng(100 * order + cascade + layout)
    ks = [[1 << 22, -(1 << 27)][:order] for _ in range(cascade)]
    cfg = H.lockin_cfg(ks)
    words = H.oracle().fn["lockin_state_words"](C.byref(cfg)) - 2
    for lanes, frames in SHAPES:
        st = rng.integers(0, 1 << 32, (words, lanes), dtype=np.uint64).astype(np.uint32)
        x = [rng.integers(-(1 << 28), 1 << 28, lanes * frames, dtype=np.int32) for _ in range(2)]
        lo = [rng.integers(-(1 << 31), (1 << 31) - 1, lanes * frames * 2, dtype=np.int64).astype(np.int32) for _ in range(2)]
        run_pair("lockin_i32_lo_process", cfg, None, st, x, lo, np.int32, lanes, frames, layout, True)


@pytest.mark.parametrize("layout", [FM, LM])
@pytest.mark.parametrize("n", [1, 2, 4])
def test_external_lo_biquad_arms_i32_and_f32(n, layout):
    rng = np.random.default_rng(1000 + 10 * n + layout)
    arr, _ = G.sections_i32(n, rng)
    arrf, _ = G.sections_f32(n, rng)
    for lanes, frames in SHAPES:
        st = rng.integers(-(1 << 20), 1 << 20, (8 * n, lanes)).astype(np.int32).view(np.uint32)
        x = [rng.integers(-(1 << 28), 1 << 28, lanes * frames, dtype=np.int32) for _ in range(2)]
        lo = [rng.integers(-(1 << 31), (1 << 31) - 1, lanes * frames * 2, dtype=np.int64).astype(np.int32) for _ in range(2)]
        run_pair("lockin_i32_biquad_lo_process", arr, n, st, x, lo, np.int32, lanes, frames, layout, True)
        stf = rng.standard_normal((8 * n, lanes)).astype(np.float32).view(np.uint32)
        xf = [rng.standard_normal(lanes * frames).astype(np.float32) for _ in range(2)]
        lof = [rng.standard_normal(lanes * frames * 2).astype(np.float32) for _ in range(2)]
        run_pair("lockin_f32_biquad_lo_process", arrf, n, stf, xf, lof, np.float32, lanes, frames, layout, True)


def test_ddc_lockin_example_recovers_dc_iq_on_hip():
    """examples/ddc_lockin.rs:100-111 on the GPU: 256 lanes carry the tone at different phases."""
    e = H.engine()
    lanes, n = 256, 16384
    arr, _ = G.sections_f32(1, None, f0=0.002)
    xs, los, exps = [], [], []
    for l in range(lanes):
        x, lo, ex = G.ddc_fixture(n, phi=0.37 + 0.02 * l) if l < 4 else (xs[l % 4], los[l % 4], exps[l % 4])
        xs.append(x), los.append(lo), exps.append(ex)
    x = np.stack(xs)        # LANE_MAJOR [lanes][frames]
    lo = np.stack(los)      # [lanes][frames][2]
    st = torch.zeros((8, lanes), dtype=torch.int32, device=DEV)
    y = torch.empty(lanes * n * 2, dtype=torch.float32, device=DEV)
    assert G.call_lo(e, "lockin_f32_biquad_lo_process", arr, 1, st, dev(x), dev(lo), y, lanes, n, LM, True) == 0, e.err()
    torch.cuda.synchronize()
    got = y.cpu().numpy().reshape(lanes, n, 2)[:, 12288:].astype(np.float64)
    for l in range(lanes):
        assert abs(got[l, :, 0].mean() - exps[l][0]) < 3e-3 and abs(got[l, :, 1].mean() - exps[l][1]) < 3e-3
        assert np.sqrt(((got[l] - np.array(exps[l])) ** 2).sum(axis=1).mean()) < 6e-3


def test_biquad_arms_run_on_the_multi_wave_kernel():
    """Which kernel: FrameMajor always, LaneMajor for whole 16-frame batches on aligned rows -> `lockin_waves_kernel` with the biquad
    chain as its arm functor (lockin_waves_biquad.hip); every other LaneMajor shape -> the one-thread-per-lane stream kernels."""
    import os

    if os.environ.get("IDSP_LOCKIN_NO_WAVES"):
        pytest.skip("forced onto the stream kernels")
    e = H.engine()
    rng = np.random.default_rng(77)
    arr, _ = G.sections_i32(2, rng)
    arrf, _ = G.sections_f32(2, rng)
    for lanes, frames, layout, waves in ((200, 37, FM, True), (128, 64, LM, True), (128, 40, LM, False)):
        x = dev(rng.integers(-(1 << 28), 1 << 28, lanes * frames, dtype=np.int32))
        lo = dev(rng.integers(-(1 << 31), (1 << 31) - 1, lanes * frames * 2, dtype=np.int64).astype(np.int32))
        y = torch.empty(lanes * frames * 2, dtype=torch.int32, device=DEV)
        st = torch.zeros((2 + 16, lanes), dtype=torch.int32, device=DEV)
        assert e.stream("lockin_i32_biquad_process", arr, 2, st, x, y, lanes, frames, layout) == 0
        assert e.last_kernel().startswith("lockin_waves_kernel[4 waves per 64 lanes]<[Biquad; 2]") == waves, e.last_kernel()
        assert G.call_lo(e, "lockin_i32_biquad_lo_process", arr, 2, st[2:], x, lo, y, lanes, frames, layout, True) == 0
        assert e.last_kernel().startswith("lockin_waves_kernel[4 waves per 64 lanes]<[Biquad; 2], LO") == waves, e.last_kernel()
        xf, lof, yf = x.view(torch.float32), lo.view(torch.float32), y.view(torch.float32)
        assert G.call_lo(e, "lockin_f32_biquad_lo_process", arrf, 2, st[2:], xf, lof, yf, lanes, frames, layout, True) == 0
        assert e.last_kernel().startswith("lockin_waves_kernel[4 waves per 64 lanes]<[Biquad<f32>; 2], LO") == waves, e.last_kernel()
    torch.cuda.synchronize()


def test_stream_kernel_forms_stay_covered():
    """The one-thread-per-lane processors of lockin_generic.hip are the fall-back for LaneMajor shapes only now: replay this file with
    the multi-wave kernel switched off so that their FrameMajor forms keep meeting the oracle too."""
    import os
    import subprocess
    import sys

    if os.environ.get("IDSP_LOCKIN_NO_WAVES"):
        pytest.skip("inner run")
    env = dict(os.environ, IDSP_DIAG="1", IDSP_LOCKIN_NO_WAVES="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_lockin_generic.py", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
