"""The stage-wave lock-in kernel (idsp_amd/csrc/lockin_waves.h: lockin_stages_kernel — one wave per cascade stage of each arm, four
read-out waves per 64 lanes, cossin through the full-circle table) against the oracle: every read-out (`Complex<i32>`, `arg`,
`norm_sqr`), `[Lowpass<1>; 2]` and `[Lowpass<2>; 2]`, arbitrary state, 1 ... 12 batches of 16 frames (the pipeline is four batches
deep: fewer batches than stages must drain correctly), chunked continuation, one and two lane groups per workgroup.  Default
dispatch takes the kernel up to 16384 lanes (`arg`: at every lane count); the forced runs put every eligible shape on it with one or two
groups per workgroup (the switches are read once per process, hence the subprocesses), and `idsp_last_kernel()` proves which kernel
ran.  Shapes the kernel does not take (ragged lanes, frames off the 16-frame grid, LaneMajor, other cascades) are the fuzz suite's."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import _harness as H

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENTRIES = (("lockin_i32_process", 2, np.int32, torch.int32), ("lockin_i32_arg", 1, np.int32, torch.int32),
           ("lockin_i32_norm_sqr", 1, np.int64, torch.int64))


def _dev(a):
    return torch.from_numpy(a).cuda()


def _one(o, e, rng, name, width, ndt, tdt, lanes, frames, order, split, expect):
    ks = [[int(rng.integers(1, 1 << 28))] if order == 1 else [int(rng.integers(1, 1 << 24)), -int(rng.integers(1, 1 << 29))] for _ in range(2)]
    cfg = H.lockin_cfg(ks)
    words = 2 + 4 * order * 2
    st = rng.integers(0, 1 << 32, size=(words, lanes), dtype=np.uint64).astype(np.uint32)
    x = rng.integers(-(1 << 31), (1 << 31) - 1, size=lanes * frames, dtype=np.int64).astype(np.int32)
    so = st.copy()
    yo = np.empty(lanes * frames * width, ndt)
    assert o.cfgcall(name, cfg, so, x, yo, lanes, frames, H.FM) == 0
    sg, xv = _dev(st.view(np.int32).copy()), _dev(x)
    yv = torch.zeros(yo.size, dtype=tdt, device="cuda")
    parts = [(0, frames)] if split is None else [(0, split), (split, frames)]
    for f0, f1 in parts:
        assert e.cfgcall(name, cfg, sg, xv[f0 * lanes:f1 * lanes], yv[f0 * lanes * width:f1 * lanes * width], lanes, f1 - f0, H.FM) == 0, e.err()
        if expect is not None:
            assert expect in e.last_kernel(), (e.last_kernel(), name, lanes, f1 - f0)
    torch.cuda.synchronize()
    ctx = (name, lanes, frames, order, split)
    assert np.array_equal(yv.cpu().numpy(), yo), ctx
    assert np.array_equal(sg.cpu().numpy().view(np.uint32), so), ctx


def _suite(expect_for):
    o, e = H.oracle(), H.engine()
    rng = np.random.default_rng(77)
    for lanes in (64, 128, 192, 384, 4096 + 128):
        for frames, split in ((16, None), (32, None), (48, 16), (64, 32), (80, None), (192, 80)):
            for order in (1, 2):
                for name, width, ndt, tdt in ENTRIES:
                    _one(o, e, rng, name, width, ndt, tdt, lanes, frames, order, split, expect_for(lanes))


def test_stage_kernel_default_dispatch(gpu):
    """Default dispatch: all of these lane counts are below 16384, so every call must land on the stage-wave kernel."""
    if os.environ.get("IDSP_LOCKIN_STAGE_GROUPS") or os.environ.get("IDSP_LOCKIN_NO_STAGES"):
        pytest.skip("inside a forced run")
    _suite(lambda lanes: "lockin_stages_kernel[8 waves per 64 lanes]")


def test_stage_kernel_forced_inner(gpu):
    """Body of the forced runs below (skipped unless a switch is set)."""
    g = os.environ.get("IDSP_LOCKIN_STAGE_GROUPS")
    if os.environ.get("IDSP_LOCKIN_NO_STAGES"):
        _suite(lambda lanes: "lockin_waves_kernel")
    elif g == "2":
        _suite(lambda lanes: "lockin_stages_kernel[16 waves per 128 lanes]" if lanes % 128 == 0 else "lockin_stages_kernel[8 waves per 64 lanes]")
    elif g == "1":
        _suite(lambda lanes: "lockin_stages_kernel[8 waves per 64 lanes]")
    else:
        pytest.skip("no switch set")


@pytest.mark.parametrize("form", [dict(IDSP_LOCKIN_STAGE_GROUPS="2"), dict(IDSP_LOCKIN_STAGE_GROUPS="1"), dict(IDSP_LOCKIN_NO_STAGES="1")],
                         ids=lambda f: ",".join(f"{k[12:]}={v}" for k, v in f.items()))
def test_stage_kernel_forced(gpu, form):
    if os.environ.get("IDSP_LOCKIN_STAGE_GROUPS") or os.environ.get("IDSP_LOCKIN_NO_STAGES"):
        pytest.skip("already inside a forced run")
    env = dict(os.environ, IDSP_DIAG="1", **form)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_lockin_stages.py", "tests/test_gpu_lockin_fuzz.py", "-m", "gpu", "-x", "-q",
                        "-k", "forced_inner or random_shapes"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_two_groups_at_the_arg_lane_counts(gpu):
    """`arg` read-out at 32768 lanes: default dispatch puts two lane groups in a workgroup; 64 frames against the oracle."""
    if os.environ.get("IDSP_LOCKIN_STAGE_GROUPS") or os.environ.get("IDSP_LOCKIN_NO_STAGES"):
        pytest.skip("inside a forced run")
    o, e = H.oracle(), H.engine()
    rng = np.random.default_rng(78)
    _one(o, e, rng, "lockin_i32_arg", 1, np.int32, torch.int32, 32768, 64, 2, None, "lockin_stages_kernel[16 waves per 128 lanes]")
    _one(o, e, rng, "lockin_i32_process", 2, np.int32, torch.int32, 32768, 64, 2, None, "lockin_waves_kernel")


def test_dds_on_the_small_table_when_forced(gpu):
    """IDSP_DDS_NO_CIRCLE=1: the DDS parity shapes (which include FrameMajor calls of 256 frames and more) on the 512-byte cossin table."""
    if os.environ.get("IDSP_DDS_NO_CIRCLE"):
        pytest.skip("already inside a forced run")
    env = dict(os.environ, IDSP_DIAG="1", IDSP_DDS_NO_CIRCLE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-m", "gpu", "-x", "-q", "-k", "dds"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
