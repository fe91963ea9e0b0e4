"""FM discriminator graph (examples/fm_disc.rs:25-50): CPU oracle vs the Python restatement, and the HIP
path vs the oracle (bit-exact incl. state, continuation, both layouts).  The reference's own test of the
graph (corr / gain / rms bounds) is in tests/_kat_cases.py::case_fm_disc_tracks_known_modulation."""
import os
import subprocess
import sys

import numpy as np
import pytest

from idsp_amd import _abi
from oracle import spec
from tests import _harness as H
from tests._backends import GpuBackend, OracleBackend

FM, LM = H.FM, H.LM
I32_MIN, I32_MAX = -(1 << 31), (1 << 31) - 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(rng):
    cfg = _abi.FmDisc()
    cfg.carrier = int(rng.integers(I32_MIN, I32_MAX))
    cfg.deemph.ba[:] = [int(v) for v in rng.integers(-(1 << 29), 1 << 29, size=5)]
    cfg.deemph.frac = int(rng.integers(20, 31))
    return cfg


def _x(rng, n):
    x = rng.integers(I32_MIN, I32_MAX, size=2 * n, dtype=np.int64, endpoint=True).astype(np.int32)
    x[rng.integers(0, 2 * n, size=max(1, n // 4))] = rng.choice(np.array([I32_MIN, I32_MAX, 0, 1, -1], np.int32))
    return x


@pytest.mark.parametrize("layout", [FM, LM])
def test_fm_disc_oracle_equals_spec(oracle_lib, layout):
    ob = OracleBackend()
    rng = np.random.default_rng(60 + layout)
    lanes, frames = 4, 60
    cfg = _cfg(rng)
    st = np.zeros((7, lanes), np.uint32)
    prevs = [[None] for _ in range(lanes)]
    dfs = [spec.DirectForm1() for _ in range(lanes)]
    for part in range(2):
        x = _x(rng, lanes * frames)
        rc, y = ob.cfgcall("fm_disc_i32", cfg, st, x, (lanes * frames,), np.int32, lanes, frames, layout)
        assert rc == 0
        xm = x.reshape(frames, lanes, 2).transpose(1, 0, 2) if layout == FM else x.reshape(lanes, frames, 2)
        ym = y.reshape(frames, lanes).T if layout == FM else y.reshape(lanes, frames)
        for l in range(lanes):
            for f in range(frames):
                want = spec.fm_disc(cfg.carrier, list(cfg.deemph.ba), cfg.deemph.frac, prevs[l], dfs[l],
                                    (int(xm[l, f, 0]), int(xm[l, f, 1])))
                assert want == int(ym[l, f]), (part, l, f)
            assert int(st[0, l]) == 1 and spec.i32(int(st[1, l])) == prevs[l][0][0] and spec.i32(int(st[2, l])) == prevs[l][0][1]


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [FM, LM])
def test_fm_disc_gpu_parity(gpu, layout):
    ob, gb = OracleBackend(), GpuBackend()
    rng = np.random.default_rng(70 + layout)
    for lanes, frames in [(1, 1), (63, 23), (65, 47), (257, 64), (3, 1000), (1024, 33), (300, 200)]:
        cfg = _cfg(rng)
        init = np.zeros((7, lanes), np.uint32)
        init[0, ::2] = 1  # half the lanes start with a previous sample
        init[1:] = rng.integers(0, 1 << 32, size=(6, lanes), dtype=np.uint64).astype(np.uint32)
        so, sg = init.copy(), init.copy()
        for part in range(2):
            x = _x(rng, lanes * frames)
            rco, yo = ob.cfgcall("fm_disc_i32", cfg, so, x, (lanes * frames,), np.int32, lanes, frames, layout)
            rcg, yg = gb.cfgcall("fm_disc_i32", cfg, sg, x, (lanes * frames,), np.int32, lanes, frames, layout)
            assert rco == 0 and rcg == 0, H.engine().err()
            assert np.array_equal(yo, yg) and np.array_equal(so, sg), (lanes, frames, part)
    bad = _cfg(rng)
    bad.deemph.frac = 32
    assert gb.cfgcall("fm_disc_i32", bad, np.zeros((7, 1), np.uint32), np.zeros(2, np.int32), (1,), np.int32, 1, 1, LM)[0] == -1


FORCED_WAVES = os.environ.get("IDSP_FM_DISC_WAVES") if os.environ.get("IDSP_DIAG") == "1" else None


@pytest.mark.gpu
def test_fm_disc_role_waves_frame_major(gpu):
    """`fm_disc_waves_kernel` (idsp_amd/csrc/dds.hip): the discriminator of a tile's frames on four front waves, the deemphasis
    biquad on a fifth.  Frame counts around the 32-frame tiles and the 8-frame wave shares, lanes around the 64-lane workgroups,
    continuation across calls (the state's `prev` feeds frame 0), lanes that start without a previous sample."""
    ob, gb = OracleBackend(), GpuBackend()
    rng = np.random.default_rng(72)
    for lanes, frames in [(64, 8), (1, 9), (130, 31), (64, 32), (65, 33), (200, 63), (128, 64), (70, 95), (64, 129), (4096, 70), (40960, 35)]:
        cfg = _cfg(rng)
        init = np.zeros((7, lanes), np.uint32)
        init[0, ::3] = 1
        init[1:] = rng.integers(0, 1 << 32, size=(6, lanes), dtype=np.uint64).astype(np.uint32)
        so, sg = init.copy(), init.copy()
        for part in range(2):
            x = _x(rng, lanes * frames)
            rco, yo = ob.cfgcall("fm_disc_i32", cfg, so, x, (lanes * frames,), np.int32, lanes, frames, FM)
            rcg, yg = gb.cfgcall("fm_disc_i32", cfg, sg, x, (lanes * frames,), np.int32, lanes, frames, FM)
            assert rco == 0 and rcg == 0, H.engine().err()
            k = gpu.fn["last_kernel"]().decode()
            assert k.startswith("stream_frame_major" if FORCED_WAVES == "0" else f"fm_disc_waves_kernel<{FORCED_WAVES or 4}>"), k
            assert np.array_equal(yo, yg) and np.array_equal(so, sg), (lanes, frames, part)


@pytest.mark.gpu
def test_fm_disc_role_waves_lane_major(gpu):
    """`fm_disc_waves_lm_kernel` (round 4): LaneMajor rows of whole 32-frame tiles — input lines by LDS-DMA, output lines stored
    eight threads per lane.  One to many tiles, lanes around the 64-lane workgroups (ragged last workgroup), continuation across
    calls, lanes that start without a previous sample; rows of 32 + 16 k frames: the whole tiles here, the last 16 frames of a row that is not whole tiles on the tile kernel
    behind it; other frame counts stay on the tile kernel."""
    ob, gb = OracleBackend(), GpuBackend()
    rng = np.random.default_rng(73)
    for lanes, frames in [(64, 32), (1, 64), (65, 96), (130, 32), (200, 160), (63, 1024), (4096, 64), (20000, 96), (64, 40), (70, 33), (65, 112), (64, 48), (130, 16),
                          (200, 2064)]:
        cfg = _cfg(rng)
        init = np.zeros((7, lanes), np.uint32)
        init[0, ::3] = 1
        init[1:] = rng.integers(0, 1 << 32, size=(6, lanes), dtype=np.uint64).astype(np.uint32)
        so, sg = init.copy(), init.copy()
        for part in range(3):
            x = _x(rng, lanes * frames)
            rco, yo = ob.cfgcall("fm_disc_i32", cfg, so, x, (lanes * frames,), np.int32, lanes, frames, LM)
            rcg, yg = gb.cfgcall("fm_disc_i32", cfg, sg, x, (lanes * frames,), np.int32, lanes, frames, LM)
            assert rco == 0 and rcg == 0, H.engine().err()
            k = gpu.fn["last_kernel"]().decode()
            aligned = not os.environ.get("IDSP_TEST_MISALIGN")  # tests/test_gpu_misaligned.py replays this file on buffers 4 / 8 bytes off
            assert k.startswith("fm_disc_waves_lm_kernel") == (frames >= 32 and frames % 16 == 0 and aligned), (k, frames)
            assert k.endswith("(last frames % 32)") == (frames >= 32 and frames % 32 == 16 and aligned), (k, frames)
            assert np.array_equal(yo, yg) and np.array_equal(so, sg), (lanes, frames, part)


@pytest.mark.gpu
@pytest.mark.parametrize("waves", ["3", "0"])
def test_fm_disc_other_forms(gpu, waves):
    """three front waves, and the one-thread-per-lane stream kernel the role waves replaced (diagnostic switch, own process)"""
    if FORCED_WAVES is not None:
        pytest.skip("already inside a forced run")
    env = dict(os.environ, IDSP_DIAG="1", IDSP_FM_DISC_WAVES=waves)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_fm_disc.py", "-m", "gpu", "-x", "-q"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
