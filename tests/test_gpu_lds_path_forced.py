"""The LDS-DMA FrameMajor kernel is normally chosen only for cheap processors on launches of 256+ waves.
Force it (IDSP_DIAG=1 + IDSP_LDS_COST / IDSP_LDS_MIN_WAVES, read once per process -> subprocess) for the heavy and the
two-word-output processors too and check them against the oracle: lock-in (Complex out, LUT in LDS),
8-section cascade, dither, Normal, a 4-section chain; and check every clamp variant, f32 DF1 and f32 DF2T
(the processors profiles/NOTES.md section 5 quotes throughput for on this kernel) out of place and in place.
Without IDSP_DIAG=1 the switches must be ignored (second test)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %r)
from idsp_amd import _abi
from tests import _harness as H
from tests import _nw_cases as W
from tests._backends import GpuBackend, OracleBackend
ob, gb = OracleBackend(), GpuBackend()
rng = np.random.default_rng(5)
FM = H.FM
lanes, frames = 512, 203   # whole 256-lane blocks, ragged last tile
def both(op, cfg, n, words, x, want="stream_frame_major_lds"):
    init = rng.integers(0, 1 << 32, size=(words, lanes), dtype=np.uint64).astype(np.uint32)
    if x.dtype == np.float32:
        init = rng.standard_normal(size=(words, lanes)).astype(np.float32).view(np.uint32)
    for inplace in (False, True):
        so, sg = init.copy(), init.copy()
        rco, yo = ob.stream(op, cfg, n, so, x.copy(), lanes, frames, FM, inplace=inplace)
        rcg, yg = gb.stream(op, cfg, n, sg, x.copy(), lanes, frames, FM, inplace=inplace)
        assert rco == 0 and rcg == 0, (op, H.engine().err())
        assert H.engine().fn["last_kernel"]().decode().startswith(want), (op, H.engine().fn["last_kernel"]())
        assert np.array_equal(yo.view(np.uint32), yg.view(np.uint32)) and np.array_equal(so, sg), (op, inplace)
xi = rng.integers(-(1 << 31), (1 << 31) - 1, size=lanes * frames, dtype=np.int64).astype(np.int32)
rows = [(rng.integers(-(1 << 29), 1 << 29, size=5).tolist(), 29) for _ in range(8)]
both("cascade_i32_df1", H.biquad_i32(rows), 8, 18, xi)
# every clamp variant and f32 DF1 on the LDS-DMA kernel (ring of 7 and of 8 tiles), ragged last tile, in place too
crow = lambda k: [(r[0], r[1], int(rng.integers(-1000, 1000)), -(1 << 29) - k, (1 << 30) + k) for r in rows[:k]]
both("biquad_i32_df1", H.biquad_i32(rows[:1]), 1, 4, xi)
both("biquad_i32_df1_clamp", H.biquad_clamp_i32(crow(1)), 1, 4, xi)
both("biquad_i32_df1_clamp", H.biquad_clamp_i32(crow(3)), 3, 12, xi)
both("biquad_i32_dither_clamp", H.biquad_clamp_i32(crow(1)), 1, 5, xi)
both("biquad_i32_wide_clamp", H.biquad_clamp_i32(crow(1)), 1, 6, xi)
both("biquad_i32_wide_clamp", H.biquad_clamp_i32(crow(2)), 2, 12, xi)
both("biquad_i32_dither", H.biquad_i32(rows[:1]), 1, 5, xi)
both("biquad_i32_wide", H.biquad_i32(rows[:4]), 4, 24, xi)
both("normal_i32_df1", H.biquad_i32(rows[:2]), 2, 8, xi)
xf = rng.standard_normal(lanes * frames).astype(np.float32)
both("biquad_f32_df2t", H.biquad_f32([(rng.standard_normal(5) * 0.3).tolist() for _ in range(3)]), 3, 6, xf)
frow = lambda k: [(rng.standard_normal(5) * 0.3).tolist() for _ in range(k)]
fcrow = lambda k: [(r, float(rng.standard_normal() * 0.1), -0.7, 0.9) for r in frow(k)]
both("biquad_f32_df1", H.biquad_f32(frow(1)), 1, 4, xf)
both("biquad_f32_df1", H.biquad_f32(frow(2)), 2, 8, xf)
both("biquad_f32_df1_clamp", H.biquad_clamp_f32(fcrow(1)), 1, 4, xf)
both("biquad_f32_df2t", H.biquad_f32(frow(1)), 1, 2, xf)
both("biquad_f32_df2t_clamp", H.biquad_clamp_f32(fcrow(1)), 1, 2, xf)
both("biquad_f32_df2t_clamp", H.biquad_clamp_f32(fcrow(2)), 2, 4, xf)
lc = H.lockin_cfg([[1 << 20, -(1 << 27)]] * 2)
st = rng.integers(0, 1 << 32, size=(18, 65536), dtype=np.uint64).astype(np.uint32)  # > split threshold: unsplit LockinProc
L, F = 65536, 40
x = rng.integers(-(1 << 28), 1 << 28, size=L * F, dtype=np.int32)
so, sg = st.copy(), st.copy()
rco, yo = ob.cfgcall("lockin_i32_process", lc, so, x, (L * F * 2,), np.int32, L, F, FM)
rcg, yg = gb.cfgcall("lockin_i32_process", lc, sg, x, (L * F * 2,), np.int32, L, F, FM)
assert rco == 0 and rcg == 0 and np.array_equal(yo, yg) and np.array_equal(so, sg), "lockin"
# the one- / two-thread-per-lane stream kernels behind the multi-wave lock-in (IDSP_LOCKIN_NO_WAVES=1), all three outputs
for L, F in ((512, 203), (100, 64)):
    x = rng.integers(-(1 << 28), 1 << 28, size=L * F, dtype=np.int32)
    st = rng.integers(0, 1 << 32, size=(18, L), dtype=np.uint64).astype(np.uint32)
    for layout in (H.FM, H.LM):
        for name, width, dt in (("lockin_i32_process", 2, np.int32), ("lockin_i32_arg", 1, np.int32), ("lockin_i32_norm_sqr", 1, np.int64)):
            so, sg = st.copy(), st.copy()
            rco, yo = ob.cfgcall(name, lc, so, x, (L * F * width,), dt, L, F, layout)
            rcg, yg = gb.cfgcall(name, lc, sg, x, (L * F * width,), dt, L, F, layout)
            assert rco == 0 and rcg == 0 and np.array_equal(yo, yg) and np.array_equal(so, sg), (name, L, F, layout)
print("forced LDS path ok")
"""


def test_heavy_processors_on_the_lds_dma_kernel(gpu):
    env = dict(os.environ, IDSP_DIAG="1", IDSP_LDS_COST="100000", IDSP_LDS_MIN_WAVES="0", IDSP_LOCKIN_NO_WAVES="1", IDSP_NO_FM_STAGED="1")
    r = subprocess.run([sys.executable, "-c", SNIPPET % ROOT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "forced LDS path ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_switches_are_ignored_without_idsp_diag(gpu):
    """A stray IDSP_NO_LDS_PATH / IDSP_LDS_COST in a production environment must not change dispatch."""
    code = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, %r)
from tests import _harness as H
from tests._backends import GpuBackend
gb = GpuBackend()
lanes, frames = 65536, 16  # from 49152 lanes up FrameMajor takes the dense-sweep LDS-DMA kernel (below: the staged single-wave kernel)
x = np.zeros(lanes * frames, np.int32); st = np.zeros((4, lanes), np.uint32)
rc, _ = gb.stream("biquad_i32_df1", H.biquad_i32([([1 << 28, 0, 0, 0, 0], 30)]), 1, st, x, lanes, frames, H.FM)
assert rc == 0
print(H.engine().fn["last_kernel"]().decode())
""" % ROOT
    env = {k: v for k, v in os.environ.items() if k != "IDSP_DIAG"}
    env.update(IDSP_NO_LDS_PATH="1", IDSP_LDS_COST="0")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("stream_frame_major_sweep[1 block/workgroup]<"), r.stdout + r.stderr
    env["IDSP_DIAG"] = "1"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("stream_frame_major<"), r.stdout + r.stderr
