"""The dense-sweep FrameMajor kernel (`stream_frame_major_sweep`, idsp_amd/csrc/fm_sweep.h, round 5) against the oracle, bit for bit
(outputs, written-back state, untouched neighbours), through the C ABI's `_pitch` entries.

What the kernel replaces is the loop nest `Lanes::process_view` over `SplitProcess::block` (dsp-process/src/compose.rs:468-494,
process.rs:122-141: any N lanes).  Covered here: every geometry class `sweep_geometry` produces — full 256-lane blocks and narrower
ones, 1 / 2 / 4 / 8 / 16 sub-blocks per workgroup, sub-blocks that lie beyond the last lane (clones), partial last blocks, several
sweeps inside one launch — at frame counts that are not whole tiles, in place and out of place, dense rows and lane blocks of wider
tensors; at full-size lane counts on the default dispatch, and at small lane counts with the grid capped by a diagnostic switch
(`IDSP_SWEEP_MAX_GRID`) so that the same classes are reached in seconds.  The kernel taken is asserted via `idsp_last_kernel()`."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import test_gpu_frame_major_staged as FMS
from tests.test_gpu_pitch import cases

pytestmark = pytest.mark.gpu
FORCED = os.environ.get("IDSP_DIAG") == "1" and os.environ.get("IDSP_SWEEP_MAX_GRID") is not None


def kernel_of(eng):
    return eng.fn["last_kernel"]().decode()


def sweep_cases(rng):
    """single-pass, 4-byte, LDS-eligible entries of the biquad family (f64 and long chains take other kernels)"""
    return [c for c in cases(rng) if c[4] != np.float64 and c[2] <= 2]


def test_default_dispatch_full_size_lane_counts(gpu):
    """65536 (one block per workgroup) ... 2^20 lanes (sixteen), ragged counts in between (narrow blocks, clones, a partial last
    block), a lane block of a wider tensor; short frame counts that are no whole tiles."""
    if FORCED:
        pytest.skip("forced small-shape run")
    rng = np.random.default_rng(501)
    cs = sweep_cases(rng)
    # (lanes, frames, pitch, lane offset, blocks per workgroup expected for the cheap single sections)
    shapes = [(65536, 19, 65536, 0, 1), (49152, 21, 49152, 0, 1), (100000, 33, 100000 + 16, 0, 2), (131072, 17, 131072, 0, 2), (200000, 18, 200000, 0, 4),
              (262144, 19, 262144 + 64, 32, 4), (300016, 20, 300016, 0, 8), (1048576, 18, 1048576, 0, 16), (1000000, 17, 1000000, 0, 16),
              (90000, 41, 131072, 16, 2)]
    big = {"biquad_i32_df1", "biquad_f32_df2t_clamp", "biquad_i32_df1_clamp"}  # the big shapes: a cheap i32, a cheap f32 and the 2-section chain
    for i, (lanes, frames, pitch, off, blocks) in enumerate(shapes):
        for j, (op, cfg, n, words, dt) in enumerate(cs):
            if lanes > 70000 and (op not in big or (lanes > 300000 and (i + j) % 2)):
                continue
            FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, bool((i + j) & 1), off=off)
            k = kernel_of(gpu)
            assert k.startswith("stream_frame_major_sweep["), (op, lanes, k)
            if n == 1 and op in ("biquad_i32_df1", "biquad_f32_df2t_clamp", "biquad_f32_df1"):
                assert k.startswith(f"stream_frame_major_sweep[{blocks} block"), (op, lanes, k)


def test_in_place_with_clone_sub_blocks_at_16_blocks_per_workgroup(gpu):
    """983296 lanes = 15 x 65536 + 256: full 256-lane blocks, sixteen per workgroup, and the last workgroups' sixteenth sub-block is a clone
    of their first (it re-requests those x rows).  With y == x the two-barrier schedule of the unclamped i32 DF1 / f32 DF2T sections would store
    a tile while the clone's request of the same rows may be in flight (round-5 advice): such launches take the one-barrier schedule.  Every
    output and the state against the oracle, in place and out of place, several tile periods of frames."""
    if FORCED:
        pytest.skip("forced small-shape run")
    rng = np.random.default_rng(504)
    for op, cfg, n, words, dt in sweep_cases(rng):
        if op not in ("biquad_i32_df1", "biquad_f32_df2t") or n != 1:
            continue
        for inplace in (True, False):
            FMS.run_case(gpu, op, cfg, n, words, dt, rng, 983296, 70, 983296, inplace)
            assert kernel_of(gpu).startswith("stream_frame_major_sweep[16 block"), kernel_of(gpu)


def test_lane_counts_a_little_above_whole_rounds_split(gpu):
    """65552 lanes = 65536 on the sweep kernel + 16 beside them on a second stream (lane_stream.h: one sweep of half-empty blocks is
    slower); rows stay on the 64-byte grid."""
    if FORCED:
        pytest.skip("forced small-shape run")
    rng = np.random.default_rng(502)
    op, cfg, n, words, dt = sweep_cases(rng)[0]
    for lanes in (65552, 131072 + 4096):  # (3 x 65536 + 3392 = 200000 lanes above: no split, the three rounds would be narrow blocks)
        FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, 37, lanes, False)
        assert kernel_of(gpu).startswith("stream_frame_major_sweep + stream_frame_major_staged (remainder, second stream)"), kernel_of(gpu)


def test_small_shapes_inner(gpu):
    """(inside the forced run below) grid capped at a few workgroups: every LPT, clones, partial blocks and several sweeps per launch"""
    if not FORCED:
        pytest.skip("runs inside test_every_geometry_class_on_a_capped_grid")
    rng = np.random.default_rng(503)
    cs = sweep_cases(rng)
    seen = set()
    lanes_list = [16, 64, 252, 256, 260, 512, 700, 1024, 1500, 2048, 3000, 4096, 5000, 8192, 9000, 16384, 20000, 40000]
    for k, lanes in enumerate(lanes_list):
        for j, (op, cfg, n, words, dt) in enumerate(cs):
            if (j + k) % 3:
                continue
            frames = int(rng.choice([16, 17, 23, 24, 57, 64, 130]))
            pad = int(rng.choice([0, 16, 64]))
            off = int(rng.choice([0, 16]))
            FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, (lanes + 15) // 16 * 16 + pad + off, bool((j + k) & 1), off=off)
            assert kernel_of(gpu).startswith("stream_frame_major_sweep["), (op, lanes, kernel_of(gpu))
            seen.add(kernel_of(gpu).split("<")[0])
    want = int(os.environ.get("IDSP_SWEEP_WANT_FORMS", "3"))
    assert len(seen) >= want, seen


@pytest.mark.parametrize("grid,forms", [("2", 5), ("5", 4), ("256", 1)])
def test_every_geometry_class_on_a_capped_grid(gpu, grid, forms):
    """IDSP_DIAG=1 IDSP_SWEEP_MIN_LANES=16 IDSP_SWEEP_MAX_GRID=g: small tensors on the sweep kernel with at most g workgroups."""
    if FORCED:
        pytest.skip("already inside a forced run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IDSP_DIAG="1", IDSP_SWEEP_MIN_LANES="16", IDSP_SWEEP_MAX_GRID=grid, IDSP_SWEEP_WANT_FORMS=str(forms))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "small_shapes_inner"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_rows_off_the_grid_inner(gpu):
    """(inside the forced run below) full 256-lane blocks on rows off the 64-byte grid: the split-request instantiation (fm_sweep.h XC) at every
    blocks-per-workgroup count, whole and partial last blocks, dense odd pitches"""
    if not (FORCED and os.environ.get("IDSP_SWEEP_OFFGRID_MIN_LANES")):
        pytest.skip("runs inside test_full_blocks_on_rows_off_the_grid")
    rng = np.random.default_rng(504)
    cs = sweep_cases(rng)
    seen = set()
    for k, lpt in enumerate((1, 2, 4, 8, 16)):
        for j, (op, cfg, n, words, dt) in enumerate(cs):
            if (j + k) % 2:
                continue
            lanes = 8 * 256 * lpt - (0 if (j + k) % 4 else 100)  # whole blocks / a partial last one
            frames = int(rng.choice([16, 23, 57, 64, 130]))
            pitch = lanes + int(rng.choice([1, 4, 8, 17]))
            FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, bool((j + k) & 1))
            assert kernel_of(gpu).startswith("stream_frame_major_sweep["), (op, lanes, kernel_of(gpu))
            seen.add(kernel_of(gpu).split("<")[0])
    assert sum("XCD-contiguous" in k for k in seen) >= 4, seen


def test_full_blocks_on_rows_off_the_grid(gpu):
    """IDSP_DIAG=1 IDSP_SWEEP_MIN_LANES=16 IDSP_SWEEP_OFFGRID_MIN_LANES=1 IDSP_SWEEP_MAX_GRID=8: small tensors with odd row pitches on the sweep kernel's
    XCD-contiguous order with 8 workgroups."""
    if FORCED:
        pytest.skip("already inside a forced run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IDSP_DIAG="1", IDSP_SWEEP_MIN_LANES="16", IDSP_SWEEP_MAX_GRID="8", IDSP_SWEEP_OFFGRID_MIN_LANES="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "rows_off_the_grid_inner"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("grid", ["3", "256"])
def test_other_processor_families_on_the_sweep_kernel(gpu, grid):
    """The parity suites of the whole biquad family, the per-lane coefficient banks, `Normal` and `Lowpass` (every LDS-eligible processor:
    chains, cascades with shared delay lines, clamp / dither / wide sections, f32) re-run with the sweep kernel taking every FrameMajor
    shape from 16 lanes and 16 frames up, on a grid of at most 3 workgroups (several blocks per workgroup, several sweeps per launch)
    and on the default one."""
    if FORCED:
        pytest.skip("already inside a forced run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IDSP_DIAG="1", IDSP_SWEEP_MIN_LANES="16", IDSP_SWEEP_MAX_GRID=grid)
    suites = ["tests/test_gpu_parity.py", "tests/test_gpu_bylane.py", "tests/test_gpu_normal_wdf.py", "tests/test_gpu_pitch.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", *suites, "-m", "gpu", "-x", "-q", "-k", "not last_kernel and not on_the_lds_kernel"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
