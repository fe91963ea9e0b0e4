"""FRAME_MAJOR lane counts that are not multiples of 256 on the LDS-DMA kernel (`stream_frame_major_lds`,
idsp_amd/csrc/lane_stream.h): the kernel's last 256-lane block may be ragged when the lane count is a multiple of 4
(whole 16-byte pieces) — threads whose piece lies beyond the last lane re-request and re-store an in-range thread's
piece; rows that do not start on 64-byte boundaries (what dense rows of such lane counts are) take the form of the kernel
that gives every XCD a contiguous eighth of the lane blocks.  The reference takes any N in `Lanes<C>` (dsp-process/src/compose.rs:468); round 2 dropped such shapes to the
register-window kernel.  Against the oracle bit for bit (outputs and state), out of place and in place, dense rows and a
lane block of a wider tensor whose neighbouring lanes must stay untouched; the kernel taken is asserted through
`idsp_last_kernel()`.  Lane counts that are not multiples of 4: the last lanes % 4 lanes on a second stream."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import test_gpu_frame_major_staged as FMS
from tests.test_gpu_pitch import cases

pytestmark = pytest.mark.gpu
SMALL = os.environ.get("IDSP_DIAG") == "1" and os.environ.get("IDSP_LDS_MIN_WAVES") == "1"


def kernel_of(eng):
    return eng.fn["last_kernel"]().decode()


def lds_cases(rng):
    """single-pass, 4-byte, LDS-eligible entries of the biquad family"""
    return [c for c in cases(rng) if c[4] != np.float64 and c[2] <= 2]


def test_default_dispatch_takes_the_lds_kernel_on_ragged_lane_counts(gpu):
    if SMALL:
        pytest.skip("forced small-shape run")
    rng = np.random.default_rng(301)
    cs = lds_cases(rng)
    # (lanes, frames, pitch, lane offset)
    shapes = [(65000, 19, 65000, 0), (49156, 70, 49156, 0), (65532, 9, 65536, 4), (100000, 33, 100000, 0), (65000, 130, 65540, 260), (65000, 20, 65024, 0),
              (65536, 41, 65544, 0)]
    for i, (lanes, frames, pitch, off) in enumerate(shapes):
        for j, (op, cfg, n, words, dt) in enumerate(cs):
            if (i + j) % 3 and lanes > 65000:
                continue  # the big shapes on a third of the entries
            inplace = bool((i + j) & 1)
            FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, inplace, off=off)
            k = kernel_of(gpu)
            # rows off the 64-byte grid (dense 65000 / 100000 lanes, a base 16 bytes into a row) take the XCD-contiguous block order
            aligned = (pitch * 4) % 64 == 0 and (off * 4) % 64 == 0
            # (round 5: rows ON the grid take the dense-sweep kernel from 16 frames up — tests/test_gpu_sweep.py; off it, these lane counts
            # (<= 98304: one round of workgroups) keep this kernel, larger ones go to the sweep kernel with the same XCD-contiguous order)
            # (... and so do the ones up to 53248 lanes, several frames per segment: dispatch_thresholds.h kSweepOffGridSmallMax)
            want = ("stream_frame_major_sweep[" if lanes <= 53248 and frames >= 16 else "stream_frame_major_lds[XCD-contiguous blocks]<") if not aligned else \
                "stream_frame_major_sweep[" if frames >= 16 else "stream_frame_major_lds<"
            assert k.startswith(want), (op, lanes, pitch, off, k)
            assert ("XCD-contiguous" in k.split("<")[0]) == (not aligned), (op, lanes, pitch, off, k)


def test_lane_counts_that_are_not_multiples_of_four_run_their_last_lanes_beside(gpu):
    """tests/test_gpu_rows_off_16_byte_grid.py holds the matrix; here: the body is this file's ragged LDS-DMA launch"""
    if SMALL:
        pytest.skip("forced small-shape run")
    rng = np.random.default_rng(302)
    op, cfg, n, words, dt = lds_cases(rng)[0]
    for lanes in (65537, 65001):
        FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, 21, lanes, False)
        k = kernel_of(gpu)
        assert k.startswith("stream_frame_major_lds[XCD-contiguous blocks]<") and k.endswith("(lanes % 4, second stream)"), k


def test_small_ragged_shapes_inner(gpu):
    """(inside the forced run below) every ragged tail length class at small sizes, one workgroup per block"""
    if not SMALL:
        pytest.skip("runs inside test_every_tail_length_on_a_forced_lds_launch")
    rng = np.random.default_rng(303)
    cs = lds_cases(rng)
    tails = [4, 8, 60, 64, 68, 128, 200, 252]
    seen = set()
    for k, tail in enumerate(tails):
        for blocks in (0, 1, 5):
            lanes = blocks * 256 + tail
            for j, (op, cfg, n, words, dt) in enumerate(cs):
                if (j + k) % 2:
                    continue
                frames = int(rng.choice([1, 7, 8, 9, 57, 64, 130]))
                pad = int(rng.choice([0, 4, 64]))
                off = int(rng.choice([0, 4]))
                FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, lanes + pad + off, bool((j + k + blocks) & 1), off=off)
                assert kernel_of(gpu).startswith("stream_frame_major_lds"), (op, lanes, kernel_of(gpu))
                seen.add(kernel_of(gpu).split("<")[0])
    assert seen == {"stream_frame_major_lds", "stream_frame_major_lds[XCD-contiguous blocks]"}, seen


@pytest.mark.parametrize("grid", ["0", "3"])
def test_every_tail_length_on_a_forced_lds_launch(gpu, grid):
    """IDSP_DIAG=1 IDSP_LDS_MIN_WAVES=1 IDSP_NO_FM_STAGED=1 puts small lane counts on the LDS-DMA kernel (one workgroup per
    block, or a persistent grid of 3) so that every tail length is reached in seconds."""
    if SMALL:
        pytest.skip("already inside a forced run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IDSP_DIAG="1", IDSP_LDS_MIN_WAVES="1", IDSP_NO_FM_STAGED="1", IDSP_LDS_GRID=grid)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "small_ragged_shapes_inner"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
