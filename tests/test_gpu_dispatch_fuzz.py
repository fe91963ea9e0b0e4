"""Randomised shapes through the biquad-family `_pitch` entries in both layouts: lane counts, frame counts, pitches, base
offsets and in-place calls drawn so that every stream kernel of the launcher is reached (staged 16-byte kernels with 64 / 32 /
16 lanes per wave, 4-byte tile kernel, register window, LDS-DMA where the shape allows) — the oracle decides, bit for bit,
and the row padding / neighbouring lanes must stay untouched.  Seeded: the cases are the same on every run."""
import numpy as np
import pytest

from tests import _harness as H
from tests.test_gpu_pitch import cases
from tests import test_gpu_frame_major_staged as FMS
from tests import test_gpu_lane_major_staged as LMS

pytestmark = pytest.mark.gpu


def draw(rng):
    kind = rng.integers(0, 4)
    lanes = int([rng.integers(1, 70), rng.integers(1, 40) * 4, rng.integers(60, 1200), rng.integers(1, 20) * 64][kind])
    frames = int([rng.integers(1, 40), rng.integers(100, 700), rng.integers(1, 5) * 128 + rng.integers(-2, 3), rng.integers(16, 300)][rng.integers(0, 4)])
    pad = int(rng.choice([0, 0, 0, 1, 3, 4, 8, 12, 64]))
    return lanes, max(frames, 1), pad


def test_random_shapes_both_layouts(gpu):
    rng = np.random.default_rng(2026)
    all_cases = cases(rng)
    seen = set()
    for it in range(220):
        op, cfg, n, words, dt = all_cases[int(rng.integers(0, len(all_cases)))]
        lanes, frames, pad = draw(rng)
        inplace = bool(rng.integers(0, 2))
        if rng.integers(0, 2):
            LMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, frames + pad, inplace)
        else:
            off = int(rng.choice([0, 0, 1, 4]))
            FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, lanes + pad + off, inplace, off=off)
        seen.add(gpu.fn["last_kernel"]().decode().split("<")[0])
    # the draw must have reached the staged kernels and their fall-backs in both layouts
    for want in ("stream_lane_major_staged", "stream_lane_major", "stream_frame_major", "stream_frame_major_staged[16 lanes/wave]"):
        assert any(k.startswith(want) for k in seen), (want, sorted(seen))
