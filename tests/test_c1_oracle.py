"""BASELINE.json configs[0] ("C1") on the CPU side: a single lane, 2^20 samples, i32 DF1 biquad, seed 1 —
the reference's `SplitProcess::block` / `SplitInplace::inplace` plumbing (dsp-process/src/process.rs:122-127,137-141)
with no GPU involved.  The oracle at full size: block == in place == any chunking (streaming state), both layouts
coincide for one lane, and the first 2^15 samples equal the independent Python spec model.  The GPU twin of this
test is tests/test_gpu_fullsize.py::test_c1_single_lane_one_million_samples."""
import ctypes as C

import numpy as np

from idsp_amd import _abi
from oracle import spec
from tests import _harness as H


def test_c1_oracle_plumbing_at_full_size():
    frames = 1 << 20
    o = H.oracle()
    q = _abi.BiquadI32()
    assert o.fn["biquad_i32_from_sos"]((C.c_double * 6)(*o.lowpass_sos(0.01)), 30, C.byref(q)) == 0
    cfg = (_abi.BiquadI32 * 1)(q)
    x = np.random.default_rng(1).integers(-(1 << 24), 1 << 24, frames, dtype=np.int32)
    y = np.empty_like(x)
    st = np.zeros((4, 1), np.uint32)
    assert o.stream("biquad_i32_df1", cfg, 1, st, x, y, 1, frames, H.LM) == 0
    # gain <= 1 lowpass on +-2^24 inputs: no overflow anywhere, so debug and release builds of the reference agree
    assert np.abs(y.astype(np.int64)).max() < 1 << 25
    # FRAME_MAJOR is the same memory for one lane
    y2, st2 = np.empty_like(x), np.zeros((4, 1), np.uint32)
    assert o.stream("biquad_i32_df1", cfg, 1, st2, x, y2, 1, frames, H.FM) == 0
    assert np.array_equal(y, y2) and np.array_equal(st, st2)
    # `inplace` == `block`
    xi, st3 = x.copy(), np.zeros((4, 1), np.uint32)
    assert o.stream("biquad_i32_df1", cfg, 1, st3, xi, xi, 1, frames, H.LM) == 0
    assert np.array_equal(xi, y) and np.array_equal(st, st3)
    # ragged chunks continue the stream exactly
    y4, st4 = np.empty_like(x), np.zeros((4, 1), np.uint32)
    cuts = [0, 1, 2, 1000, 65537, 700001, frames]
    for a, b in zip(cuts, cuts[1:]):
        assert o.stream("biquad_i32_df1", cfg, 1, st4, x[a:b], y4[a:b], 1, b - a, H.LM) == 0
    assert np.array_equal(y4, y) and np.array_equal(st, st4)
    # independent restatement (exact Python ints, reference-shaped state) on a prefix
    s = spec.DirectForm1()
    ba = list(q.ba)
    want = [spec.biquad_i32_df1(ba, 30, s, int(v)) for v in x[: 1 << 15]]
    assert np.array_equal(y[: 1 << 15], np.array(want, np.int32))
    # final state = the last two inputs and outputs (biquad.rs:260-269 layout: x0 x1 y0 y1)
    assert st[:, 0].view(np.int32).tolist() == [int(x[-1]), int(x[-2]), int(y[-1]), int(y[-2])]
