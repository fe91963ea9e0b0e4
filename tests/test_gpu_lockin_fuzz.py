"""Randomised differential test of the three lock-in entries (`Complex<i32>`, `arg`, `norm_sqr`) against the oracle:
random lane / frame counts around the kernels' boundaries (64-lane workgroups, 8- and 16-frame batches, the DMA ring's
three batches in flight), both layouts, every `[Lowpass<N>; K]`, arbitrary state, chunked continuation, and input /
output buffers that start 4, 8 or 12 bytes into an allocation (which must take the paths without 16-byte vectors).
Fixed seed; a few hundred launches."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import _harness as H

pytestmark = pytest.mark.gpu
ENTRIES = (("lockin_i32_process", 2, np.int32, torch.int32), ("lockin_i32_arg", 1, np.int32, torch.int32),
           ("lockin_i32_norm_sqr", 1, np.int64, torch.int64))


def _dev(a):
    return torch.from_numpy(a).cuda()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_lockin_entries_random_shapes(gpu, seed):
    o, e = H.oracle(), H.engine()
    rng = np.random.default_rng(1234 + seed)
    lane_choices = [1, 2, 63, 64, 65, 127, 128, 129, 192, 200, 256, 320]
    frame_choices = [1, 2, 7, 8, 9, 15, 16, 17, 23, 24, 25, 31, 32, 33, 40, 47, 48, 49, 64, 65, 80, 96, 100]
    for it in range(60):
        lanes = int(rng.choice(lane_choices))
        frames = int(rng.choice(frame_choices))
        order, cascade = int(rng.integers(1, 3)), int(rng.integers(1, 5))
        layout = int(rng.integers(0, 2))
        name, width, ndt, tdt = ENTRIES[int(rng.integers(0, 3))]
        ks = [[int(rng.integers(1, 1 << 28))] if order == 1 else [int(rng.integers(1, 1 << 24)), -int(rng.integers(1, 1 << 29))]
              for _ in range(cascade)]
        cfg = H.lockin_cfg(ks)
        words = 2 + 4 * order * cascade
        st = rng.integers(0, 1 << 32, size=(words, lanes), dtype=np.uint64).astype(np.uint32)
        x = rng.integers(-(1 << 31), (1 << 31) - 1, size=lanes * frames, dtype=np.int64).astype(np.int32)
        # oracle, one call
        so = st.copy()
        yo = np.empty(lanes * frames * width, ndt)
        assert o.cfgcall(name, cfg, so, x, yo, lanes, frames, layout) == 0
        # engine: buffers offset by 0 / 4 / 8 / 12 bytes, optionally in two chunks (FrameMajor: a frame split)
        xoff = int(rng.integers(0, 4))                            # elements: 0 / 4 / 8 / 12 bytes
        yoff = int(rng.integers(0, 2 if ndt == np.int64 else 4))  # elements: 0 / 8 bytes for i64, 0 .. 12 bytes for i32
        xbuf = torch.zeros(x.size + 4, dtype=torch.int32, device="cuda")
        xbuf[xoff:xoff + x.size] = _dev(x)
        ybuf = torch.zeros(yo.size + 8, dtype=tdt, device="cuda")
        sg = _dev(st.view(np.int32).copy())
        xv, yv = xbuf[xoff:xoff + x.size], ybuf[yoff:yoff + yo.size]
        split = int(rng.integers(1, frames)) if (layout == H.FM and frames > 1 and rng.random() < 0.5) else None
        if split is None:
            assert e.cfgcall(name, cfg, sg, xv, yv, lanes, frames, layout) == 0, e.err()
        else:
            a = split * lanes
            assert e.cfgcall(name, cfg, sg, xv[:a], yv[:a * width], lanes, split, layout) == 0, e.err()
            assert e.cfgcall(name, cfg, sg, xv[a:], yv[a * width:], lanes, frames - split, layout) == 0, e.err()
        torch.cuda.synchronize()
        ctx = (it, name, lanes, frames, order, cascade, layout, xoff, yoff, split)
        assert np.array_equal(yv.cpu().numpy(), yo), ctx
        assert np.array_equal(sg.cpu().numpy().view(np.uint32), so), ctx
        assert int(ybuf[:yoff].abs().sum()) == 0 and int(ybuf[yoff + yo.size:].abs().sum()) == 0, ("wrote outside y", ctx)
