"""Two interchangeable back ends with numpy in / numpy out so every known-answer
and parity case is written once: `OracleBackend` (CPU restatement, checker) and
`GpuBackend` (the HIP engine through its C ABI, device buffers via torch).

Test infrastructure only."""
from __future__ import annotations

import ctypes as C

import numpy as np

from tests import _harness as H


class OracleBackend:
    name = "oracle"

    def __init__(self):
        self.o = H.oracle()

    def stream(self, op, cfg, n, state, x, lanes, frames, layout, inplace=False):
        x = np.ascontiguousarray(x)
        y = x if inplace else np.empty_like(x)
        rc = self.o.stream(op, cfg, n, state, x, y, lanes, frames, layout)
        return rc, y

    def bylane(self, op, coef, frac, n, state, x, lanes, frames, layout, inplace=False):
        """`op` without the `_bylane` suffix; coef = [n, CV, lanes] planes; frac None for floats."""
        x = np.ascontiguousarray(x)
        coef = np.ascontiguousarray(coef)
        y = x if inplace else np.empty_like(x)
        args = (H._ptr(coef),) + (() if frac is None else (frac,)) + (n, H._ptr(state), H._ptr(x), H._ptr(y), lanes, frames, layout)
        return self.o.fn[op + "_bylane"](*args), y

    def cfgcall(self, op, cfg, state, x, y_shape, y_dtype, lanes, frames, layout):
        x = np.ascontiguousarray(x) if x is not None else None
        y = np.empty(y_shape, dtype=y_dtype)
        rc = self.o.cfgcall(op, cfg, state, x, y, lanes, frames, layout)
        return rc, y

    def cossin(self, phases):
        phases = np.ascontiguousarray(phases, dtype=np.int32)
        out = np.empty((phases.size, 2), dtype=np.int32)
        rc = self.o.fn["cossin_i32"](H._ptr(phases), H._ptr(out), phases.size)
        return rc, out

    def atan2(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.int32)
        out = np.empty(xy.size // 2, dtype=np.int32)
        rc = self.o.fn["atan2_i32"](H._ptr(xy), H._ptr(out), out.size)
        return rc, out

    def dds(self, state, lanes, frames, layout):
        out = np.empty(lanes * frames * 2, dtype=np.int32)
        rc = self.o.fn["dds_i32"](H._ptr(state), H._ptr(out), lanes, frames, layout)
        return rc, out

    def helper(self, name, *args):
        return self.o.fn[name](*args)


class GpuBackend:
    name = "hip"

    def __init__(self):
        import torch

        self.torch = torch
        self.e = H.engine()
        self.dev = torch.device("cuda:0")

    # IDSP_TEST_MISALIGN=1: every device buffer starts one element (4 or 8 bytes) into its allocation, so the whole parity
    # suite can be replayed on buffers without 16-byte alignment (tests/test_gpu_misaligned.py does that in a subprocess)
    def _alloc(self, n, dtype):
        import os

        if os.environ.get("IDSP_TEST_MISALIGN"):
            return self.torch.empty(int(n) + 1, dtype=dtype, device=self.dev)[1:]
        return self.torch.empty(int(n), dtype=dtype, device=self.dev)

    def _up(self, a):
        if a is None:
            return None
        a = np.ascontiguousarray(a)
        view = a.view(np.int32) if a.dtype == np.uint32 else a
        host = self.torch.from_numpy(view.copy())
        t = self._alloc(host.numel(), host.dtype)
        t.copy_(host.reshape(-1))
        return t.reshape(host.shape)

    def _down(self, t, like_dtype):
        a = t.cpu().numpy()
        return a.view(like_dtype) if a.dtype != like_dtype else a

    def stream(self, op, cfg, n, state, x, lanes, frames, layout, inplace=False):
        import os

        torch = self.torch
        inplace = inplace or bool(os.environ.get("IDSP_TEST_INPLACE"))  # `Inplace::inplace` for every stream call
        xs = self._up(x)
        ys = xs if inplace else self._alloc(xs.numel(), xs.dtype).reshape(xs.shape)
        if not inplace:
            ys.fill_(-77 if xs.dtype == torch.int32 else float("nan"))  # poison: every element must be written
        ss = self._up(state)
        rc = self.e.stream(op, cfg, n, ss, xs, ys, lanes, frames, layout)
        torch.cuda.synchronize()
        if ss is not None:
            state[...] = self._down(ss, np.uint32).reshape(state.shape)
        return rc, self._down(ys, x.dtype).reshape(np.shape(x))

    def bylane(self, op, coef, frac, n, state, x, lanes, frames, layout, inplace=False):
        import os

        torch = self.torch
        inplace = inplace or bool(os.environ.get("IDSP_TEST_INPLACE"))
        xs, cs, ss = self._up(x), self._up(coef), self._up(state)
        ys = xs if inplace else self._alloc(xs.numel(), xs.dtype).reshape(xs.shape)
        if not inplace:
            ys.fill_(-77 if xs.dtype == torch.int32 else float("nan"))
        args = (H._ptr(cs),) + (() if frac is None else (frac,)) + (n, H._ptr(ss), H._ptr(xs), H._ptr(ys), lanes, frames, layout, None)
        rc = self.e.fn[op + "_bylane"](*args)
        torch.cuda.synchronize()
        if ss is not None:
            state[...] = self._down(ss, np.uint32).reshape(state.shape)
        assert np.array_equal(self._down(cs, coef.dtype).reshape(coef.shape), coef)  # coefficients are read-only
        return rc, self._down(ys, x.dtype).reshape(np.shape(x))

    def cfgcall(self, op, cfg, state, x, y_shape, y_dtype, lanes, frames, layout):
        torch = self.torch
        xs = self._up(x)
        tdt = {np.dtype(np.float32): torch.float32, np.dtype(np.int64): torch.int64}.get(np.dtype(y_dtype), torch.int32)
        ys = self._alloc(int(np.prod(y_shape)), tdt)
        ys.fill_(float("nan") if tdt == torch.float32 else -77)  # poison: every element must be written
        ss = self._up(state)
        rc = self.e.cfgcall(op, cfg, ss, xs, ys, lanes, frames, layout)
        torch.cuda.synchronize()
        state[...] = self._down(ss, np.uint32).reshape(state.shape)
        return rc, self._down(ys, y_dtype).reshape(y_shape)

    def cossin(self, phases):
        torch = self.torch
        ps = self._up(np.ascontiguousarray(phases, dtype=np.int32))
        out = self._alloc(ps.numel() * 2, torch.int32)
        rc = self.e.fn["cossin_i32"](H._ptr(ps), H._ptr(out), ps.numel(), None)
        torch.cuda.synchronize()
        return rc, out.cpu().numpy().reshape(-1, 2)

    def atan2(self, xy):
        torch = self.torch
        xs = self._up(np.ascontiguousarray(xy, dtype=np.int32))
        out = self._alloc(xs.numel() // 2, torch.int32)
        rc = self.e.fn["atan2_i32"](H._ptr(xs), H._ptr(out), out.numel(), None)
        torch.cuda.synchronize()
        return rc, out.cpu().numpy()

    def dds(self, state, lanes, frames, layout):
        torch = self.torch
        ss = self._up(state)
        out = self._alloc(lanes * frames * 2, torch.int32)
        rc = self.e.fn["dds_i32"](H._ptr(ss), H._ptr(out), lanes, frames, layout, None)
        torch.cuda.synchronize()
        state[...] = self._down(ss, np.uint32).reshape(state.shape)
        return rc, out.cpu().numpy()

    def helper(self, name, *args):
        return self.e.fn[name](*args)
