"""One rank of bench.py with the CPU oracle standing in for the HIP engine — TEST ONLY.

`bench.spawn(n, argv, script=<this file>)` starts n of these under torch.distributed.run exactly as it starts
bench.py itself; every rank then runs `bench.rank_main` (rendezvous, rank count all-reduce, timed region, MAX
all-reduce, integrity step, checksum all-reduce, C5 sub-object, the JSON line) with `OracleEngine` injected where
bench.py constructs `HipEngine`.  bench.py never refers to this file or to the oracle engine."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


class OracleEngine:
    """bench.HipEngine's interface on host memory (the checker library)."""

    @staticmethod
    def device_setup(local_rank):
        return 0, 0  # no devices: the ranks "share" and meet over gloo

    def __init__(self, cfg_name, cfg, lane_lo, lanes, frames, layout, rank, device_index):
        from idsp_amd import _abi
        from tests import _harness as H

        self.H, self.o = H, H.oracle()
        sos = (C.c_double * 6)(*bench.lowpass_sos(bench.F0))
        if cfg["dtype"] == "i32":
            q = _abi.BiquadI32()
            assert self.o.fn["biquad_i32_from_sos"](sos, bench.FRAC, C.byref(q)) == 0
            self.cfg = (_abi.BiquadI32 * 1)(q)
            self.x = bench.c2_input_host(frames, lanes, layout, rank)
        else:
            q = _abi.BiquadF32()
            assert self.o.fn["biquad_f32_from_sos_f64"](sos, C.byref(q)) == 0
            self.cfg = (_abi.BiquadF32 * 1)(q)
            self.x = bench.c5_input_host(lane_lo, lanes, frames, layout)
        self.x_host = self.x if rank == 0 and cfg["dtype"] == "i32" else None
        self.entry, self.lanes, self.frames = cfg["entry"], lanes, frames
        self.layout = H.FM if layout == "frame" else H.LM
        self.y = np.empty_like(self.x)
        self.state = np.zeros((cfg["state_words"], lanes), np.uint32)

    def step(self):
        assert self.o.stream(self.entry, self.cfg, 1, self.state, self.x, self.y, self.lanes, self.frames, self.layout) == 0

    def sync(self):
        pass

    def timed_steps(self, k):
        ts = []
        for _ in range(k):
            t0 = time.perf_counter()
            self.step()
            ts.append((time.perf_counter() - t0) * 1e3)
        return lambda: ts

    def verify(self, sample_lanes=0):
        self.state[:] = 0
        self.step()
        cs = lambda a: bench.wrap64(int(a.reshape(-1).view(np.int32).sum(dtype=np.int64)))  # noqa: E731
        out = [cs(self.y), cs(self.state)]
        if sample_lanes:
            out.append(cs(np.ascontiguousarray(self.y[:, :sample_lanes] if self.layout == self.H.FM else self.y[:sample_lanes])))
        return out

    def kernel_name(self):
        return "oracle (test stub)"

    def reduce_device(self, backend):
        return "cpu"

    def free(self):
        self.x = self.y = self.state = None


if __name__ == "__main__":
    bench.rank_main(bench.parse_args(), engine_factory=OracleEngine)
