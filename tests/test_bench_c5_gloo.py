"""`bench.py --config c5` control flow on CPU: world-size-2 gloo run of the strong-scaling lane split.  Each rank
takes the contiguous lane block `job_shard` assigns, holds its own FRAME_MAJOR `[[f32; L/G]; frames]` tensor and steps
it through bench.run_timed() — with the CPU oracle standing in for the HIP engine (tests may use it; bench.py itself
only ever constructs HipEngine).  Checks: the shards partition the job, the ranks meet only in barriers and the MAX
all-reduce, the JSON line carries the strong-scaling fields, and the sharded result equals the unsharded one."""
import ctypes as C
import json
import os
import socket
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LANES, FRAMES, STEPS = 1000, 48, 3


class OracleEngine:
    """bench.HipEngine's interface on host memory (the checker library; tests only)."""

    def __init__(self, cfg, lanes, frames, x):
        from idsp_amd import _abi
        from tests import _harness as H

        self.o = H.oracle()
        q = _abi.BiquadF32()
        import bench

        assert self.o.fn["biquad_f32_from_sos_f64"]((C.c_double * 6)(*bench.lowpass_sos(bench.F0)), C.byref(q)) == 0
        self.cfg = (_abi.BiquadF32 * 1)(q)
        self.lanes, self.frames, self.x = lanes, frames, np.ascontiguousarray(x)
        self.y = np.empty_like(self.x)
        self.state = np.zeros((cfg["state_words"], lanes), np.uint32)
        self.outputs = []

    def step(self):
        from tests import _harness as H

        assert self.o.stream("biquad_f32_df2t", self.cfg, 1, self.state, self.x, self.y, self.lanes, self.frames, H.FM) == 0
        self.outputs.append(self.y.copy())

    def sync(self):
        pass

    def timed_steps(self, k):
        ts = []
        for _ in range(k):
            t0 = time.perf_counter()
            self.step()
            ts.append((time.perf_counter() - t0) * 1e3)
        return lambda: ts

    def kernel_name(self):
        return "oracle"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_input():
    return np.random.default_rng(5).standard_normal((FRAMES, LANES)).astype(np.float32)  # FRAME_MAJOR, all lanes


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench

    cfg = bench.CONFIGS["c5"]
    lo, n = bench.job_shard(cfg, rank, world, LANES)
    eng = OracleEngine(cfg, n, FRAMES, _global_input()[:, lo:lo + n])
    elapsed, kern_ms, untimed = bench.run_timed(eng, STEPS, 2, 0.0, dist)
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() >= elapsed and len(kern_ms) == STEPS and untimed == 2
    np.save(os.path.join(out_dir, f"y{rank}.npy"), np.stack(eng.outputs))
    np.save(os.path.join(out_dir, f"shard{rank}.npy"), np.array([lo, n]))
    if rank == 0:
        args = bench.argparse.Namespace(steps=STEPS, warmup=2, settle_ms=0.0, layout="frame", lanes=LANES)
        line = bench.report("c5", cfg, args, world, n, FRAMES, float(t.item()), kern_ms, untimed, eng.kernel_name(), LANES)
        with open(os.path.join(out_dir, "line.json"), "w") as f:
            json.dump(line, f)
    dist.barrier()
    dist.destroy_process_group()


def test_c5_mode_two_ranks_strong_scaling(tmp_path):
    sys.path.insert(0, ROOT)
    import bench

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    shards = [np.load(tmp_path / f"shard{r}.npy") for r in range(world)]
    assert shards[0][0] == 0 and shards[0][0] + shards[0][1] == shards[1][0] and shards[1][0] + shards[1][1] == LANES
    line = json.load(open(tmp_path / "line.json"))
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["dtype"] == "f32" and line["steps"] == STEPS
    assert line["config"]["lanes_total"] == LANES and line["config"]["lanes_per_gpu"] == shards[0][1]
    assert line["roofline"]["algorithmic_bytes"] == bench.algorithmic_bytes(bench.CONFIGS["c5"], int(shards[0][1]), FRAMES)
    assert abs(line["value"] - LANES * FRAMES * STEPS / (line["ms_per_step"] * 1e-3 * STEPS) / 1e6) < 1e-2 * line["value"]
    # numerics: the two shards side by side are the unsharded run (all steps: the state carries across steps)
    whole = OracleEngine(bench.CONFIGS["c5"], LANES, FRAMES, _global_input())
    for _ in range(2 + STEPS):
        whole.step()
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)], axis=2)
    assert np.array_equal(got.view(np.uint32), np.stack(whole.outputs).view(np.uint32))


def test_weak_and_strong_job_shards():
    sys.path.insert(0, ROOT)
    import bench

    for world in (1, 2, 4, 8):
        strong = [bench.job_shard(bench.CONFIGS["c5"], r, world) for r in range(world)]
        assert sum(n for _, n in strong) == 1 << 20 and all(n == (1 << 20) // world for _, n in strong)
        assert [lo for lo, _ in strong] == [r * ((1 << 20) // world) for r in range(world)]
        weak = [bench.job_shard(bench.CONFIGS["c2"], r, world) for r in range(world)]
        assert all(n == 65536 for _, n in weak)


def test_by_direction_roof_adds_up():
    """`roofline.by_direction` (bench.py): read + written bytes = the algorithmic bytes of the line; read time + write time at the committed skeleton
    rates (profiles/r06_ubench_cu_ceiling.txt); SURVEY 8(d) bytes per unit: biquads 4 + 4, HbfDec /16 64 + 4 per output frame, lock-in 4 + 8."""
    sys.path.insert(0, ROOT)
    import bench

    want_written = {"c2": 4, "c3": 4, "c4": 8, "c5": 4}
    for name, w in want_written.items():
        cfg = bench.CONFIGS[name]
        lanes, frames = cfg["lanes"], cfg["frames"]
        bd = bench.by_direction(name, cfg, lanes, frames, 1.0)
        assert bd["read_bytes"] + bd["written_bytes"] == bench.algorithmic_bytes(cfg, lanes, frames), name
        assert bd["written_bytes"] == lanes * frames * w + cfg["state_words"] * 4 * lanes, name
        ms = (bd["read_bytes"] / bd["read_gbs"] + bd["written_bytes"] / bd["write_gbs"]) / 1e6
        assert abs(bd["sum_of_directions_ms"] - ms) < 1e-3 and abs(bd["kernel_ms_over_it"] - 1.0 / ms) < 1e-3, name
    assert bench.by_direction("c2", bench.CONFIGS["c2"], 65536, 4096, 0.0) is None
    # the committed rates are those of the microbenchmark's 1024-wave dense lines
    txt = open(os.path.join(ROOT, "profiles", "r06_ubench_cu_ceiling.txt")).read()
    rows = [json.loads(l) for l in txt.splitlines() if l.startswith("{")]
    rd = [r["TB/s"] for r in rows if r["pattern"] == "dense" and r["mode"] == "read" and r["waves"] == 1024 and r["slot_us"] == 0]
    wr = [r["TB/s"] for r in rows if r["pattern"] == "dense" and r["mode"] == "write" and r["waves"] == 1024 and r["slot_us"] == 0 and r.get("store", "nt x4") == "nt x4"]
    assert rd and wr and abs(rd[0] * 1e3 - bench.READ_GBS_FILE) < 150 and abs(wr[0] * 1e3 - bench.WRITE_GBS_FILE) < 150
