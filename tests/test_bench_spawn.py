"""`bench.py --gpus N` starts N ranks itself.  On CPU: the launcher line is the contract's, and a 2-rank run of
bench.rank_main (started by bench.spawn exactly as bench.py starts its own ranks; tests/_bench_rank_stub.py injects the
CPU oracle where bench.py constructs HipEngine) prints ONE line with n_gpus = group_ranks = 2 (`rccl_ranks` is null: the group is gloo, not RCCL), per-rank kernel medians,
the §8e checksum all-reduce, the weak-scaling headline and the strong-scaling C5 sub-object — whose summed checksums
equal the unsharded oracle run (lanes never meet: dsp-process/src/compose.rs:468-494)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STUB = os.path.join(ROOT, "tests", "_bench_rank_stub.py")


def test_spawn_command_is_the_contract_launcher_line():
    import bench

    cmd = bench.spawn_command(4, ["--gpus", "4", "--steps", "7"])
    assert cmd[1:5] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4"]
    assert cmd[5:7] == ["--master-addr", "127.0.0.1"] and cmd[7] == "--master-port" and 1024 < int(cmd[8]) < 65536
    assert cmd[9] == os.path.join(ROOT, "bench.py") and cmd[10:] == ["--gpus", "4", "--steps", "7"]


def test_main_self_spawns_only_without_a_launcher(monkeypatch):
    import bench

    calls = []
    monkeypatch.setattr(bench, "spawn", lambda n, argv, script=None, **kw: calls.append((n, list(argv))) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    try:
        bench.main(["--gpus", "3", "--steps", "2"])
    except SystemExit as e:
        assert e.code == 0
    assert calls == [(3, ["--gpus", "3", "--steps", "2"])]
    # under an external torchrun the process is a rank: no second launcher
    monkeypatch.setenv("WORLD_SIZE", "3")
    monkeypatch.setattr(bench, "rank_main", lambda args, **kw: calls.append("rank"))
    bench.main(["--gpus", "3"])
    assert calls[-1] == "rank" and len(calls) == 2


def _oracle_sums(kind_name, x, lanes, frames, words, rec_cls, from_sos, *from_args):
    import bench
    from tests import _harness as H

    o = H.oracle()
    q = rec_cls()
    assert o.fn[from_sos]((C.c_double * 6)(*bench.lowpass_sos(bench.F0)), *from_args, C.byref(q)) == 0
    st, y = np.zeros((words, lanes), np.uint32), np.empty_like(x)
    assert o.stream(kind_name, (rec_cls * 1)(q), 1, st, x, y, lanes, frames, H.FM) == 0
    cs = lambda a: bench.wrap64(int(a.reshape(-1).view(np.int32).sum(dtype=np.int64)))  # noqa: E731
    return cs(y), cs(st)


def test_two_ranks_started_by_bench_spawn_print_one_line(tmp_path):
    import bench
    from idsp_amd import _abi

    out = tmp_path / "line.txt"
    argv = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--settle-ms", "0", "--lanes", "192", "--frames", "40",
            "--c5-lanes", "1000", "--c5-frames", "48"]
    env = dict(os.environ, IDSP_BENCH_LAUNCHER="self-spawned torch.distributed.run")
    env.pop("WORLD_SIZE", None)
    with open(out, "w") as fh:
        rc = subprocess.call(bench.spawn_command(2, argv, STUB), stdout=fh, stderr=subprocess.PIPE, env=env, cwd=ROOT,
                             timeout=300)
    assert rc == 0
    lines = [ln for ln in open(out).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["group_ranks"] == 2 and line["rccl_ranks"] is None and line["backend"] == "gloo" and line["ranks_share_device"]
    assert list(line)[-1] == "summary" and line["summary"]["c2"][0] == line["ms_per_step"] and "c5" in line["summary"]
    assert line["scaling"] == "weak" and line["config"]["lanes_per_gpu"] == 192 and line["config"]["lanes_total"] == 384
    assert line["steps"] == 3 and len(line["ranks"]["kernel_ms_median"]) == 2 and line["ranks"]["lanes"] == [192, 192]
    assert line["cpu_baseline"] is None
    # weak scaling: rank r runs its own stream; the all-reduced checksum is the sum of the two oracle runs
    want = [_oracle_sums("biquad_i32_df1", bench.c2_input_host(40, 192, "frame", r), 192, 40, 4, _abi.BiquadI32,
                         "biquad_i32_from_sos", bench.FRAC) for r in range(2)]
    integ = line["integrity"]
    assert [tuple(p) for p in integ["per_rank"]] == want
    assert integ["checksum_y_allreduce"] == bench.wrap64(want[0][0] + want[1][0])
    assert integ["match"] is None  # overridden sizes: nothing on file to compare with
    # strong scaling: the two shards' sums are the unsharded job's
    c5 = line["c5"]
    assert c5["scaling"] == "strong" and c5["n_gpus"] == 2 and c5["ranks"]["lanes"] == [500, 500]
    assert c5["ranks"]["first_lane"] == [0, 500] and c5["config"]["lanes_total"] == 1000
    whole = _oracle_sums("biquad_f32_df2t", bench.c5_input_host(0, 1000, 48, "frame"), 1000, 48, 2, _abi.BiquadF32,
                         "biquad_f32_from_sos_f64")
    assert c5["integrity"]["checksum_y_allreduce"] == whole[0] and c5["integrity"]["checksum_state_sum"] == whole[1]


def test_c5_input_is_the_same_on_numpy_and_torch_and_under_any_split():
    import torch

    import bench

    whole = bench.c5_input_host(0, 700, 33, "frame")
    assert whole.dtype == np.float32 and -1.0 <= whole.min() and whole.max() < 1.0 and abs(float(whole.mean())) < 0.02
    parts = np.concatenate([bench.c5_input_host(lo, n, 33, "frame") for lo, n in ((0, 300), (300, 400))], axis=1)
    assert np.array_equal(whole, parts)
    assert np.array_equal(whole, bench.c5_input(torch, 0, 700, 0, 33, "frame").numpy())
    assert np.array_equal(whole.T, bench.c5_input(torch, 0, 700, 0, 33, "lane").numpy())
    # the last lanes and frames of the full job: the index fills 32 bits exactly
    a = bench.c5_input(np, (1 << 20) - 4, 4, 4094, 4096, "frame")
    assert np.array_equal(a, bench.c5_input(torch, (1 << 20) - 4, 4, 4094, 4096, "frame").numpy())


def test_expected_checksums_on_file_cover_every_rank_and_split():
    import bench

    tab = bench.expected_checksums()
    assert set(tab["c2"]["ranks"]) == {str(r) for r in range(8)}
    assert tab["c5"]["block_lanes"] == bench.C5_BLOCK and len(tab["c5"]["blocks"]) == (1 << 20) // bench.C5_BLOCK
    for world in (1, 2, 4, 8):
        tot = [0, 0]
        for r in range(world):
            lo, n = bench.job_shard(bench.CONFIGS["c5"], r, world)
            e = bench.expected_for("c5", "frame", r, lo, n, False)
            assert e is not None
            tot = [bench.wrap64(tot[0] + e[0]), bench.wrap64(tot[1] + e[1])]
        assert tot == bench.expected_for("c5", "frame", 0, 0, 1 << 20, False)
    assert bench.expected_for("c2", "frame", 3, 0, 65536, False) is not None
    # LaneMajor runs the same tensor transposed and the sums do not depend on the order: one table for both layouts (C5: FrameMajor blocks)
    assert bench.expected_for("c2", "lane", 0, 0, 65536, False) == bench.expected_for("c2", "frame", 0, 0, 65536, False)
    assert bench.expected_for("c5", "lane", 0, 0, 1 << 20, False) is None and bench.expected_for("c2", "frame", 0, 0, 65536, True) is None
    assert np.array_equal(bench.c2_input_host(5, 7, "lane", 0), bench.c2_input_host(5, 7, "frame", 0).T)


def test_c3_c4_inputs_are_the_same_on_host_and_device_code_paths_and_their_checksums_are_on_file():
    """bench.py's C3 / C4 sub-objects: counter-hash inputs (numpy == torch, FRAME_MAJOR == transposed LANE_MAJOR), and the
    oracle checksums tests/golden/make_bench_checksums.py recorded for them."""
    import torch

    import bench

    a = bench.c3_input(np, 3, 5, 2, 6, "frame")
    assert a.shape == (4, 5, 16) and a.dtype == np.float32 and np.array_equal(a, bench.c3_input(torch, 3, 5, 2, 6, "frame").numpy())
    assert np.array_equal(a.transpose(1, 0, 2), bench.c3_input(np, 3, 5, 2, 6, "lane")) and -1.0 <= a.min() and a.max() < 1.0
    b = bench.c4_input(np, 3, 5, 2, 6, "frame")
    assert b.dtype == np.int32 and np.array_equal(b, bench.c4_input(torch, 3, 5, 2, 6, "frame").numpy())
    assert np.array_equal(b.T, bench.c4_input(np, 3, 5, 2, 6, "lane")) and -(1 << 28) <= b.min() and b.max() < (1 << 28)
    assert np.array_equal(bench.c4_steps(np, 7, 9), bench.c4_steps(torch, 7, 9).numpy())
    tab = bench.expected_checksums()
    for name in ("c3", "c4"):
        assert tab[name]["lanes"] == bench.CONFIGS[name]["lanes"] and tab[name]["frames"] == bench.CONFIGS[name]["frames"]
        assert bench.expected_for(name, "frame", 5, 0, bench.CONFIGS[name]["lanes"], False) == bench.expected_for(name, "frame", 0, 0, 1, False)
    assert bench.job_shard(bench.CONFIGS["c3"], 3, 8) == (0, 16384) and bench.job_shard(bench.CONFIGS["c4"], 1, 2) == (0, 32768)
    assert bench.algorithmic_bytes(bench.CONFIGS["c3"], 16384, 4096) == 16384 * 4096 * 68 + 2 * 118 * 4 * 16384
    assert bench.algorithmic_bytes(bench.CONFIGS["c4"], 32768, 4096) == 32768 * 4096 * 12 + 2 * 18 * 4 * 32768


def test_eight_ranks_rendezvous_and_c5_blocks(tmp_path):
    """World size 8 (what the driver's 8-GPU run uses) over gloo on this CPU box with scaled-down lanes: the rendezvous, the
    per-rank C5 shards of lane_shard(lanes, g, 8), the rank-0 JSON assembly and the `summary` tail hold at N = 8."""
    import bench
    from idsp_amd.sharding import lane_shard

    out = tmp_path / "line8.txt"
    argv = ["--gpus", "8", "--steps", "2", "--warmup", "1", "--settle-ms", "0", "--lanes", "64", "--frames", "24",
            "--c5-lanes", "2056", "--c5-frames", "24"]
    env = dict(os.environ, IDSP_BENCH_LAUNCHER="self-spawned torch.distributed.run")
    env.pop("WORLD_SIZE", None)
    with open(out, "w") as fh:
        rc = subprocess.call(bench.spawn_command(8, argv, STUB), stdout=fh, stderr=subprocess.PIPE, env=env, cwd=ROOT, timeout=600)
    assert rc == 0
    lines = [ln for ln in open(out).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["group_ranks"] == 8 and line["rccl_ranks"] is None
    assert line["config"]["lanes_total"] == 8 * 64 and len(line["ranks"]["lanes"]) == 8
    shards = [lane_shard(2056, g, 8) for g in range(8)]
    assert line["c5"]["ranks"]["first_lane"] == [lo for lo, _ in shards]
    assert line["c5"]["ranks"]["lanes"] == [hi - lo for lo, hi in shards] and sum(line["c5"]["ranks"]["lanes"]) == 2056
    assert list(line)[-1] == "summary" and set(line["summary"]) >= {"c2", "c5"}
