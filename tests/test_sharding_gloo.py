"""Multi-GPU path on CPU: world_size-2 gloo run of the lane-sharded driver
logic.  Each rank takes its contiguous lane block (idsp_amd.sharding), runs the
per-shard computation (here the CPU oracle stands in for the engine — tests may
use it) and the ranks only meet in a barrier and an 8-byte checksum all-reduce:
the sharded result must equal the unsharded one lane for lane."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from idsp_amd.sharding import allreduce_checksum, checksum_i64, lane_shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lane_shard_partitions_exactly():
    for lanes in (0, 1, 7, 64, 65536, 1 << 20, 1000003):
        for world in (1, 2, 3, 4, 8):
            blocks = [lane_shard(lanes, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == lanes
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        lane_shard(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lanes, frames, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _harness as H

    o = H.oracle()
    rng = np.random.default_rng(5)  # same stream on every rank: the global tensor
    x = rng.integers(-(1 << 24), 1 << 24, size=(lanes, frames), dtype=np.int32)  # LANE_MAJOR
    cfg = H.biquad_i32([([1 << 28, 1 << 27, 0, 1 << 29, -(1 << 27)], 30)])
    lo, hi = lane_shard(lanes, rank, world)
    xs = np.ascontiguousarray(x[lo:hi])
    ys = np.empty_like(xs)
    st = np.zeros((4, hi - lo), np.uint32)
    dist.barrier()
    assert o.stream("biquad_i32_df1", cfg, 1, st, xs, ys, hi - lo, frames, H.LM) == 0
    dist.barrier()
    total = allreduce_checksum(checksum_i64(torch.from_numpy(ys)))
    np.save(os.path.join(out_dir, f"y{rank}.npy"), ys)
    if rank == 0:
        np.save(os.path.join(out_dir, "sum.npy"), np.array([total], dtype=np.int64))
    dist.destroy_process_group()


def test_two_rank_lane_split_matches_single(tmp_path):
    from tests import _harness as H

    lanes, frames, world = 37, 50, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, lanes, frames, str(tmp_path)), nprocs=world, join=True)
    o = H.oracle()
    rng = np.random.default_rng(5)
    x = rng.integers(-(1 << 24), 1 << 24, size=(lanes, frames), dtype=np.int32)
    cfg = H.biquad_i32([([1 << 28, 1 << 27, 0, 1 << 29, -(1 << 27)], 30)])
    y = np.empty_like(x)
    st = np.zeros((4, lanes), np.uint32)
    assert o.stream("biquad_i32_df1", cfg, 1, st, x, y, lanes, frames, H.LM) == 0
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)])
    assert np.array_equal(got, y)
    assert int(np.load(tmp_path / "sum.npy")[0]) == checksum_i64(torch.from_numpy(y))
