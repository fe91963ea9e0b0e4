"""`idsp_multi_*`: the single-process lane split (include/idsp_hip.h).  The test box has ONE GPU, so the split is
exercised with several lane blocks that share device 0 (the ABI allows a device to appear more than once): block
bounds, per-block buffers, scatter / gather, the two headline operators and the generic for_each driver must give the
unsplit oracle result lane for lane."""
import ctypes as C

import numpy as np
import pytest

from idsp_amd import _abi
from tests import _harness as H

pytestmark = pytest.mark.gpu
LM, FM = H.LM, H.FM


def make(gpu, devices):
    m = C.c_void_p()
    arr = (C.c_int * len(devices))(*devices) if devices is not None else None
    rc = gpu.fn["multi_create"](arr, len(devices) if devices is not None else 0, C.byref(m))
    return rc, m


def test_shards_partition_and_errors(gpu):
    from idsp_amd.sharding import lane_shard

    rc, m = make(gpu, [0, 0, 0])
    assert rc == 0 and gpu.fn["multi_size"](m) == 3 and gpu.fn["multi_device"](m, 2) == 0
    assert gpu.fn["multi_stream"](m, 0) and gpu.fn["multi_stream"](m, 0) != gpu.fn["multi_stream"](m, 1)
    for lanes in (0, 1, 2, 1000, 65536, (1 << 20) + 1):
        for g in range(3):
            lo, hi = C.c_size_t(), C.c_size_t()
            assert gpu.fn["multi_shard"](m, lanes, g, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == lane_shard(lanes, g, 3)
    assert gpu.fn["multi_shard"](m, 10, 3, C.byref(C.c_size_t()), C.byref(C.c_size_t())) == _abi.IDSP_EINVAL
    assert gpu.fn["multi_destroy"](m) == 0
    rc, _ = make(gpu, [0, 99])
    assert rc == _abi.IDSP_EINVAL and b"99" in gpu.fn["last_error"]()
    rc, m = make(gpu, None)  # every visible device
    assert rc == 0 and gpu.fn["multi_size"](m) == gpu.fn["device_count"]()
    assert gpu.fn["multi_destroy"](m) == 0


@pytest.mark.parametrize("blocks", [1, 2, 5])
def test_split_biquads_equal_the_unsplit_oracle(gpu, blocks):
    o = H.oracle()
    rng = np.random.default_rng(blocks)
    lanes, frames = 1003, 77
    rc, m = make(gpu, [0] * blocks)
    assert rc == 0
    cases = [("biquad_i32_df1", "multi_biquad_i32_df1", H.biquad_i32([(rng.integers(-(1 << 29), 1 << 29, size=5).tolist(), 30)]), 4, np.int32),
             ("biquad_f32_df2t", "multi_biquad_f32_df2t", H.biquad_f32([(rng.standard_normal(5) * 0.3).tolist()]), 2, np.float32)]
    for op, mop, cfg, words, dt in cases:
        x = (rng.integers(-(1 << 31), (1 << 31) - 1, size=(lanes, frames), dtype=np.int64).astype(np.int32) if dt == np.int32
             else rng.standard_normal((lanes, frames)).astype(np.float32))  # LANE_MAJOR: lane blocks are contiguous pieces
        want = np.empty_like(x)
        so = np.zeros((words, lanes), np.uint32)
        assert o.stream(op, cfg, 1, so, x, want, lanes, frames, LM) == 0
        P = C.c_void_p * blocks
        xs, ys, ss = P(), P(), P()
        assert gpu.fn["multi_alloc"](m, lanes, frames * 4, xs) == 0
        assert gpu.fn["multi_alloc"](m, lanes, frames * 4, ys) == 0
        # the state of a block is its own [words, block lanes] plane set: allocate per block
        assert gpu.fn["multi_alloc"](m, lanes, words * 4, ss) == 0
        assert gpu.fn["multi_copy"](m, lanes, frames * 4, xs, x.ctypes.data, 1) == 0
        assert gpu.fn[mop](m, C.cast(cfg, C.c_void_p), 1, ss, xs, ys, lanes, frames, LM) == 0, gpu.err()
        got = np.empty_like(x)
        assert gpu.fn["multi_copy"](m, lanes, frames * 4, ys, got.ctypes.data, 0) == 0
        assert gpu.fn["multi_sync"](m) == 0
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (op, blocks)
        # gather the states block by block: planes of each block are [words][block lanes]
        st = np.zeros((words, lanes), np.uint32)
        for g in range(blocks):
            lo, hi = C.c_size_t(), C.c_size_t()
            gpu.fn["multi_shard"](m, lanes, g, C.byref(lo), C.byref(hi))
            n = hi.value - lo.value
            blk = np.empty((words, n), np.uint32)
            assert gpu.fn["device_d2h"](blk.ctypes.data, ss[g], blk.nbytes, gpu.fn["multi_stream"](m, g)) == 0
            assert gpu.fn["multi_sync"](m) == 0
            st[:, lo.value:hi.value] = blk
        assert np.array_equal(st, so), (op, "state")
        for ptrs in (xs, ys, ss):
            assert gpu.fn["multi_free"](m, ptrs) == 0
    assert gpu.fn["multi_destroy"](m) == 0


def test_for_each_drives_any_entry_point(gpu):
    """The generic driver with a host callback: FRAME_MAJOR clamp biquad on per-block `[[i32; L/G]; frames]` tensors."""
    o = H.oracle()
    rng = np.random.default_rng(3)
    lanes, frames, blocks = 640, 50, 3
    cfg = H.biquad_clamp_i32([(rng.integers(-(1 << 29), 1 << 29, size=5).tolist(), 29, 5, -(1 << 27), 1 << 27)])
    x = rng.integers(-(1 << 31), (1 << 31) - 1, size=(frames, lanes), dtype=np.int64).astype(np.int32)  # FRAME_MAJOR, all lanes
    want = np.empty_like(x)
    so = np.zeros((4, lanes), np.uint32)
    assert o.stream("biquad_i32_df1_clamp", cfg, 1, so, x, want, lanes, frames, FM) == 0
    rc, m = make(gpu, [0] * blocks)
    assert rc == 0
    import torch

    bufs = {}

    def body(user, index, lo, hi, stream):
        xb = torch.from_numpy(np.ascontiguousarray(x[:, lo:hi])).to("cuda:0")
        yb = torch.empty_like(xb)
        sb = torch.zeros((4, hi - lo), dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        bufs[index] = (lo, hi, xb, yb, sb)
        return gpu.fn["biquad_i32_df1_clamp"](C.cast(cfg, C.c_void_p), 1, C.c_void_p(sb.data_ptr()), C.c_void_p(xb.data_ptr()),
                                              C.c_void_p(yb.data_ptr()), hi - lo, frames, FM, stream)

    cb = _abi.SHARD_FN(body)
    assert gpu.fn["multi_for_each"](m, lanes, cb, None) == 0, gpu.err()
    assert gpu.fn["multi_sync"](m) == 0
    got = np.empty_like(x)
    st = np.zeros_like(so)
    for lo, hi, xb, yb, sb in bufs.values():
        got[:, lo:hi] = yb.cpu().numpy()
        st[:, lo:hi] = sb.cpu().numpy().view(np.uint32)
    assert len(bufs) == blocks and np.array_equal(got, want) and np.array_equal(st, so)
    # a failing callee stops the loop and its status comes back
    bad = _abi.SHARD_FN(lambda user, index, lo, hi, stream: _abi.IDSP_EINVAL if index == 1 else 0)
    assert gpu.fn["multi_for_each"](m, lanes, bad, None) == _abi.IDSP_EINVAL
    assert gpu.fn["multi_last_block"]() == 1
    # ... and so does a positive "stop" code of the callback's own; a run through resets the block index
    seen = []
    stop = _abi.SHARD_FN(lambda user, index, lo, hi, stream: seen.append(index) or (7 if index == 2 else 0))
    assert gpu.fn["multi_for_each"](m, lanes, stop, None) == 7 and seen == [0, 1, 2] and gpu.fn["multi_last_block"]() == 2
    ok = _abi.SHARD_FN(lambda user, index, lo, hi, stream: 0)
    assert gpu.fn["multi_for_each"](m, lanes, ok, None) == 0 and gpu.fn["multi_last_block"]() == -1
    assert gpu.fn["multi_destroy"](m) == 0


def test_headline_operators_validate_every_block_before_launching_any(gpu):
    """A bad LAST block (NULL buffer) or a bad configuration is reported with nothing launched: the earlier blocks'
    state and output stay untouched (round 2 returned after launching the blocks before the bad one)."""
    import torch

    rng = np.random.default_rng(4)
    lanes, frames, blocks = 300, 40, 3
    rc, m = make(gpu, [0] * blocks)
    assert rc == 0
    cfg = H.biquad_i32([(rng.integers(-(1 << 29), 1 << 29, size=5).tolist(), 30)])
    xs, ys, ss = [], [], []
    for g in range(blocks):
        lo, hi = C.c_size_t(), C.c_size_t()
        gpu.fn["multi_shard"](m, lanes, g, C.byref(lo), C.byref(hi))
        n = hi.value - lo.value
        xs.append(torch.randint(-1000, 1000, (n * frames,), dtype=torch.int32, device="cuda:0"))
        ys.append(torch.full((n * frames,), 123, dtype=torch.int32, device="cuda:0"))
        ss.append(torch.full((4, n), 5, dtype=torch.int32, device="cuda:0"))
    torch.cuda.synchronize()
    arr = lambda ts, null=None: (C.c_void_p * blocks)(*[None if i == null else t.data_ptr() for i, t in enumerate(ts)])  # noqa: E731
    args = lambda **kw: (m, C.cast(kw.get("cfg", cfg), C.c_void_p), 1, arr(ss), arr(xs), arr(ys, kw.get("null")), lanes, frames, LM)  # noqa: E731
    assert gpu.fn["multi_biquad_i32_df1"](*args(null=2)) == _abi.IDSP_EINVAL and gpu.fn["multi_last_block"]() == 2
    bad = H.biquad_i32([([1, 2, 3, 4, 5], 32)])  # F = 32: `const assert!(F < 32)`
    assert gpu.fn["multi_biquad_i32_df1"](*args(cfg=bad)) == _abi.IDSP_EINVAL and gpu.fn["multi_last_block"]() == 0
    assert gpu.fn["multi_sync"](m) == 0
    for y, st in zip(ys, ss):
        assert bool((y == 123).all()) and bool((st == 5).all()), "a block was launched before validation finished"
    assert gpu.fn["multi_biquad_i32_df1"](*args()) == 0 and gpu.fn["multi_sync"](m) == 0 and gpu.fn["multi_last_block"]() == -1
    assert not bool((ys[0] == 123).all())
    # overflow of lanes * bytes_per_lane and a NULL block buffer in the copy helper
    ptrs = (C.c_void_p * blocks)()
    assert gpu.fn["multi_alloc"](m, 1 << 40, 1 << 40, ptrs) == _abi.IDSP_EINVAL
    host = np.zeros(lanes, np.int32)
    assert gpu.fn["multi_copy"](m, lanes, 4, (C.c_void_p * blocks)(), host.ctypes.data, 1) == _abi.IDSP_EINVAL
    assert gpu.fn["device_sync"]() == 0
    assert gpu.fn["multi_destroy"](m) == 0
