"""LANE_MAJOR on the staged 16-byte-piece kernel (`stream_lane_major_staged`, idsp_amd/csrc/lane_stream.h): every
biquad-family entry (1- and 2-word samples), ragged lane counts (partial last wave), every kind of frame count — whole
512-byte tiles, a partial last tile of whole 16-byte pieces, a scalar remainder of frames % (4 / words) samples — padded
rows, out of place and in place, against the oracle bit for bit; the kernel taken is asserted through
`idsp_last_kernel()`.  Rows that are not 16-byte aligned must fall back to the 4-byte tile kernel with the same result.
Reference semantics: one independent filter per lane over its own contiguous row (`View<LaneMajor>`,
dsp-process/src/view.rs:181-195; `Lanes::process`, dsp-process/src/compose.rs:468-494)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import _harness as H
from tests.test_gpu_pitch import DEV, SENT, cases, init_state, p, sample, tdtype

pytestmark = pytest.mark.gpu
LM = H.LM
# lanes per wave forced by the environment (the last test re-runs this file that way); default: chosen by lane count and cost
FORCED_LW = os.environ.get("IDSP_LM_LANES_PER_WAVE") if os.environ.get("IDSP_DIAG") == "1" else None

# (lanes, frames, pitch): pitch * sizeof(sample) is a multiple of 16 in every case
SHAPES = [
    (64, 128, 128), (64, 256, 256), (1, 128, 128), (63, 384, 384), (65, 129, 132), (130, 131, 132), (257, 77, 80),
    (1000, 513, 516), (64, 1025, 1028), (100, 36, 36), (7, 32, 48), (192, 640, 644), (300, 127, 128), (129, 255, 260),
]


def kernel_of(eng):
    return eng.fn["last_kernel"]().decode()


def run_case(eng, op, cfg, n, words, dt, rng, lanes, frames, pitch, inplace):
    o = H.oracle()
    xh = sample(rng, dt, lanes * frames).reshape(lanes, frames)
    want = np.empty_like(xh)
    st0 = init_state(rng, dt, words * n, lanes)
    so = st0.copy()
    assert o.stream(op, cfg, n, so, xh, want, lanes, frames, LM) == 0
    t = tdtype(dt)
    xb = torch.full((lanes * pitch,), SENT, dtype=t, device=DEV)
    xb.view(lanes, pitch)[:, :frames] = torch.from_numpy(xh).to(DEV)
    yb = xb if inplace else torch.full((lanes * pitch,), SENT, dtype=t, device=DEV)
    sg = torch.from_numpy(st0.view(np.int32)).to(DEV)
    rc = eng.fn[op + "_pitch"](C.cast(cfg, C.c_void_p), n, p(sg), p(xb), pitch, p(yb), pitch, lanes, frames, LM, None)
    torch.cuda.synchronize()
    assert rc == 0, (op, eng.err())
    yv = yb.view(lanes, pitch)
    got = yv[:, :frames].cpu().numpy()
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (op, lanes, frames, pitch, inplace)
    assert np.array_equal(sg.cpu().numpy().view(np.uint32), so), (op, lanes, frames, "state")
    assert (yv[:, frames:] == SENT).all(), (op, "row padding must stay untouched")


def test_every_biquad_entry_on_the_staged_kernel(gpu):
    rng = np.random.default_rng(77)
    for op, cfg, n, words, dt in cases(rng):
        for lanes, frames, pitch in SHAPES:
            for inplace in (False, True):
                run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, inplace)
                k = kernel_of(gpu)
                staged = frames * np.dtype(dt).itemsize >= 128
                assert k.startswith("stream_lane_major_staged" if staged else "stream_lane_major<"), (op, lanes, frames, k)
                if staged and FORCED_LW:
                    assert ("lanes/wave]" in k) == (FORCED_LW != "64"), k  # heavy processors answer 16 with the 32-lane form


def test_cascades_normal_lowpass_on_the_staged_kernel(gpu):
    """The other one-in / one-out per-lane processors of the path that share the LANE_MAJOR launcher."""
    from tests._backends import GpuBackend, OracleBackend
    ob, gb = OracleBackend(), GpuBackend()
    rng = np.random.default_rng(78)
    rows = [(rng.integers(-(1 << 29), 1 << 29, size=5).tolist(), 29) for _ in range(8)]
    frows = [(rng.standard_normal(5) * 0.3).tolist() for _ in range(8)]
    for lanes, frames in ((130, 132), (64, 1028), (1000, 516), (63, 128)):
        xi = rng.integers(-(1 << 31), (1 << 31) - 1, size=lanes * frames, dtype=np.int64).astype(np.int32)
        xf = rng.standard_normal(lanes * frames).astype(np.float32)
        xd = rng.standard_normal(lanes * frames)
        todo = [
            ("cascade_i32_df1", H.biquad_i32(rows), 8, 18, xi), ("cascade_i32_df1", H.biquad_i32(rows[:3]), 3, 8, xi),
            ("cascade_f32_df1", H.biquad_f32(frows[:4]), 4, 10, xf), ("cascade_f64_df1", H.biquad_f64(frows[:2]), 2, 12, xd),
            ("normal_i32_df1", H.biquad_i32(rows[:2]), 2, 8, xi), ("normal_f32_df1", H.biquad_f32(frows[:1]), 1, 4, xf),
        ]
        for op, cfg, n, words, x in todo:
            for inplace in (False, True):
                if x.dtype == np.int32:
                    init = rng.integers(0, 1 << 32, size=(words, lanes), dtype=np.uint64).astype(np.uint32)
                elif x.dtype == np.float32:
                    init = rng.standard_normal(size=(words, lanes)).astype(np.float32).view(np.uint32)
                else:
                    init = init_state(rng, np.float64, words, lanes)
                so, sg = init.copy(), init.copy()
                rco, yo = ob.stream(op, cfg, n, so, x.copy(), lanes, frames, LM, inplace=inplace)
                rcg, yg = gb.stream(op, cfg, n, sg, x.copy(), lanes, frames, LM, inplace=inplace)
                assert rco == 0 and rcg == 0, (op, H.engine().err())
                assert kernel_of(H.engine()).startswith("stream_lane_major_staged"), (op, kernel_of(H.engine()))
                assert np.array_equal(yo.view(np.uint8), yg.view(np.uint8)) and np.array_equal(so, sg), (op, lanes, frames, inplace)
        # Lowpass<N> cascades
        for order, casc in ((1, 1), (2, 2)):
            ks = [[int(v) for v in rng.integers(1 << 16, 1 << 26, size=order)] for _ in range(casc)]
            if order == 2:
                ks = [[k[0], -abs(k[1])] for k in ks]
            cfg = H.lockin_cfg(ks)
            st = rng.integers(0, 1 << 32, size=(2 * order * casc, lanes), dtype=np.uint64).astype(np.uint32)
            so, sg = st.copy(), st.copy()
            rco, yo = ob.cfgcall("lowpass_i32", cfg, so, xi, (lanes * frames,), np.int32, lanes, frames, LM)
            rcg, yg = gb.cfgcall("lowpass_i32", cfg, sg, xi, (lanes * frames,), np.int32, lanes, frames, LM)
            assert rco == 0 and rcg == 0 and np.array_equal(yo, yg) and np.array_equal(so, sg), ("lowpass", order, casc)
            assert kernel_of(H.engine()).startswith("stream_lane_major_staged"), kernel_of(H.engine())


def test_unaligned_rows_fall_back_to_the_tile_kernel(gpu):
    rng = np.random.default_rng(79)
    op, cfg, n, words, dt = cases(rng)[0]
    for lanes, frames, pitch in ((130, 257, 257), (64, 130, 130), (65, 1001, 1003)):
        run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, False)
        assert kernel_of(gpu).startswith("stream_lane_major<"), kernel_of(gpu)
    # aligned pitch, base 4 bytes off a 16-byte boundary
    o = H.oracle()
    lanes, frames = 70, 256
    xh = sample(rng, dt, lanes * frames).reshape(lanes, frames)
    want = np.empty_like(xh)
    so = np.zeros((words * n, lanes), np.uint32)
    assert o.stream(op, cfg, n, so, xh, want, lanes, frames, LM) == 0
    xb = torch.zeros(lanes * frames + 4, dtype=torch.int32, device=DEV)
    xb[1:1 + lanes * frames] = torch.from_numpy(xh.reshape(-1)).to(DEV)
    yb = torch.zeros(lanes * frames, dtype=torch.int32, device=DEV)
    sg = torch.zeros((words * n, lanes), dtype=torch.int32, device=DEV)
    rc = gpu.fn[op](C.cast(cfg, C.c_void_p), n, p(sg), C.c_void_p(xb.data_ptr() + 4), p(yb), lanes, frames, LM, None)
    torch.cuda.synchronize()
    assert rc == 0 and kernel_of(gpu).startswith("stream_lane_major<"), kernel_of(gpu)
    assert np.array_equal(yb.cpu().numpy().reshape(lanes, frames), want)


def test_dds_and_polar_lockin_on_the_staged_kernel(gpu):
    """Processors without an input (DDS: Complex<i32> out, above the I/Q thread-split lane count) and with a pre-stage
    batch (the stream lock-in with `arg` read-out, taken when the frame count is not a whole number of 16-frame batches)."""
    from tests._backends import GpuBackend, OracleBackend
    ob, gb = OracleBackend(), GpuBackend()
    rng = np.random.default_rng(80)
    for lanes, frames in ((41000, 70), (40961, 18)):  # 40960 lanes and fewer split the I / Q arms over two threads
        st = rng.integers(0, 1 << 32, size=(2, lanes), dtype=np.uint64).astype(np.uint32)
        so, sg = st.copy(), st.copy()
        _, yo = ob.dds(so, lanes, frames, LM)
        rc, yg = gb.dds(sg, lanes, frames, LM)
        assert rc == 0 and np.array_equal(yo, yg) and np.array_equal(so, sg), (lanes, frames)
        assert kernel_of(H.engine()).startswith("stream_lane_major_staged"), kernel_of(H.engine())
    lc = H.lockin_cfg([[1 << 20, -(1 << 27)]] * 2)
    # rows of fewer than 32 frames that are not whole batches: the stream lock-in; from 32 frames (round 4) the multi-wave kernel takes the whole
    # batches of every row and the stream kernel only the last frames % 16
    for lanes, frames in ((200, 28), (65, 20), (64, 24), (200, 100), (65, 36), (64, 516)):
        x = rng.integers(-(1 << 28), 1 << 28, size=lanes * frames, dtype=np.int32)
        st = rng.integers(0, 1 << 32, size=(18, lanes), dtype=np.uint64).astype(np.uint32)
        so, sg = st.copy(), st.copy()
        rco, yo = ob.cfgcall("lockin_i32_arg", lc, so, x, (lanes * frames,), np.int32, lanes, frames, LM)
        rcg, yg = gb.cfgcall("lockin_i32_arg", lc, sg, x, (lanes * frames,), np.int32, lanes, frames, LM)
        assert rco == 0 and rcg == 0 and np.array_equal(yo, yg) and np.array_equal(so, sg), (lanes, frames)
        if os.environ.get("IDSP_LOCKIN_NO_LM_TAIL"):  # the round-3 dispatch: the whole call on the stream lock-in (staged kernel from 32 frames)
            want = "stream_lane_major_staged" if frames >= 32 else "stream_lane_major"
        else:
            want = "stream_lane_major" if frames < 32 else "lockin_waves_kernel + stream kernel (last frames % 16)"
        assert kernel_of(H.engine()).startswith(want), kernel_of(H.engine())


def test_polar_lockin_on_the_staged_kernel_without_the_row_split(gpu):
    """`stream_lane_major_staged<LockinPolarProc>` is what default dispatch took for such rows before round 4; keep it meeting the oracle."""
    import subprocess
    import sys

    if os.environ.get("IDSP_LOCKIN_NO_LM_TAIL"):
        pytest.skip("inner run")
    env = dict(os.environ, IDSP_DIAG="1", IDSP_LOCKIN_NO_LM_TAIL="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_lane_major_staged.py", "-m", "gpu", "-x", "-q", "-k", "polar_lockin_on_the_staged_kernel and dds",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("lw", ["64", "32", "16"])
def test_every_lanes_per_wave_form_on_the_ragged_shapes(gpu, lw):
    """The launcher picks 64 / 32 / 16 lanes per wave from the lane count and the processor's cost; the shapes above are all
    small, so force each form in turn (IDSP_DIAG=1 IDSP_LM_LANES_PER_WAVE, read once per process) and re-run the biquad and
    cascade tests of this file: partial last waves, partial tiles and scalar remainders on every form."""
    if FORCED_LW:
        pytest.skip("already inside a forced run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IDSP_DIAG="1", IDSP_LM_LANES_PER_WAVE=lw)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "every_biquad or cascades_normal"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
