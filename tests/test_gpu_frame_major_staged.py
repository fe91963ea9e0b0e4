"""FRAME_MAJOR at lane counts that do not fill the chip: the staged single-wave kernel (`stream_frame_major_staged`,
idsp_amd/csrc/lane_stream.h; 64 / 32 / 16 lanes per wave).  Every biquad-family entry (1- and 2-word samples), lane counts
that leave a partial last wave, frame counts around the 128 / 256 / 512-frame tiles, padded rows, a lane block of a wider
tensor, out of place and in place, against the oracle bit for bit; the kernel taken is asserted through
`idsp_last_kernel()`.  Rows that are not 16-byte aligned must take the other kernels with the same result.
Reference semantics: every frame holds one sample per lane (`View<FrameMajor>`, dsp-process/src/view.rs:10-17), lanes are
independent filters (`Lanes::process`, dsp-process/src/compose.rs:468-494)."""
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
FM = H.FM
FORCED_LW = os.environ.get("IDSP_FM_LANES_PER_WAVE") if os.environ.get("IDSP_DIAG") == "1" else None

# (lanes, frames, pitch): lanes and pitch multiples of 4 (16-byte aligned rows for 4- and 8-byte samples)
SHAPES = [
    (64, 128, 64), (64, 129, 64), (16, 513, 16), (32, 256, 32), (4, 300, 4), (68, 1025, 72), (100, 131, 104), (1000, 77, 1000),
    (132, 40, 260), (36, 16, 36), (8, 17, 8), (2048, 300, 2048), (4100, 130, 4100),
]


def kernel_of(eng):
    return eng.fn["last_kernel"]().decode()


def run_case(eng, op, cfg, n, words, dt, rng, lanes, frames, pitch, inplace, off=0):
    o = H.oracle()
    xh = sample(rng, dt, lanes * frames).reshape(frames, lanes)
    want = np.empty_like(xh)
    st0 = init_state(rng, dt, words * n, lanes)
    so = st0.copy()
    assert o.stream(op, cfg, n, so, xh, want, lanes, frames, FM) == 0
    t = tdtype(dt)
    xb = torch.full((frames * pitch,), SENT, dtype=t, device=DEV)
    xb.view(frames, pitch)[:, off:off + lanes] = torch.from_numpy(xh).to(DEV)
    yb = xb if inplace else torch.full((frames * pitch,), SENT, dtype=t, device=DEV)
    sg = torch.from_numpy(st0.view(np.int32)).to(DEV)
    esz = xb.element_size()
    rc = eng.fn[op + "_pitch"](C.cast(cfg, C.c_void_p), n, p(sg), C.c_void_p(xb.data_ptr() + off * esz), pitch,
                               C.c_void_p(yb.data_ptr() + off * esz), pitch, lanes, frames, FM, None)
    torch.cuda.synchronize()
    assert rc == 0, (op, eng.err())
    yv = yb.view(frames, pitch)
    got = yv[:, off:off + lanes].cpu().numpy()
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (op, lanes, frames, pitch, inplace)
    assert np.array_equal(sg.cpu().numpy().view(np.uint32), so), (op, lanes, frames, "state")
    assert (yv[:, :off] == SENT).all() and (yv[:, off + lanes:] == SENT).all(), (op, "neighbouring lanes / padding must stay untouched")


def test_every_biquad_entry_on_the_staged_kernel(gpu):
    rng = np.random.default_rng(91)
    for op, cfg, n, words, dt in cases(rng):
        for lanes, frames, pitch in SHAPES:
            for inplace in (False, True):
                run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, inplace)
                k = kernel_of(gpu)
                # (chains of more than 4 sections run as passes of up to 4: the name is that of the last, short pass)
                # round 6: from 512 frames the cheap 4-byte processors take the compute + mover pair kernel (tests/test_gpu_frame_major_pair.py)
                from tests.test_gpu_frame_major_pair import PAIR, takes

                if frames >= 512 and pitch % 16 == 0 and FORCED_LW is None and takes(op, n, dt) is not False:
                    assert k.startswith(PAIR) or (takes(op, n, dt) is None and k.startswith("stream_frame_major_staged[")), (op, n, lanes, frames, k)
                else:
                    assert k.startswith("stream_frame_major_staged[") == (frames >= 16), (op, n, lanes, frames, k)


def test_lane_block_of_a_wider_tensor_in_place(gpu):
    """16384 lanes at lane offset 8192 of a 32768-lane tensor, in place (pitch 32768): the neighbours stay untouched."""
    rng = np.random.default_rng(92)
    op, cfg, n, words, dt = cases(rng)[0]
    run_case(gpu, op, cfg, n, words, dt, rng, 16384, 67, 32768, True, off=8192)
    assert kernel_of(gpu).startswith("stream_frame_major_staged["), kernel_of(gpu)
    if FORCED_LW is None:
        assert "[32 lanes/wave]" in kernel_of(gpu)
        # (round 5: from 24576 lanes up the sweep kernel with several frames per segment — fm_sweep.h — on the 64-byte grid and, up to 53248 lanes,
        # off it; 64 lanes per wave of this kernel in between)
        run_case(gpu, op, cfg, n, words, dt, rng, 32768, 40, 32772, False)
        assert kernel_of(gpu).startswith("stream_frame_major_sweep[1 block/workgroup, XCD-contiguous]<"), kernel_of(gpu)
        run_case(gpu, op, cfg, n, words, dt, rng, 20480, 40, 20484, False)
        assert "[32 lanes/wave]" in kernel_of(gpu), kernel_of(gpu)
        run_case(gpu, op, cfg, n, words, dt, rng, 32768, 40, 32768, False)
        assert kernel_of(gpu).startswith("stream_frame_major_sweep[1 block/workgroup]<"), kernel_of(gpu)
        # the 6-section chain: a 4-section pass (COST > 120: staged only around 16384-32768 lanes) and a 2-section pass
        op, cfg, n, words, dt = cases(rng)[1]
        run_case(gpu, op, cfg, n, words, dt, rng, 16384, 30, 16384, False)
        assert "stream_frame_major_staged[32 lanes/wave]" in kernel_of(gpu), kernel_of(gpu)


def test_rows_off_the_16_byte_grid(gpu):
    """Round 3: the staged kernel's 16-byte pieces need dword alignment only (tests/test_gpu_rows_off_16_byte_grid.py); lane counts
    that are not multiples of four still take the register-window kernel below 8192 lanes."""
    rng = np.random.default_rng(93)
    op, cfg, n, words, dt = cases(rng)[0]
    for lanes, frames, pitch in ((130, 200, 130), (1001, 64, 1001)):
        run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, False)
        assert kernel_of(gpu).startswith("stream_frame_major<"), kernel_of(gpu)
    run_case(gpu, op, cfg, n, words, dt, rng, 64, 200, 65, False)
    assert kernel_of(gpu).startswith("stream_frame_major_staged["), kernel_of(gpu)
    run_case(gpu, op, cfg, n, words, dt, rng, 64, 200, 72, False, off=1)  # aligned pitch, base 4 bytes off
    assert kernel_of(gpu).startswith("stream_frame_major_staged["), kernel_of(gpu)


def test_row_pitch_beyond_the_32_bit_offsets_of_the_staged_kernel(gpu):
    """A 64-lane block of a FRAME_MAJOR tensor whose rows are 2^28 + 256 bytes apart: the staged kernel builds per-thread
    byte offsets of up to 15 row pitches in 32 bits, so the launcher must hand such a pitch to the register-window kernel
    (round 2 guarded 2^30 and would have wrapped here).  Same result, neighbours untouched."""
    if FORCED_LW:
        pytest.skip("forced form")
    rng = np.random.default_rng(94)
    op, cfg, n, words, dt = cases(rng)[0]
    pitch = (1 << 26) + 64  # elements: 2^28 + 256 bytes
    run_case(gpu, op, cfg, n, words, dt, rng, 64, 17, pitch, False, off=128)
    assert kernel_of(gpu).startswith("stream_frame_major<"), kernel_of(gpu)
    torch.cuda.empty_cache()
    # just below the guard the staged kernel still runs, and is right
    pitch = (1 << 26) - 64
    run_case(gpu, op, cfg, n, words, dt, rng, 64, 17, pitch, False, off=128)
    assert kernel_of(gpu).startswith("stream_frame_major_staged["), kernel_of(gpu)
    torch.cuda.empty_cache()


@pytest.mark.parametrize("lw", ["64", "32", "16"])
def test_every_lanes_per_wave_form_on_the_ragged_shapes(gpu, lw):
    """The launcher picks the lanes per wave from the lane count; force each form in turn (IDSP_DIAG=1
    IDSP_FM_LANES_PER_WAVE, read once per process) and re-run the tests above on it."""
    if FORCED_LW:
        pytest.skip("already inside a forced run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IDSP_DIAG="1", IDSP_FM_LANES_PER_WAVE=lw, IDSP_NO_FM_PAIR="1")  # (the pair kernel would take the long shapes)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k", "every_biquad or lane_block"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
