"""FRAME_MAJOR rows that are only 4-byte aligned (round 3): dense tensors whose lane count is not a multiple of four (row f of a
65537-lane tensor starts 4 f bytes off the 16-byte grid), odd row pitches, base pointers one or three words into a row.  The
reference takes any N in `Lanes<C>` and any slice (dsp-process/src/compose.rs:468-476, view.rs:10-17); round 2 ran all of these on
the register-window kernel's 4-byte accesses (0.39-0.41 of the HBM peak at 65537 lanes).  Now the 16-byte kernels run on such rows
as they are — `global_load_lds_dwordx4` / `global_load_dwordx4` / 16-byte stores need dword alignment only — and the last
`lanes % 4` lanes run beside them on a second stream (idsp_amd/csrc/lane_stream.h, launch_stream).  Against the oracle bit for
bit (outputs, state), out of place and in place, the bytes between the rows untouched, the kernels taken asserted."""
import numpy as np
import pytest

from tests import test_gpu_frame_major_staged as FMS
from tests.test_gpu_pitch import cases
from tests.test_gpu_ragged_lds_block import kernel_of, lds_cases

pytestmark = pytest.mark.gpu
ODD = " + stream_frame_major_few (lanes % 4, second stream)"
# 24576 ... 53248 lanes on rows off the grid (round 5, dispatch_thresholds.h kSweepOffGridSmallMax): the sweep kernel with several frames per segment for 4-byte
# outputs, the staged kernel for the rest
SMALL = ("stream_frame_major_sweep[1 block/workgroup, XCD-contiguous]<", "stream_frame_major_staged[64 lanes/wave]<")


def test_lane_counts_that_are_not_multiples_of_four(gpu):
    rng = np.random.default_rng(311)
    cs = lds_cases(rng)
    # (lanes, frames, pitch, lane offset, kernel of the lanes - lanes % 4 body)
    # (round 5: beyond 98304 lanes — where this kernel would walk panels on a persistent grid — the body runs on the dense-sweep kernel, its
    # blocks dealt to the XCDs in contiguous eighths on rows off the 64-byte grid: fm_sweep.h)
    shapes = [(65537, 40, 65537, 0, "stream_frame_major_lds[XCD-contiguous blocks]<"),
              (65539, 17, 65543, 3, "stream_frame_major_lds[XCD-contiguous blocks]<"),
              (65001, 21, 65001, 0, "stream_frame_major_lds[XCD-contiguous blocks]<"),
              (8195, 50, 8195, 0, "stream_frame_major_staged[32 lanes/wave]<"),
              (32770, 64, 32771, 1, SMALL),
              (73731, 24, 73731, 0, "stream_frame_major_lds + stream_frame_major_staged (remainder, second stream)<"),
              (131073, 16, 131073, 0, "stream_frame_major_sweep[2 blocks/workgroup, XCD-contiguous]<")]
    for i, (lanes, frames, pitch, off, body) in enumerate(shapes):
        for j, (op, cfg, n, words, dt) in enumerate(cs):
            if (i + j) % 3:
                continue
            FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, bool((i + j) & 1), off=off)
            k = kernel_of(gpu)
            assert k.startswith(body) and k.endswith(ODD), (op, lanes, pitch, off, k)  # (str.startswith takes a tuple of alternatives)


def test_whole_pieces_on_rows_off_the_grid(gpu):
    """lane counts that are multiples of four with odd pitches / odd base offsets: no second stream, 16-byte kernels"""
    rng = np.random.default_rng(312)
    cs = lds_cases(rng)
    shapes = [(65536, 33, 65537, 0, "stream_frame_major_lds[XCD-contiguous blocks]<"),
              (65536, 20, 65544, 1, "stream_frame_major_lds[XCD-contiguous blocks]<"),
              (65000, 19, 65003, 2, "stream_frame_major_lds[XCD-contiguous blocks]<"),
              (16384, 130, 16387, 0, "stream_frame_major_staged[32 lanes/wave]<"),  # round 4: plain accesses on such rows, 64 from 25600 lanes
              (28672, 40, 28675, 1, SMALL),
              (4096, 257, 4099, 3, "stream_frame_major_staged[16 lanes/wave]<"),
              (69632, 18, 69633, 0, "stream_frame_major_lds + stream_frame_major_staged (remainder, second stream)<")]
    for i, (lanes, frames, pitch, off, want) in enumerate(shapes):
        for j, (op, cfg, n, words, dt) in enumerate(cs):
            if (i + j) % 3:
                continue
            FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, bool((i + j) & 1), off=off)
            k = kernel_of(gpu)
            assert k.startswith(want) and not k.endswith(ODD), (op, lanes, pitch, off, k)


def test_eight_byte_samples_and_chains_on_odd_pitches(gpu):
    """f64 entries (8-byte samples: rows 8-byte aligned) and multi-section chains on the staged kernel with an odd pitch"""
    rng = np.random.default_rng(313)
    for op, cfg, n, words, dt in cases(rng):
        if dt != np.float64 and n <= 2:
            continue
        FMS.run_case(gpu, op, cfg, n, words, dt, rng, 8192, 37, 8193, False, off=0)
        FMS.run_case(gpu, op, cfg, n, words, dt, rng, 8192, 37, 8195, True, off=1)


def test_few_lanes_keep_the_register_window_kernel(gpu):
    rng = np.random.default_rng(314)
    op, cfg, n, words, dt = lds_cases(rng)[0]
    for lanes, frames, pitch in ((130, 200, 130), (1001, 64, 1001), (8193, 12, 8193)):  # body below 8192 lanes / fewer than 16 frames
        FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, False)
        assert kernel_of(gpu).startswith("stream_frame_major<"), kernel_of(gpu)


def test_one_to_three_lanes_in_all(gpu):
    """`stream_frame_major_few` as the whole launch: every 4-byte biquad entry, 1 .. 3 lanes, frame counts around the 256-frame
    tiles and the 8-frame groups, dense and as a lane block of a wider tensor, in place and out of place."""
    rng = np.random.default_rng(315)
    for i, (op, cfg, n, words, dt) in enumerate(cases(rng)):
        if dt == np.float64:
            continue
        for j, (lanes, frames, pitch, off) in enumerate(((1, 64, 1, 0), (2, 255, 2, 0), (3, 256, 3, 0), (3, 1031, 7, 2), (1, 4096, 5, 4), (2, 777, 64, 61))):
            if (i + j) % 2:
                continue
            FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, bool(j & 1), off=off)
            assert kernel_of(gpu).startswith("stream_frame_major_few<"), (op, kernel_of(gpu))
    op, cfg, n, words, dt = lds_cases(rng)[0]
    FMS.run_case(gpu, op, cfg, n, words, dt, rng, 3, 63, 3, False)  # fewer than 64 frames: the register-window kernel
    assert kernel_of(gpu).startswith("stream_frame_major<"), kernel_of(gpu)


def test_bylane_banks_on_lane_counts_that_are_not_multiples_of_four(gpu):
    """`ByLane` banks (dsp-process/src/compose.rs:363-390): the last lanes % 4 lanes read THEIR columns of the coefficient and
    state planes (plane pitch = the call's lane count) on the few-lanes kernel."""
    from tests import _bylane_cases as B
    from tests._backends import GpuBackend, OracleBackend

    ob, gb = OracleBackend(), GpuBackend()
    rng = np.random.default_rng(316)
    for i, (op, dtype, words, clamp) in enumerate(B.OPS):
        if np.dtype(dtype).itemsize != 4:
            continue
        for n, lanes, frames in ((1, 8195, 37), (2, 65537, 18), (1, 3, 300), (2, 41003, 33)):
            if (i + lanes) % 2:
                continue
            frac = 29 if dtype == np.int32 else None
            coef = B.coef_planes(rng, dtype, n, lanes, clamp, frac or 0)
            x = B.samples(rng, dtype, lanes * frames)
            init = B.init_state(rng, dtype, words * n, lanes)
            so, sg = init.copy(), init.copy()
            rco, yo = ob.bylane(op, coef, frac, n, so, x.copy(), lanes, frames, B.FM)
            rcg, yg = gb.bylane(op, coef, frac, n, sg, x.copy(), lanes, frames, B.FM)
            assert rco == 0 and rcg == 0, (op, gpu.err())
            k = kernel_of(gpu)
            assert k.endswith(ODD) or k.startswith("stream_frame_major_few<"), (op, n, lanes, k)
            assert np.array_equal(B.bits(yo), B.bits(yg)) and np.array_equal(so, sg), (op, n, lanes)
