"""Chains of 4 .. 8 biquad sections on the two-wave kernel (`stream_frame_major_duo`, idsp_amd/csrc/lane_stream.h): FRAME_MAJOR
launches from 40960 lanes split the serial chain over two waves per 64 lanes that hand the samples over through LDS — five to
eight sections of a slice composition (`[C] x [S]`, dsp-process/src/compose.rs:43-77) in ONE pass, and the `Cascade` form
(src/iir/biquad.rs:339-364) with wave 1 started 2 NA values into the state record.  Against the oracle bit for bit: outputs,
written-back state, out of place and in place, frame counts around the 32-frame tiles, a ragged last workgroup, lane blocks
of wider tensors."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import _harness as H
from tests import test_gpu_frame_major_staged as FMS
from tests.test_gpu_pitch import DEV, SENT, init_state, p, sample, tdtype

pytestmark = pytest.mark.gpu
FM = H.FM


def kernel_of(eng):
    return eng.fn["last_kernel"]().decode()


def chain_cases(rng, n):
    ri = [(rng.integers(-(1 << 29), 1 << 29, size=5).tolist(), 29) for _ in range(n)]
    rc = [(r[0], 29, 77, -(1 << 28), 1 << 28) for r in ri]
    rf = [(rng.standard_normal(5) * 0.3).tolist() for _ in range(n)]
    rfc = [(r, 0.01, -0.5, 0.6) for r in rf]
    return [("biquad_i32_df1", H.biquad_i32(ri), 4, np.int32), ("biquad_i32_df1_clamp", H.biquad_clamp_i32(rc), 4, np.int32),
            ("biquad_i32_dither", H.biquad_i32(ri), 5, np.int32), ("biquad_i32_dither_clamp", H.biquad_clamp_i32(rc), 5, np.int32),
            ("biquad_i32_wide", H.biquad_i32(ri), 6, np.int32), ("biquad_i32_wide_clamp", H.biquad_clamp_i32(rc), 6, np.int32),
            ("biquad_f32_df1", H.biquad_f32(rf), 4, np.float32), ("biquad_f32_df1_clamp", H.biquad_clamp_f32(rfc), 4, np.float32),
            ("biquad_f32_df2t", H.biquad_f32(rf), 2, np.float32), ("biquad_f32_df2t_clamp", H.biquad_clamp_f32(rfc), 2, np.float32)]


def test_chains_of_four_to_eight_sections(gpu):
    rng = np.random.default_rng(501)
    shapes = [(40960, 70, 40960, 0), (41000, 33, 41000, 0), (65536, 32, 65540, 4), (49152, 5, 49152, 0), (131072, 64, 131072, 0)]
    for n in (4, 5, 6, 7, 8):
        cs = chain_cases(rng, n)
        for i, (lanes, frames, pitch, off) in enumerate(shapes):
            for j, (op, cfg, words, dt) in enumerate(cs):
                if (i + j + n) % 4:
                    continue
                FMS.run_case(gpu, op, cfg, n, words, dt, rng, lanes, frames, pitch, bool((i + j) & 1), off=off)
                want_duo = n >= 5 or (lanes <= 98304 and "clamp" not in op and words <= 4)  # four sections: plain DF1 / DF2T forms only
                assert kernel_of(gpu).startswith("stream_frame_major_duo<") == want_duo, (op, n, lanes, kernel_of(gpu))


def test_twelve_sections_are_two_launches(gpu):
    rng = np.random.default_rng(502)
    op, cfg, words, dt = chain_cases(rng, 12)[0]
    FMS.run_case(gpu, op, cfg, 12, words, dt, rng, 45056, 40, 45056, False)
    assert kernel_of(gpu).startswith("stream_frame_major_duo<"), kernel_of(gpu)  # 8 sections, then 4 in place


@pytest.mark.parametrize("op,dt,mk", [("cascade_i32_df1", np.int32, "i32"), ("cascade_f32_df1", np.float32, "f32")])
def test_cascades_of_five_to_eight_sections(gpu, op, dt, mk):
    rng = np.random.default_rng(503)
    o = H.oracle()
    for n in (5, 6, 7, 8):
        if mk == "i32":
            cfg = H.biquad_i32([(rng.integers(-(1 << 28), 1 << 28, size=5).tolist(), 29) for _ in range(n)])
        else:
            cfg = H.biquad_f32([(rng.standard_normal(5) * 0.3).tolist() for _ in range(n)])
        words = (2 + 2 * n)
        for lanes, frames, pitch, off, inplace in ((40960, 67, 40960, 0, False), (65537, 32, 65600, 3, True)):
            xh = sample(rng, dt, lanes * frames).reshape(frames, lanes)
            want = np.empty_like(xh)
            st0 = init_state(rng, dt, words, lanes)
            so = st0.copy()
            assert o.stream(op, cfg, n, so, xh, want, lanes, frames, FM) == 0
            xb = torch.full((frames * pitch,), SENT, dtype=tdtype(dt), device=DEV)
            xb.view(frames, pitch)[:, off:off + lanes] = torch.from_numpy(xh).to(DEV)
            yb = xb if inplace else torch.full((frames * pitch,), SENT, dtype=tdtype(dt), device=DEV)
            sg = torch.from_numpy(st0.view(np.int32)).to(DEV)
            rc = gpu.fn[op + "_pitch"](C.cast(cfg, C.c_void_p), n, p(sg), C.c_void_p(xb.data_ptr() + off * 4), pitch, C.c_void_p(yb.data_ptr() + off * 4), pitch,
                                       lanes, frames, FM, None)
            torch.cuda.synchronize()
            assert rc == 0 and kernel_of(gpu).startswith("stream_frame_major_duo<"), (gpu.err(), kernel_of(gpu))
            yv = yb.view(frames, pitch)
            assert np.array_equal(yv[:, off:off + lanes].cpu().numpy().view(np.uint32), want.view(np.uint32)), (op, n, lanes)
            assert np.array_equal(sg.cpu().numpy().view(np.uint32), so), (op, n, "state")
            assert (yv[:, :off] == SENT).all() and (yv[:, off + lanes:] == SENT).all()


def test_bylane_banks_of_three_to_six_sections(gpu):
    """`ByLane` banks (dsp-process/src/compose.rs:363-390): three or four sections per pass on the two-wave kernel, the second
    wave's coefficient and state planes starting two sections further into the records."""
    from tests import _bylane_cases as B
    from tests._backends import GpuBackend, OracleBackend

    ob, gb = OracleBackend(), GpuBackend()
    rng = np.random.default_rng(504)
    for op, dtype, words, clamp in B.OPS:
        if np.dtype(dtype).itemsize != 4:
            continue
        for n, lanes, frames in ((3, 40960, 37), (4, 41001, 32), (6, 40960, 70)):
            if (n + len(op)) % 2:
                continue  # half of the (entry, shape) pairs
            frac = 29 if dtype == np.int32 else None
            coef = B.coef_planes(rng, dtype, n, lanes, clamp, frac or 0)
            x = B.samples(rng, dtype, lanes * frames)
            init = B.init_state(rng, dtype, words * n, lanes)
            so, sg = init.copy(), init.copy()
            rco, yo = ob.bylane(op, coef, frac, n, so, x.copy(), lanes, frames, B.FM)
            rcg, yg = gb.bylane(op, coef, frac, n, sg, x.copy(), lanes, frames, B.FM)
            assert rco == 0 and rcg == 0, H.engine().err()
            assert kernel_of(gpu).startswith("stream_frame_major_duo<") == (n != 6), (op, n, kernel_of(gpu))  # 6 = 4 (two waves) + 2
            assert np.array_equal(B.bits(yo), B.bits(yg)) and np.array_equal(so, sg), (op, n, lanes)


def test_wdf_chains_of_two_to_four_sections(gpu):
    """`Wdf` sections in series (src/iir/wdf.rs:178-214 per section; the chain is a slice composition): the first
    ceil(k/2) sections on wave 0, the rest on wave 1 with its state planes starting after wave 0's `sum n` words."""
    from tests import _nw_cases as W
    from tests._backends import GpuBackend, OracleBackend

    ob, gb = OracleBackend(), GpuBackend()
    rng = np.random.default_rng(505)
    bench = [W.wdf_section(ob, m, g)[1] for m, g in W.WDF_BENCH]
    for k, lanes, frames in ((2, 40960, 37), (3, 41001, 32), (4, 40960, 70), (1, 40960, 33), (7, 45056, 40)):
        low = W.random_wdf(rng, k)
        for sec in low:  # orders 1 .. 4: the two-wave form
            sec.n = 1 + sec.n % 4
            sec.m &= (1 << (4 * sec.n)) - 1
        for secs in (W.random_wdf(rng, k), low, bench[:k]):
            cfg = W.wdf_array(secs)
            words = gb.helper("wdf_state_words", C.cast(cfg, C.c_void_p), k)
            init = rng.integers(0, 1 << 32, size=(words, lanes), dtype=np.uint64).astype(np.uint32)
            x = rng.integers(W.I32_MIN, W.I32_MAX, size=lanes * frames, dtype=np.int64, endpoint=True).astype(np.int32)
            for inplace in (False, True):
                so, sg = init.copy(), init.copy()
                rco, yo = ob.stream("wdf_i32", cfg, k, so, x.copy(), lanes, frames, W.FM, inplace=inplace)
                rcg, yg = gb.stream("wdf_i32", cfg, k, sg, x.copy(), lanes, frames, W.FM, inplace=inplace)
                assert rco == 0 and rcg == 0, H.engine().err()
                last = secs[4 * ((k - 1) // 4):]  # the sections of the last pass; orders 5 .. 8 stay on one wave
                want_duo = len(last) >= 2 and max(s.n for s in last) <= 4
                assert kernel_of(gpu).startswith("stream_frame_major_duo<") == want_duo, (k, kernel_of(gpu))
                assert np.array_equal(yo, yg) and np.array_equal(so, sg), (k, lanes, inplace)
