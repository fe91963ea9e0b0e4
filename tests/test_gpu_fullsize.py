"""BASELINE.json configurations at FULL size on the GPU (C2, C3, C4, C5 shapes):
parity against the CPU oracle on a lane subset that the oracle finishes in
seconds (all lanes for C2), plus size-independent properties over the whole
tensor: chunked calls == one call (streaming continuity), LANE_MAJOR ==
transposed FRAME_MAJOR, and for the decimator a DC-gain check."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from idsp_amd import _abi
from tests import _harness as H

pytestmark = pytest.mark.gpu
FM, LM = H.FM, H.LM
DEV = "cuda:0"


@pytest.fixture(scope="module")
def eng(gpu):
    return gpu


def p(t):
    return C.c_void_p(t.data_ptr())


def lowpass_i32(frac=30, f0=0.01):
    o = H.oracle()
    q = _abi.BiquadI32()
    assert o.fn["biquad_i32_from_sos"]((C.c_double * 6)(*o.lowpass_sos(f0)), frac, C.byref(q)) == 0
    return (_abi.BiquadI32 * 1)(q)


def test_c2_i32_df1_65536_lanes_full_parity(eng):
    lanes, frames = 65536, 4096
    o = H.oracle()
    cfg = lowpass_i32()
    g = torch.Generator(device=DEV)
    g.manual_seed(2)
    x = torch.randint(-(1 << 24), 1 << 24, (frames, lanes), dtype=torch.int32, device=DEV, generator=g)
    y = torch.empty_like(x)
    st = torch.zeros((4, lanes), dtype=torch.int32, device=DEV)
    assert eng.stream("biquad_i32_df1", cfg, 1, st, x, y, lanes, frames, FM) == 0
    torch.cuda.synchronize()
    # oracle over ALL lanes (threaded over lane blocks), bit-exact incl. final state
    xh = x.cpu().numpy()
    yh = np.empty_like(xh)
    sh = np.zeros((4, lanes), np.uint32)
    rc = o.lib.idsp_ref_biquad_i32_df1_mt(C.cast(cfg, C.c_void_p), 1, sh.ctypes.data, xh.ctypes.data, yh.ctypes.data,
                                          lanes, frames, FM, 64)
    assert rc == 0
    assert np.array_equal(y.cpu().numpy(), yh)
    assert np.array_equal(st.cpu().numpy().view(np.uint32), sh)
    # chunked == whole (three ragged pieces), on the GPU at full width
    y2 = torch.empty_like(x)
    st2 = torch.zeros_like(st)
    for a, b in ((0, 1000), (1000, 1001), (1001, 4096)):
        assert eng.stream("biquad_i32_df1", cfg, 1, st2, x[a:b], y2[a:b], lanes, b - a, FM) == 0
    assert torch.equal(y, y2) and torch.equal(st, st2)
    # LANE_MAJOR on the transposed tensor gives the transposed result; in place
    xt = x.t().contiguous()
    st3 = torch.zeros_like(st)
    assert eng.stream("biquad_i32_df1", cfg, 1, st3, xt, xt, lanes, frames, LM) == 0
    assert torch.equal(xt.t(), y) and torch.equal(st, st3)


def test_c3_hbf_dec16_16384_lanes(eng):
    lanes, frames, R = 16384, 4096, 16
    o = H.oracle()
    cfg = _abi.HbfCascadeF32()
    assert o.fn["hbf_dec_cascade"](0, 4, C.byref(cfg)) == 0
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    x = torch.randn((lanes, frames * R), dtype=torch.float32, device=DEV, generator=g)  # LANE_MAJOR streams
    y = torch.empty((lanes, frames), dtype=torch.float32, device=DEV)
    st = torch.zeros((118, lanes), dtype=torch.int32, device=DEV)
    assert eng.cfgcall("hbf_dec_f32", cfg, st, x, y, lanes, frames, LM) == 0
    torch.cuda.synchronize()
    idx = np.unique(np.concatenate([np.arange(0, lanes, 127), [0, 1, lanes - 1]]))
    xs = np.ascontiguousarray(x.cpu().numpy()[idx])
    ys = np.empty((idx.size, frames), np.float32)
    ss = np.zeros((118, idx.size), np.uint32)
    assert o.cfgcall("hbf_dec_f32", cfg, ss, xs, ys, idx.size, frames, LM) == 0
    assert H.ulp_diff_f32(y.cpu().numpy()[idx], ys).max() == 0
    assert np.array_equal(st.cpu().numpy().view(np.uint32)[:, idx], ss)
    # chunked == whole over all lanes (ragged split that is not a multiple of the kernel chunk)
    y2 = torch.empty_like(y)
    st2 = torch.zeros_like(st)
    cut = 1500
    xa = x.view(lanes, frames, R)
    # LANE_MAJOR chunks of each lane are not contiguous inside x: copy the two time segments out
    for a, b in ((0, cut), (cut, frames)):
        xc = xa[:, a:b].contiguous()
        yc = torch.empty((lanes, b - a), dtype=torch.float32, device=DEV)
        assert eng.cfgcall("hbf_dec_f32", cfg, st2, xc, yc, lanes, b - a, LM) == 0
        y2[:, a:b] = yc
    assert torch.equal(y.view(torch.int32), y2.view(torch.int32)) and torch.equal(st, st2)
    # FRAME_MAJOR ([f][lane][R]) on the permuted tensor gives the same samples
    xf = xa.permute(1, 0, 2).contiguous()
    yf = torch.empty((frames, lanes), dtype=torch.float32, device=DEV)
    st3 = torch.zeros_like(st)
    assert eng.cfgcall("hbf_dec_f32", cfg, st3, xf, yf, lanes, frames, FM) == 0
    assert torch.equal(yf.t().contiguous().view(torch.int32), y.view(torch.int32))
    # property: DC gain 2 per stage (src/hbf.rs:548-555 KAT scaled up): constant input -> 16x
    x.fill_(1.0)
    st.zero_()
    assert eng.cfgcall("hbf_dec_f32", cfg, st, x, y, lanes, frames, LM) == 0
    assert torch.allclose(y[:, 100:], torch.full_like(y[:, 100:], 16.0), atol=1e-4)
    # interpolator x16 at full width: oracle on a lane subset, FRAME_MAJOR (4-lane workgroups) == LANE_MAJOR
    icfg = _abi.HbfCascadeF32()
    assert o.fn["hbf_int_cascade"](0, 4, C.byref(icfg)) == 0
    fi = 1024
    words = o.fn["hbf_int_state_words"](C.byref(icfg))
    xi = torch.randn((lanes, fi), dtype=torch.float32, device=DEV, generator=g)
    yi = torch.empty((lanes, fi * R), dtype=torch.float32, device=DEV)
    sti = torch.zeros((words, lanes), dtype=torch.int32, device=DEV)
    assert eng.cfgcall("hbf_int_f32", icfg, sti, xi, yi, lanes, fi, LM) == 0
    torch.cuda.synchronize()
    xs = np.ascontiguousarray(xi.cpu().numpy()[idx])
    ys = np.empty((idx.size, fi * R), np.float32)
    ss = np.zeros((words, idx.size), np.uint32)
    assert o.cfgcall("hbf_int_f32", icfg, ss, xs, ys, idx.size, fi, LM) == 0
    assert H.ulp_diff_f32(yi.cpu().numpy()[idx], ys).max() == 0
    assert np.array_equal(sti.cpu().numpy().view(np.uint32)[:, idx], ss)
    yif = torch.empty((fi, lanes, R), dtype=torch.float32, device=DEV)
    sti2 = torch.zeros_like(sti)
    assert eng.cfgcall("hbf_int_f32", icfg, sti2, xi.t().contiguous(), yif, lanes, fi, FM) == 0
    assert torch.equal(yif.permute(1, 0, 2).reshape(lanes, fi * R).view(torch.int32), yi.view(torch.int32)) and torch.equal(sti, sti2)


def test_c4_lockin_32768_lanes(eng):
    lanes, frames = 32768, 4096
    o = H.oracle()
    k = math.pi * (1 << 31) * 1e-3
    lp = [int(k * k / (1 << 32)), -int(k * math.sqrt(2.0))]
    cfg = H.lockin_cfg([lp, lp])
    words = 18
    g = torch.Generator(device=DEV)
    g.manual_seed(4)
    x = torch.randint(-(1 << 28), 1 << 28, (frames, lanes), dtype=torch.int32, device=DEV, generator=g)
    st0 = torch.zeros((words, lanes), dtype=torch.int32, device=DEV)
    st0[1] = torch.randint(-(1 << 31), (1 << 31) - 1, (lanes,), dtype=torch.int64, device=DEV, generator=g).to(torch.int32)
    st = st0.clone()
    y = torch.empty((frames, lanes, 2), dtype=torch.int32, device=DEV)
    assert eng.cfgcall("lockin_i32_process", cfg, st, x, y, lanes, frames, FM) == 0
    torch.cuda.synchronize()
    idx = np.unique(np.concatenate([np.arange(0, lanes, 61), [lanes - 1]]))
    xs = np.ascontiguousarray(x.cpu().numpy()[:, idx])
    ss = np.ascontiguousarray(st0.cpu().numpy().view(np.uint32)[:, idx])
    ys = np.empty((frames, idx.size, 2), np.int32)
    assert o.cfgcall("lockin_i32_process", cfg, ss, xs, ys, idx.size, frames, FM) == 0
    assert np.array_equal(y.cpu().numpy()[:, idx], ys)
    assert np.array_equal(st.cpu().numpy().view(np.uint32)[:, idx], ss)
    # chunked == whole, all lanes
    st2 = st0.clone()
    y2 = torch.empty_like(y)
    for a, b in ((0, 7), (7, 2048), (2048, 4096)):
        assert eng.cfgcall("lockin_i32_process", cfg, st2, x[a:b], y2[a:b], lanes, b - a, FM) == 0
    assert torch.equal(y, y2) and torch.equal(st, st2)
    # LANE_MAJOR equals transposed FRAME_MAJOR
    st3 = st0.clone()
    yl = torch.empty((lanes, frames, 2), dtype=torch.int32, device=DEV)
    assert eng.cfgcall("lockin_i32_process", cfg, st3, x.t().contiguous(), yl, lanes, frames, LM) == 0
    assert torch.equal(yl.permute(1, 0, 2), y) and torch.equal(st, st3)
    # polar read-outs fused into the pass == Complex::arg / norm_sqr of the Complex<i32> output, both layouts
    want_arg = torch.empty((frames, lanes), dtype=torch.int32, device=DEV)
    assert eng.fn["atan2_i32"](C.c_void_p(y.data_ptr()), C.c_void_p(want_arg.data_ptr()), frames * lanes, None) == 0
    z = y.to(torch.int64)
    want_pow = z[..., 0] * z[..., 0] + z[..., 1] * z[..., 1]
    del z
    for layout in (FM, LM):
        xin = x if layout == FM else x.t().contiguous()
        shape = (frames, lanes) if layout == FM else (lanes, frames)
        for name, dt, want in (("lockin_i32_arg", torch.int32, want_arg), ("lockin_i32_norm_sqr", torch.int64, want_pow)):
            st4 = st0.clone()
            out = torch.empty(shape, dtype=dt, device=DEV)
            assert eng.cfgcall(name, cfg, st4, xin, out, lanes, frames, layout) == 0
            got = out if layout == FM else out.t()
            assert torch.equal(got, want) and torch.equal(st, st4), (name, layout)


def test_c5_f32_df2t_one_million_lanes(eng):
    """The whole C5 tensor (2^20 lanes x 4096, 16 GiB in + 16 GiB out) on ONE GPU, and the
    8-way lane split of it: shard results concatenate to the unsharded result."""
    from idsp_amd.sharding import lane_shard

    lanes, frames = 1 << 20, 4096
    o = H.oracle()
    q = _abi.BiquadF32()
    assert o.fn["biquad_f32_from_sos_f64"]((C.c_double * 6)(*o.lowpass_sos(0.01)), C.byref(q)) == 0
    cfg = (_abi.BiquadF32 * 1)(q)
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    x = torch.randn((frames, lanes), dtype=torch.float32, device=DEV, generator=g)
    y = torch.empty_like(x)
    st = torch.zeros((2, lanes), dtype=torch.int32, device=DEV)
    assert eng.stream("biquad_f32_df2t", cfg, 1, st, x, y, lanes, frames, FM) == 0
    torch.cuda.synchronize()
    idx = np.unique(np.concatenate([np.arange(0, lanes, 2039), [1, lanes - 1]]))
    tidx = torch.from_numpy(idx).to(DEV)
    xs = np.ascontiguousarray(x[:, tidx].cpu().numpy())
    ys = np.empty_like(xs)
    ss = np.zeros((2, idx.size), np.uint32)
    assert o.stream("biquad_f32_df2t", cfg, 1, ss, xs, ys, idx.size, frames, FM) == 0
    assert H.ulp_diff_f32(y[:, tidx].cpu().numpy(), ys).max() == 0  # bar: 0 ULP (allowed: 1)
    assert np.array_equal(st[:, tidx].cpu().numpy().view(np.uint32), ss)
    # 8-way lane split (what each rank of an 8-GPU run computes), LANE_MAJOR shards of 512 frames
    fr = 512
    xt = x[:fr].t().contiguous()  # [lanes, fr]
    whole = torch.empty_like(xt)
    stw = torch.zeros((2, lanes), dtype=torch.int32, device=DEV)
    assert eng.stream("biquad_f32_df2t", cfg, 1, stw, xt, whole, lanes, fr, LM) == 0
    for r in range(8):
        lo, hi = lane_shard(lanes, r, 8)
        ysh = torch.empty((hi - lo, fr), dtype=torch.float32, device=DEV)
        sts = torch.zeros((2, hi - lo), dtype=torch.int32, device=DEV)
        assert eng.stream("biquad_f32_df2t", cfg, 1, sts, xt[lo:hi], ysh, hi - lo, fr, LM) == 0
        assert torch.equal(ysh.view(torch.int32), whole[lo:hi].view(torch.int32))
        assert torch.equal(sts, stw[:, lo:hi])
    assert torch.equal(whole.t().contiguous().view(torch.int32), y[:fr].view(torch.int32))


def test_bylane_bank_65536_lanes(eng):
    """`ByLane` at the C2 shape (65536 different Q30 lowpasses x 4096 samples): replicated coefficients
    reproduce the shared-coefficient launch bit for bit over the whole tensor; a bank of distinct filters
    is checked against the oracle on a lane subset; LANE_MAJOR == transposed FRAME_MAJOR."""
    lanes, frames = 65536, 4096
    o = H.oracle()
    cfg = lowpass_i32()
    g = torch.Generator(device=DEV)
    g.manual_seed(12)
    x = torch.randint(-(1 << 24), 1 << 24, (frames, lanes), dtype=torch.int32, device=DEV, generator=g)
    base = torch.tensor(list(cfg[0].ba), dtype=torch.int32, device=DEV)
    coef = base.view(1, 5, 1).expand(1, 5, lanes).contiguous()
    y_sh, y_bl = torch.empty_like(x), torch.empty_like(x)
    st_sh = torch.zeros((4, lanes), dtype=torch.int32, device=DEV)
    st_bl = torch.zeros_like(st_sh)
    assert eng.stream("biquad_i32_df1", cfg, 1, st_sh, x, y_sh, lanes, frames, FM) == 0
    assert eng.fn["biquad_i32_df1_bylane"](p(coef), 30, 1, p(st_bl), p(x), p(y_bl), lanes, frames, FM, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(y_sh, y_bl) and torch.equal(st_sh, st_bl)
    # distinct filters per lane: f0 spread over 3 decades, quantised on the host like `Biquad::from`
    f0 = torch.logspace(-4, -1, lanes, dtype=torch.float64)
    w0 = 2 * math.pi * f0
    alpha = 0.5 * torch.sin(w0) * math.sqrt(2.0)
    b = 0.5 * (1 - torch.cos(w0))
    a0 = 1 + alpha
    ba = torch.stack([b / a0, 2 * b / a0, b / a0, 2 * torch.cos(w0) / a0, -(1 - alpha) / a0], 0)
    q = torch.clamp(torch.round(ba * float(1 << 30)), -(1 << 31), (1 << 31) - 1).to(torch.int32)
    coef = q.view(1, 5, lanes).contiguous().to(DEV)
    st = torch.zeros((4, lanes), dtype=torch.int32, device=DEV)
    y = torch.empty_like(x)
    assert eng.fn["biquad_i32_df1_bylane"](p(coef), 30, 1, p(st), p(x), p(y), lanes, frames, FM, None) == 0
    torch.cuda.synchronize()
    sub = np.r_[0:64, 30000:30064, 65472:65536]
    xs = np.ascontiguousarray(x[:, sub].cpu().numpy())
    cs = np.ascontiguousarray(coef[:, :, sub].cpu().numpy())
    ys = np.empty_like(xs)
    ss = np.zeros((4, sub.size), np.uint32)
    assert o.fn["biquad_i32_df1_bylane"](H._ptr(cs), 30, 1, H._ptr(ss), H._ptr(xs), H._ptr(ys), sub.size, frames, FM) == 0
    assert np.array_equal(y[:, sub].cpu().numpy(), ys)
    assert np.array_equal(st[:, sub].cpu().numpy().view(np.uint32), ss)
    # LANE_MAJOR == transposed FRAME_MAJOR
    xt = x.t().contiguous()
    yt = torch.empty_like(xt)
    st2 = torch.zeros_like(st)
    assert eng.fn["biquad_i32_df1_bylane"](p(coef), 30, 1, p(st2), p(xt), p(yt), lanes, frames, LM, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(yt.t(), y) and torch.equal(st, st2)


def test_cic_decimator_into_hbf_16384_lanes(eng):
    """`Cic` /16 in front of the half-band chain at 16384 lanes: parity on a lane subset, chunked ==
    whole, LANE_MAJOR == FRAME_MAJOR, and interpolate -> decimate returns gain^2 * x once settled."""
    lanes, frames, R = 16384, 1024, 16
    o = H.oracle()
    cfg = _abi.Cic(3, 1, R - 1)
    g = torch.Generator(device=DEV)
    g.manual_seed(13)
    x = torch.randint(-(1 << 15), 1 << 15, (frames, lanes, R), dtype=torch.int32, device=DEV, generator=g)
    words = eng.fn["cic_state_words"](C.byref(cfg), 32)
    st = torch.zeros((words, lanes), dtype=torch.int32, device=DEV)
    y = torch.empty((frames, lanes), dtype=torch.int32, device=DEV)
    assert eng.cfgcall("cic_dec_i32", cfg, st, x, y, lanes, frames, FM) == 0
    torch.cuda.synchronize()
    sub = np.r_[0:32, 9000:9032, 16352:16384]
    xs = np.ascontiguousarray(x[:, sub].cpu().numpy())
    ys = np.empty((frames, sub.size), np.int32)
    ss = np.zeros((words, sub.size), np.uint32)
    assert o.cfgcall("cic_dec_i32", cfg, ss, xs, ys, sub.size, frames, FM) == 0
    assert np.array_equal(y[:, sub].cpu().numpy(), ys)
    assert np.array_equal(st[:, sub].cpu().numpy().view(np.uint32), ss)
    # chunked == whole
    st2 = torch.zeros_like(st)
    y2 = torch.empty_like(y)
    for a, b in ((0, 300), (300, 301), (301, frames)):
        assert eng.cfgcall("cic_dec_i32", cfg, st2, x[a:b], y2[a:b], lanes, b - a, FM) == 0
    torch.cuda.synchronize()
    assert torch.equal(y, y2) and torch.equal(st, st2)
    # LANE_MAJOR
    xl = x.permute(1, 0, 2).contiguous()
    yl = torch.empty((lanes, frames), dtype=torch.int32, device=DEV)
    st3 = torch.zeros_like(st)
    assert eng.cfgcall("cic_dec_i32", cfg, st3, xl, yl, lanes, frames, LM) == 0
    torch.cuda.synchronize()
    assert torch.equal(yl.t(), y) and torch.equal(st, st3)
    # x -> interpolate x16 -> decimate /16: DC gain = gain()^2 = R^6 once both impulse responses have passed
    dc = torch.randint(-100, 100, (1, lanes), dtype=torch.int64, device=DEV).expand(64, lanes).contiguous()
    sti = torch.zeros((eng.fn["cic_state_words"](C.byref(cfg), 64), lanes), dtype=torch.int32, device=DEV)
    std = torch.zeros_like(sti)
    up = torch.empty((64, lanes, R), dtype=torch.int64, device=DEV)
    down = torch.empty((64, lanes), dtype=torch.int64, device=DEV)
    assert eng.cfgcall("cic_int_i64", cfg, sti, dc, up, lanes, 64, FM) == 0
    assert eng.cfgcall("cic_dec_i64", cfg, std, up, down, lanes, 64, FM) == 0
    torch.cuda.synchronize()
    assert torch.equal(down[-1], dc[0] * (R ** 6))


def _subset(lanes, step):
    return np.unique(np.concatenate([np.arange(0, lanes, step), [0, 1, 255, 256, lanes - 257, lanes - 1]]))


C2_VARIANTS = [
    # op, state words, dtype, clamp record?  (the processors profiles/NOTES.md section 5 quotes C2-shape throughput for)
    ("biquad_i32_df1_clamp", 4, np.int32, True),
    ("biquad_i32_dither", 5, np.int32, False),
    ("biquad_i32_dither_clamp", 5, np.int32, True),
    ("biquad_i32_wide", 6, np.int32, False),
    ("biquad_i32_wide_clamp", 6, np.int32, True),
    ("biquad_f32_df1", 4, np.float32, False),
    ("biquad_f32_df1_clamp", 4, np.float32, True),
    ("biquad_f32_df2t_clamp", 2, np.float32, True),
]


@pytest.mark.parametrize("op,words,dtype,clamp", C2_VARIANTS, ids=[v[0] for v in C2_VARIANTS])
def test_c2_shape_variants_on_default_dispatch(eng, op, words, dtype, clamp):
    """65536 lanes x 4096 frames, FRAME_MAJOR, default dispatch (= the LDS-DMA kernel, asserted): every clamp / dither /
    wide / f32 variant against the oracle on a lane subset, outputs and written-back state; clamp limits chosen so
    that they bite (the lowpass output of +-2^24 inputs exceeds +-2^22)."""
    lanes, frames = 65536, 4096
    o = H.oracle()
    g = torch.Generator(device=DEV)
    g.manual_seed(21)
    if dtype == np.int32:
        q = lowpass_i32()[0]
        cfg = H.biquad_clamp_i32([(list(q.ba), 30, 1234, -(1 << 22), (1 << 22) + 5)]) if clamp else H.biquad_i32([(list(q.ba), 30)])
        x = torch.randint(-(1 << 24), 1 << 24, (frames, lanes), dtype=torch.int32, device=DEV, generator=g)
    else:
        q = _abi.BiquadF32()
        assert o.fn["biquad_f32_from_sos_f64"]((C.c_double * 6)(*o.lowpass_sos(0.01)), C.byref(q)) == 0
        cfg = H.biquad_clamp_f32([(list(q.ba), 0.01, -0.05, 0.04)]) if clamp else H.biquad_f32([list(q.ba)])
        x = torch.randn((frames, lanes), dtype=torch.float32, device=DEV, generator=g)
    y = torch.empty_like(x)
    st = torch.zeros((words, lanes), dtype=torch.int32, device=DEV)
    assert eng.stream(op, cfg, 1, st, x, y, lanes, frames, FM) == 0
    torch.cuda.synchronize()
    assert eng.fn["last_kernel"]().decode().startswith("stream_frame_major_sweep[1 block/workgroup]<"), eng.fn["last_kernel"]()
    idx = _subset(lanes, 509)
    tidx = torch.from_numpy(idx).to(DEV)
    xs = np.ascontiguousarray(x[:, tidx].cpu().numpy())
    ys = np.empty_like(xs)
    ss = np.zeros((words, idx.size), np.uint32)
    assert o.stream(op, cfg, 1, ss, xs, ys, idx.size, frames, FM) == 0
    got = y[:, tidx].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ys.view(np.uint32))
    assert np.array_equal(st[:, tidx].cpu().numpy().view(np.uint32), ss)
    if clamp and dtype == np.int32:
        assert (got == (1 << 22) + 5).any() and (got == -(1 << 22)).any(), "the clamp must be active in this test"
    # in place at full width gives the same bits
    st2 = torch.zeros_like(st)
    assert eng.stream(op, cfg, 1, st2, x, x, lanes, frames, FM) == 0
    assert torch.equal(x.view(torch.int32), y.view(torch.int32)) and torch.equal(st, st2)


@pytest.mark.parametrize("lanes,want", [(131072, "stream_frame_major_sweep[2 blocks/workgroup]<"), (262144, "stream_frame_major_sweep[4 blocks/workgroup]<"),
                                        (327680, "stream_frame_major_sweep[8 blocks/workgroup]<")])
def test_large_lane_counts_on_default_dispatch(eng, lanes, want):
    """Beyond 65536 lanes the dense-sweep kernel (fm_sweep.h, round 5) gives every workgroup 2 / 4 / 8 interleaved lane blocks
    (C5 shards at 8 / 4 GPUs; 327680 lanes = 256 workgroups x 8 blocks of 160 lanes): i32 DF1 and f32
    DF2T against the oracle on a lane subset, ragged frame count, chunked == whole."""
    frames = 1003
    o = H.oracle()
    g = torch.Generator(device=DEV)
    g.manual_seed(22)
    q = _abi.BiquadF32()
    assert o.fn["biquad_f32_from_sos_f64"]((C.c_double * 6)(*o.lowpass_sos(0.01)), C.byref(q)) == 0
    cases = [("biquad_i32_df1", lowpass_i32(), 4, torch.randint(-(1 << 24), 1 << 24, (frames, lanes), dtype=torch.int32, device=DEV, generator=g)),
             ("biquad_f32_df2t", (_abi.BiquadF32 * 1)(q), 2, torch.randn((frames, lanes), dtype=torch.float32, device=DEV, generator=g))]
    for op, cfg, words, x in cases:
        y = torch.empty_like(x)
        st = torch.zeros((words, lanes), dtype=torch.int32, device=DEV)
        assert eng.stream(op, cfg, 1, st, x, y, lanes, frames, FM) == 0
        torch.cuda.synchronize()
        assert eng.fn["last_kernel"]().decode().startswith(want), eng.fn["last_kernel"]()
        idx = _subset(lanes, 1021)
        tidx = torch.from_numpy(idx).to(DEV)
        xs = np.ascontiguousarray(x[:, tidx].cpu().numpy())
        ys = np.empty_like(xs)
        ss = np.zeros((words, idx.size), np.uint32)
        assert o.stream(op, cfg, 1, ss, xs, ys, idx.size, frames, FM) == 0
        assert np.array_equal(y[:, tidx].cpu().numpy().view(np.uint32), ys.view(np.uint32)), op
        assert np.array_equal(st[:, tidx].cpu().numpy().view(np.uint32), ss), op
        y2 = torch.empty_like(x)
        st2 = torch.zeros_like(st)
        for a, b in ((0, 1), (1, 502), (502, frames)):
            assert eng.stream(op, cfg, 1, st2, x[a:b], y2[a:b], lanes, b - a, FM) == 0
        assert torch.equal(y.view(torch.int32), y2.view(torch.int32)) and torch.equal(st, st2), op


def test_c1_single_lane_one_million_samples(eng):
    """BASELINE.json configs[0] ("C1"): ONE lane, 2^20 samples, i32 DF1, seed 1 — the reference's plain
    `SplitProcess::block` / `SplitInplace::inplace` plumbing (dsp-process/src/process.rs:122-127,137-141) with no lane
    parallelism at all.  HIP == oracle bit for bit, out of place, in place and in ragged chunks; both layouts (they
    coincide for one lane)."""
    frames = 1 << 20
    o = H.oracle()
    cfg = lowpass_i32()
    x = np.random.default_rng(1).integers(-(1 << 24), 1 << 24, frames, dtype=np.int32)
    want = np.empty_like(x)
    sw = np.zeros((4, 1), np.uint32)
    assert o.stream("biquad_i32_df1", cfg, 1, sw, x, want, 1, frames, FM) == 0
    xd = torch.from_numpy(x).to(DEV)
    for layout in (FM, LM):
        y = torch.empty_like(xd)
        st = torch.zeros((4, 1), dtype=torch.int32, device=DEV)
        assert eng.stream("biquad_i32_df1", cfg, 1, st, xd, y, 1, frames, layout) == 0
        assert np.array_equal(y.cpu().numpy(), want) and np.array_equal(st.cpu().numpy().view(np.uint32), sw)
    xi = xd.clone()
    st = torch.zeros((4, 1), dtype=torch.int32, device=DEV)
    assert eng.stream("biquad_i32_df1", cfg, 1, st, xi, xi, 1, frames, FM) == 0
    assert np.array_equal(xi.cpu().numpy(), want)
    y = torch.empty_like(xd)
    st = torch.zeros((4, 1), dtype=torch.int32, device=DEV)
    cuts = [0, 1, 63, 64, 4097, 500000, frames]
    for a, b in zip(cuts, cuts[1:]):
        assert eng.stream("biquad_i32_df1", cfg, 1, st, xd[a:b], y[a:b], 1, b - a, LM) == 0
    assert np.array_equal(y.cpu().numpy(), want) and np.array_equal(st.cpu().numpy().view(np.uint32), sw)
