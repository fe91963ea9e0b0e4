// lockin_stream_procs.h — the one- / two-thread-per-lane stream processors of the lock-in and of `[Lowpass<N>; K]` (the shapes the
// multi-wave kernels of lockin_waves.h do not take, and idsp_lowpass_i32), with their dispatch over N and K.  Split out of dds.hip in
// round 3 so that the 40 launch_stream instantiations compile in four translation units beside it instead of inside it
// (lockin_stream_iq.hip, lockin_stream_arg.hip, lockin_stream_norm_sqr.hip, lowpass.hip): dds.hip alone took 4.5 minutes.
#pragma once
#include "dds_dev.h"

namespace idsp {
namespace {

template <int N, int K>
struct LowpassProc {
    using In = int32_t;
    using Out = int32_t;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 0;
    static constexpr int IN_DIV = 1;
    static constexpr int COST = 40 * N * K;
    using Params = LpParams;
    LpBank<N, K> b;
    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t lanes, size_t lane) { b.load(st, lanes, lane, 0); }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane) { b.store(st, lanes, lane, 0); }
    __device__ __forceinline__ Out step(const Params &p, In x) { return b.step(p, x); }
};

// src/lockin.rs:30-39 -> :17-27.  Mixer `x * Q32<32>` =
// ((q as i64 * x as i64) >> 32) as i32 (dsp-fixedpoint/src/lib.rs:449-456).
template <int N, int K>
struct LockinProc {
    using In = int32_t;
    using Out = Cplx;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 1 << kCossinDepth;
    static constexpr int IN_DIV = 1;
    static constexpr int COST = 110 + 80 * N * K;
    using Params = LpParams;
    const uint32_t *lut;
    uint32_t acc, inc;
    LpBank<N, K> bi, bq;
    static __device__ __forceinline__ void fill_shared(uint32_t *sh, int tid, int n) { fill_cossin(sh, tid, n); }
    __device__ __forceinline__ void set_shared(const uint32_t *sh) { lut = sh; }
    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t lanes, size_t lane)
    {
        acc = st[lane];
        inc = st[lanes + lane];
        bi.load(st, lanes, lane, 2);
        bq.load(st, lanes, lane, 2 + 2 * N * K);
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane)
    {
        st[lane] = acc;
        bi.store(st, lanes, lane, 2);
        bq.store(st, lanes, lane, 2 + 2 * N * K);
    }
    static constexpr int BATCH = 4;
    using Pre = Cplx;
    __device__ __forceinline__ Pre pre(const Params &)
    {
        acc += inc;
        return cossin_dev(int32_t(acc), lut);
    }
    __device__ __forceinline__ Out step(const Params &p, In x, const Pre &lo)
    {
        const int32_t xi = __mulhi(lo.re, x);
        const int32_t xq = __mulhi(lo.im, x);
        return Cplx{bi.step(p, xi), bq.step(p, xq)};
    }
};

// I/Q arms on two adjacent threads ("virtual lanes" 2*lane + iq): used when the
// lane count alone cannot give every SIMD a wave (C4: 32768 lanes = 512 waves).
// Both threads step the same phase accumulator and evaluate cossin; each runs
// one arm of the mixer + lowpass cascade and writes one word of Complex<i32>,
// so a wave still stores 256 contiguous bytes per frame.
template <int N, int K>
struct LockinSplitProc {
    using In = int32_t;
    using Out = int32_t;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 1 << kCossinDepth;
    static constexpr int IN_DIV = 2;
    static constexpr int COST = 70 + 40 * N * K;
    using Params = LpParams;
    const uint32_t *lut;
    uint32_t acc, inc;
    bool q;
    LpBank<N, K> b;
    static __device__ __forceinline__ void fill_shared(uint32_t *sh, int tid, int n) { fill_cossin(sh, tid, n); }
    __device__ __forceinline__ void set_shared(const uint32_t *sh) { lut = sh; }
    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t vlanes, size_t vlane)
    {
        const size_t lanes = vlanes / 2, lane = vlane / 2;
        q = vlane & 1;
        acc = st[lane];
        inc = st[lanes + lane];
        b.load(st, lanes, lane, 2 + (q ? 2 * N * K : 0));
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t vlanes, size_t vlane)
    {
        const size_t lanes = vlanes / 2, lane = vlane / 2;
        if (!q) st[lane] = acc;
        b.store(st, lanes, lane, 2 + (q ? 2 * N * K : 0));
    }
    static constexpr int BATCH = 4;
    using Pre = int32_t;  // this arm's LO component
    __device__ __forceinline__ Pre pre(const Params &)
    {
        acc += inc;
        const Cplx lo = cossin_dev(int32_t(acc), lut);
        return q ? lo.im : lo.re;
    }
    // The two threads of a lane evaluate the LO of alternate frames (thread q: frames 2j + q) and
    // swap the component the partner needs with one DPP move: 16 instead of 32 cossin
    // instructions per frame and thread on a VALU-bound kernel.
    __device__ __forceinline__ void pre_batch(const Params &, Pre (&out)[BATCH])
    {
#pragma unroll
        for (int j = 0; j < BATCH / 2; j++) {
            const uint32_t ph = acc + inc * uint32_t(2 * j + 1) + (q ? inc : 0u);
            const Cplx lo = cossin_dev(int32_t(ph), lut);
            const int32_t mine = q ? lo.im : lo.re, other = q ? lo.re : lo.im;
            const int32_t recv = pair_swap(other);
            out[2 * j] = q ? recv : mine;
            out[2 * j + 1] = q ? mine : recv;
        }
        acc += inc * uint32_t(BATCH);
    }
    __device__ __forceinline__ Out step(const Params &p, In x, const Pre &lo) { return b.step(p, __mulhi(lo, x)); }
};

// Lock-in with the polar read-out fused on the same thread: `Lockin::process(..)` (src/lockin.rs:30-39)
// followed by `Complex::<i32>::arg()` (src/complex.rs:254-256, MODE 0, i32) or `norm_sqr()`
// (src/complex.rs:214-217, MODE 1, i64 with the wrapping sum of a release build).  Saves the 8 byte/sample
// Complex<i32> round trip through HBM of `lockin` + `atan2`.
template <int N, int K, int MODE>
struct LockinPolarProc {
    using In = int32_t;
    using Out = std::conditional_t<MODE == 0, int32_t, int64_t>;
    static constexpr bool HAS_IN = true;
    static constexpr int kLut = 1 << kCossinDepth;
    static constexpr int LDS_WORDS = kLut + (MODE == 0 ? 32 : 0);  // cossin table, atan2 reciprocal table
    static constexpr int IN_DIV = 1;
    static constexpr bool LM_ONE_FORM = true;  // stream fall-back of the multi-wave kernel: one LaneMajor form is enough
    // three or four cascaded second-order arms + atan2 next to the staged kernel's 128 staging registers: 12 / 116 B of scratch per
    // thread (tools/check_scratch.py) — those stay on the tile kernel
    static constexpr bool LM_STAGED = !(MODE == 0 && N * K >= 6);
    static constexpr int COST = 110 + 80 * N * K + (MODE == 0 ? 80 : 10);
    using Params = LpParams;
    const uint32_t *lut;
    uint32_t acc, inc;
    LpBank<N, K> bi, bq;
    static __device__ __forceinline__ void fill_shared(uint32_t *sh, int tid, int n)
    {
        fill_cossin(sh, tid, n);
        if constexpr (MODE == 0)
            for (int i = tid; i < 32; i += n) sh[kLut + i] = d_atan2_table[i];
    }
    __device__ __forceinline__ void set_shared(const uint32_t *sh) { lut = sh; }
    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t lanes, size_t lane)
    {
        acc = st[lane];
        inc = st[lanes + lane];
        bi.load(st, lanes, lane, 2);
        bq.load(st, lanes, lane, 2 + 2 * N * K);
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane)
    {
        st[lane] = acc;
        bi.store(st, lanes, lane, 2);
        bq.store(st, lanes, lane, 2 + 2 * N * K);
    }
    static constexpr int BATCH = 4;
    using Pre = Cplx;
    __device__ __forceinline__ Pre pre(const Params &)
    {
        acc += inc;
        return cossin_dev(int32_t(acc), lut);
    }
    __device__ __forceinline__ Out step(const Params &p, In x, const Pre &lo)
    {
        const int32_t re = bi.step(p, __mulhi(lo.re, x));
        const int32_t im = bq.step(p, __mulhi(lo.im, x));
        if constexpr (MODE == 0)
            return atan2_dev(im, re, lut + kLut);
        else
            return int64_t(uint64_t(int64_t(re) * re) + uint64_t(int64_t(im) * im));
    }
};
template <int N, int K>
using LockinArgProc = LockinPolarProc<N, K, 0>;
template <int N, int K>
using LockinNormSqrProc = LockinPolarProc<N, K, 1>;

template <template <int, int> class Proc, class OutT>
int dispatch_nk(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, OutT *y, size_t lanes, size_t frames,
                int layout, hipStream_t s, size_t pitch = 0)
{
    // pitch (LaneMajor): elements between the rows of x and of y, 0 = dense
    const LpParams p = lp_params(cfg);
#define IDSP_CASE(N, K) \
    if (cfg->order == N && cfg->cascade == K) return launch_stream<Proc<N, K>>(p, state, x, y, lanes, frames, layout, s, Pitch{pitch, pitch})
    IDSP_CASE(1, 1);
    IDSP_CASE(1, 2);
    IDSP_CASE(1, 3);
    IDSP_CASE(1, 4);
    IDSP_CASE(2, 1);
    IDSP_CASE(2, 2);
    IDSP_CASE(2, 3);
    IDSP_CASE(2, 4);
#undef IDSP_CASE
    return fail(IDSP_EINVAL, "unsupported lowpass configuration");
}

}  // namespace
}  // namespace idsp
