// biquad_sections.h — iir::Biquad / BiquadClamp / Cascade over many lanes
// (reference: src/iir/biquad.rs; fixed-point semantics dsp-fixedpoint/src).
//
// Each section type is a small device functor operating on the per-lane state
// words of include/idsp_hip.h held in registers; `Chain<Sec, N>` folds N
// serial sections sample-major, which yields the same values as the
// reference's stage-major slice composition (dsp-process/src/compose.rs:43-77)
// because every section is a causal function of its own input sequence.
// The extern "C" entry points live in biquad_*.hip (one translation unit per
// family so the many template instances compile in parallel).
//
// Integer paths are bit-exact restatements of Rust release (wrapping)
// arithmetic: i32 x i32 -> i64 products (v_mad_i64_i32), wrapping i64 sums,
// arithmetic `>> F`, truncating casts.  Float paths keep the reference's
// left-to-right association with every product and sum rounded separately
// (the library is compiled with -ffp-contract=off; f32 denormals enabled).
#pragma once

#include <type_traits>

#include "lane_stream.h"

namespace idsp {
namespace bq {

constexpr int kMaxChain = 4;    // sections fused per launch; longer chains run in passes
constexpr int kMaxCascade = 8;  // Cascade<[Biquad; N]> shares delay lines: single launch

__device__ __forceinline__ int64_t mulw(int32_t c, int32_t v) { return int64_t(c) * int64_t(v); }
__device__ __forceinline__ int64_t wadd(int64_t a, int64_t b) { return int64_t(uint64_t(a) + uint64_t(b)); }
// low 32 bits of (acc >> f) for 0 <= f < 32: one v_alignbit_b32
__device__ __forceinline__ int32_t shr_lo(int64_t acc, int f)
{
    return int32_t(__builtin_amdgcn_alignbit(uint32_t(uint64_t(acc) >> 32), uint32_t(uint64_t(acc)), uint32_t(f)));
}
__device__ __forceinline__ int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
// num_traits::clamp on floats: NaN input passes through
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ------------------------------------------------------------ parameter PODs
struct SecI32 {
    int32_t ba[5];
    int32_t frac;
    int32_t u, mn, mx;
};
struct SecF32 {
    float ba[5];
    float u, mn, mx;
};
struct SecF64 {
    double ba[5];
    double u, mn, mx;
};
template <class S, int N>
struct ChainParams {
    S sec[N];
};

// ------------------------------------------------------------ i32 sections
// src/iir/biquad.rs:366-383 (C = Q<i32,i64,F>): acc = b0*x0 + b1*x1 + b2*x2 +
// a1*y1 + a2*y2 in i64, y0 = (acc >> F) as i32.
__device__ __forceinline__ int64_t sum5(const SecI32 &c, int32_t x0, int32_t x1, int32_t x2, int32_t y1, int32_t y2)
{
    int64_t acc = mulw(c.ba[0], x0);
    acc = wadd(acc, mulw(c.ba[1], x1));
    acc = wadd(acc, mulw(c.ba[2], x2));
    acc = wadd(acc, mulw(c.ba[3], y1));
    acc = wadd(acc, mulw(c.ba[4], y2));
    return acc;
}

template <bool CLAMP>
struct Df1I32 {
    static constexpr bool kClamp = CLAMP;
    static constexpr bool SWEEP_BIG_TWO_BARRIER = !CLAMP;
#ifdef IDSP_EXP_DF1_RING  // experiment: ring depth of the plain i32 DF1 (co-residency of two workgroups per CU against the ring's LDS)
    static constexpr int LDS_RING = CLAMP ? 4 : IDSP_EXP_DF1_RING;
#else
    static constexpr int LDS_RING = CLAMP ? 4 : 7;   // tools/tune_lds.hip: best worst-case over four output placements (plain: 5, 6, 7 within 1.5 %)
#endif
    static constexpr bool LDS_RUN = false;
    static constexpr int LDS_MAX_N = 3;     // serial sections up to which the LDS-DMA kernel beats the register window
    using T = int32_t;
    using Sec = SecI32;
    static constexpr int W = 4;  // x0 x1 y0 y1
    static constexpr int COST = 50;  // ~VALU cycles per sample and wave (5 quarter-rate 64-bit MADs)
    static __device__ __forceinline__ int32_t step(const SecI32 &c, uint32_t (&s)[W], int32_t x0)
    {
        int32_t y0 = shr_lo(sum5(c, x0, int32_t(s[0]), int32_t(s[1]), int32_t(s[2]), int32_t(s[3])), c.frac);
        if (CLAMP) y0 = clampi(int32_t(uint32_t(y0) + uint32_t(c.u)), c.mn, c.mx);  // biquad.rs:394-404
        s[1] = s[0];
        s[0] = uint32_t(x0);
        s[3] = s[2];
        s[2] = uint32_t(y0);
        return y0;
    }
    // A whole tile of samples at once (round 5): SB independent lanes (states st[0 .. SB)), R consecutive samples of each, x and y in the
    // order g = r SB + lane.  step() is ONE chain of five dependent v_mad_i64_i32 per sample, and an in-order wave that has nothing else
    // to issue pays the ~10-cycle latency of each (86 cycles per sample measured where the arithmetic is 25).  The wrapping i64 sum of
    // biquad.rs:366-383 is associative and commutative mod 2^64, so the same bits come out of any order: here the three feed-forward
    // products of ALL samples go first, term by term across the tile (independent chains, back to back), and only a1 y1 — with a2 y2
    // issued beside it — and the shift stay on the per-sample chain.
    static constexpr bool HAS_TILE = true;
    template <int SB, int R, class StateOf>
    static __device__ __forceinline__ void tile(const SecI32 &c, StateOf &&state_of, const int32_t (&x)[SB * R], int32_t (&y)[SB * R])
    {
        int64_t acc[SB * R];
#pragma unroll
        for (int g = 0; g < SB * R; g++) acc[g] = mulw(c.ba[0], x[g]);
#pragma unroll
        for (int g = 0; g < SB * R; g++) acc[g] = wadd(acc[g], mulw(c.ba[1], g >= SB ? x[g - SB] : int32_t(state_of(g % SB)[0])));
#pragma unroll
        for (int g = 0; g < SB * R; g++)
            acc[g] = wadd(acc[g], mulw(c.ba[2], g >= 2 * SB ? x[g - 2 * SB] : int32_t(state_of(g % SB)[g >= SB ? 0 : 1])));
#pragma unroll
        for (int r = 0; r < R; r++) {
#pragma unroll
            for (int l = 0; l < SB; l++) {
                const int g = r * SB + l;
                acc[g] = wadd(acc[g], mulw(c.ba[4], r >= 2 ? y[g - 2 * SB] : int32_t(state_of(l)[r == 1 ? 2 : 3])));
            }
#pragma unroll
            for (int l = 0; l < SB; l++) {
                const int g = r * SB + l;
                acc[g] = wadd(acc[g], mulw(c.ba[3], r >= 1 ? y[g - SB] : int32_t(state_of(l)[2])));
            }
#pragma unroll
            for (int l = 0; l < SB; l++) {
                const int g = r * SB + l;
                int32_t y0 = shr_lo(acc[g], c.frac);
                if (CLAMP) y0 = clampi(int32_t(uint32_t(y0) + uint32_t(c.u)), c.mn, c.mx);
                y[g] = y0;
            }
        }
#pragma unroll
        for (int l = 0; l < SB; l++) {
            uint32_t(&s)[W] = state_of(l);
            const uint32_t nx1 = R >= 2 ? uint32_t(x[(R - 2) * SB + l]) : s[0], ny1 = R >= 2 ? uint32_t(y[(R - 2) * SB + l]) : s[2];
            s[1] = nx1, s[0] = uint32_t(x[(R - 1) * SB + l]);
            s[3] = ny1, s[2] = uint32_t(y[(R - 1) * SB + l]);
        }
    }
};

// src/iir/biquad.rs:511-538 (first-order error feedback)
template <bool CLAMP>
struct DitherI32 {
    static constexpr bool kClamp = CLAMP;
    static constexpr int LDS_RING = CLAMP ? 4 : 7;   // tools/tune_lds.hip: best worst-case over four output placements
    static constexpr bool LDS_RUN = false;
    static constexpr int LDS_MAX_N = 2;     // serial sections up to which the LDS-DMA kernel beats the register window
    using T = int32_t;
    using Sec = SecI32;
    static constexpr int W = 5;  // x0 x1 y0 y1 e
    static constexpr int COST = 60;
    static __device__ __forceinline__ int32_t step(const SecI32 &c, uint32_t (&s)[W], int32_t x0)
    {
        int64_t acc = wadd(int64_t(uint64_t(s[4])), sum5(c, x0, int32_t(s[0]), int32_t(s[1]), int32_t(s[2]), int32_t(s[3])));
        const int sh = 32 - c.frac;  // 1..32
        const uint64_t a = uint64_t(acc) << sh;
        // `(acc as u32) >> (32 - F)`: low word of a; F == 0 leaves a zero low word
        s[4] = c.frac == 0 ? 0u : (uint32_t(a) >> sh);
        int32_t y0 = int32_t(uint32_t(a >> 32));
        if (CLAMP) y0 = clampi(int32_t(uint32_t(y0) + uint32_t(c.u)), c.mn, c.mx);
        s[1] = s[0];
        s[0] = uint32_t(x0);
        s[3] = s[2];
        s[2] = uint32_t(y0);
        return y0;
    }
};

// src/iir/biquad.rs:456-480 (64-bit y state)
template <bool CLAMP>
struct WideI32 {
    static constexpr bool kClamp = CLAMP;
    using T = int32_t;
    static constexpr int LDS_RING = CLAMP ? 4 : 5;   // tools/tune_lds.hip: best worst-case over four output placements
    static constexpr bool LDS_RUN = !CLAMP;
    static constexpr int LDS_MAX_N = 2;     // serial sections up to which the LDS-DMA kernel beats the register window
    using Sec = SecI32;
    static constexpr int W = 6;  // x0 x1 y0.lo y0.hi y1.lo y1.hi
    static constexpr int COST = 100;
    static __device__ __forceinline__ int32_t step(const SecI32 &c, uint32_t (&s)[W], int32_t x0)
    {
        int64_t acc = mulw(c.ba[0], x0);
        acc = wadd(acc, mulw(c.ba[1], int32_t(s[0])));
        acc = wadd(acc, mulw(c.ba[2], int32_t(s[1])));
        s[1] = s[0];
        s[0] = uint32_t(x0);
        // (y.lo as u32 as i64 * a) >> 32, then (y >> 32) as i32 as i64 * a
        acc = wadd(acc, (int64_t(uint64_t(s[2])) * int64_t(c.ba[3])) >> 32);
        acc = wadd(acc, mulw(int32_t(s[3]), c.ba[3]));
        acc = wadd(acc, (int64_t(uint64_t(s[4])) * int64_t(c.ba[4])) >> 32);
        acc = wadd(acc, mulw(int32_t(s[5]), c.ba[4]));
        const uint64_t a = uint64_t(acc) << (32 - c.frac);
        s[4] = s[2];
        s[5] = s[3];
        s[2] = uint32_t(a);
        s[3] = uint32_t(a >> 32);
        int32_t y0 = int32_t(s[3]);
        if (CLAMP) {
            y0 = clampi(int32_t(uint32_t(y0) + uint32_t(c.u)), c.mn, c.mx);
            s[3] = uint32_t(y0);  // y[0] = (y0 << 32) | y[0] as u32
        }
        return y0;
    }
};

// ------------------------------------------------------------ f32 sections
// src/iir/biquad.rs:366-383 (C = T = A = f32)
template <bool CLAMP>
struct Df1F32 {
    static constexpr bool kClamp = CLAMP;
    static constexpr int LDS_RING = CLAMP ? 4 : 5;   // tools/tune_lds.hip: best worst-case over four output placements
    static constexpr bool LDS_RUN = false;
    static constexpr int LDS_MAX_N = 2;     // serial sections up to which the LDS-DMA kernel beats the register window
    using T = float;
    using Sec = SecF32;
    static constexpr int W = 4;
    static constexpr int COST = 24;  // 9 full-rate f32 ops + moves
    static __device__ __forceinline__ float step(const SecF32 &c, uint32_t (&s)[W], float x0)
    {
        float acc = c.ba[0] * x0;
        acc = acc + c.ba[1] * __uint_as_float(s[0]);
        acc = acc + c.ba[2] * __uint_as_float(s[1]);
        acc = acc + c.ba[3] * __uint_as_float(s[2]);
        acc = acc + c.ba[4] * __uint_as_float(s[3]);
        if (CLAMP) acc = clampf(acc + c.u, c.mn, c.mx);
        s[1] = s[0];
        s[0] = __float_as_uint(x0);
        s[3] = s[2];
        s[2] = __float_as_uint(acc);
        return acc;
    }
};

// src/iir/biquad.rs:418-440
template <bool CLAMP>
struct Df2tF32 {
    static constexpr bool kClamp = CLAMP;
    static constexpr bool SWEEP_BIG_TWO_BARRIER = !CLAMP;
    static constexpr int LDS_RING = CLAMP ? 4 : 7;   // tools/tune_lds.hip: best worst-case over four output placements
    static constexpr bool LDS_RUN = false;
    static constexpr int LDS_MAX_N = 2;     // serial sections up to which the LDS-DMA kernel beats the register window
    using T = float;
    using Sec = SecF32;
    static constexpr int W = 2;  // s0 s1
    static constexpr int COST = 22;
    static __device__ __forceinline__ float step(const SecF32 &c, uint32_t (&s)[W], float x0)
    {
        float y0 = __uint_as_float(s[0]) + c.ba[0] * x0;
        if (CLAMP) y0 = clampf(y0 + c.u, c.mn, c.mx);
        const float n0 = (__uint_as_float(s[1]) + c.ba[1] * x0) + c.ba[3] * y0;
        const float n1 = c.ba[2] * x0 + c.ba[4] * y0;
        s[0] = __float_as_uint(n0);
        s[1] = __float_as_uint(n1);
        return y0;
    }
};

// ------------------------------------------------------------ f64 sections
// Same generic impls with C = T = A = f64; a value is two state words (low first).
__device__ __forceinline__ double ldd(const uint32_t *s, int i)
{
    return __builtin_bit_cast(double, uint64_t(s[2 * i]) | (uint64_t(s[2 * i + 1]) << 32));
}
__device__ __forceinline__ void std_(uint32_t *s, int i, double v)
{
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    s[2 * i] = uint32_t(u), s[2 * i + 1] = uint32_t(u >> 32);
}
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <bool CLAMP>
struct Df1F64 {
    static constexpr bool kClamp = CLAMP;
    using T = double;
    using Sec = SecF64;
    static constexpr int W = 8;
    static constexpr int COST = 60;
    static __device__ __forceinline__ double step(const SecF64 &c, uint32_t (&s)[W], double x0)
    {
        double acc = c.ba[0] * x0;
        acc = acc + c.ba[1] * ldd(s, 0);
        acc = acc + c.ba[2] * ldd(s, 1);
        acc = acc + c.ba[3] * ldd(s, 2);
        acc = acc + c.ba[4] * ldd(s, 3);
        if (CLAMP) acc = clampd(acc + c.u, c.mn, c.mx);
        s[2] = s[0], s[3] = s[1];
        std_(s, 0, x0);
        s[6] = s[4], s[7] = s[5];
        std_(s, 2, acc);
        return acc;
    }
};

template <bool CLAMP>
struct Df2tF64 {
    static constexpr bool kClamp = CLAMP;
    using T = double;
    using Sec = SecF64;
    static constexpr int W = 4;
    static constexpr int COST = 55;
    static __device__ __forceinline__ double step(const SecF64 &c, uint32_t (&s)[W], double x0)
    {
        double y0 = ldd(s, 0) + c.ba[0] * x0;
        if (CLAMP) y0 = clampd(y0 + c.u, c.mn, c.mx);
        const double n0 = (ldd(s, 1) + c.ba[1] * x0) + c.ba[3] * y0;
        const double n1 = c.ba[2] * x0 + c.ba[4] * y0;
        std_(s, 0, n0);
        std_(s, 1, n1);
        return y0;
    }
};

// ------------------------------------------------------------ normal form
// `Normal<C>` x `DirectForm1<T>` (src/iir/normal.rs:37-58); ba = [b0, b1, b2, p.re, p.im], state words
// {x0, x1, y0, y1}: y1' = (b0 x0 + b1 x1 + b2 x2 + re y1 + (-im) y0).as_(), y0' = (im y1 + re y0).as_().
struct NormalI32 {
    static constexpr int SWEEP_LPT = 8;
    using T = int32_t;
    static constexpr int LDS_RING = 5;   // tools/tune_lds.hip: best worst-case over four output placements
    static constexpr bool LDS_RUN = true;
    static constexpr int LDS_MAX_N = 2;     // serial sections up to which the LDS-DMA kernel beats the register window
    using Sec = SecI32;
    static constexpr bool kClamp = false;
    static constexpr int W = 4;
    static constexpr int COST = 70;
    static __device__ __forceinline__ int32_t step(const SecI32 &c, uint32_t (&s)[W], int32_t x0)
    {
        const int32_t y0o = int32_t(s[2]), y1o = int32_t(s[3]);
        int64_t acc = mulw(c.ba[0], x0);
        acc = wadd(acc, mulw(c.ba[1], int32_t(s[0])));
        acc = wadd(acc, mulw(c.ba[2], int32_t(s[1])));
        acc = wadd(acc, mulw(c.ba[3], y1o));
        acc = wadd(acc, mulw(int32_t(0u - uint32_t(c.ba[4])), y0o));  // `-self.p.im()` wraps like Neg on Q
        const int32_t y1 = shr_lo(acc, c.frac);
        const int32_t y0 = shr_lo(wadd(mulw(c.ba[4], y1o), mulw(c.ba[3], y0o)), c.frac);
        s[1] = s[0];
        s[0] = uint32_t(x0);
        s[2] = uint32_t(y0);
        s[3] = uint32_t(y1);
        return y0;
    }
};
struct NormalF32 {
    static constexpr int SWEEP_LPT = 8;
    static constexpr bool SWEEP_BIG_TWO_BARRIER = true;
    static constexpr bool SWEEP_UNPACED = true;
    using T = float;
    static constexpr int LDS_RING = 5;   // tools/tune_lds.hip: best worst-case over four output placements
    static constexpr bool LDS_RUN = true;
    static constexpr int LDS_MAX_N = 2;     // serial sections up to which the LDS-DMA kernel beats the register window
    using Sec = SecF32;
    static constexpr bool kClamp = false;
    static constexpr int W = 4;
    static constexpr int COST = 30;
    static __device__ __forceinline__ float step(const SecF32 &c, uint32_t (&s)[W], float x0)
    {
        const float y0o = __uint_as_float(s[2]), y1o = __uint_as_float(s[3]);
        float acc = c.ba[0] * x0;
        acc = acc + c.ba[1] * __uint_as_float(s[0]);
        acc = acc + c.ba[2] * __uint_as_float(s[1]);
        acc = acc + c.ba[3] * y1o;
        acc = acc + (-c.ba[4]) * y0o;
        const float y0 = c.ba[4] * y1o + c.ba[3] * y0o;
        s[1] = s[0];
        s[0] = __float_as_uint(x0);
        s[2] = __float_as_uint(y0);
        s[3] = __float_as_uint(acc);
        return y0;
    }
};
struct NormalF64 {
    using T = double;
    using Sec = SecF64;
    static constexpr bool kClamp = false;
    static constexpr int W = 8;
    static constexpr int COST = 70;
    static __device__ __forceinline__ double step(const SecF64 &c, uint32_t (&s)[W], double x0)
    {
        const double y0o = ldd(s, 2), y1o = ldd(s, 3);
        double acc = c.ba[0] * x0;
        acc = acc + c.ba[1] * ldd(s, 0);
        acc = acc + c.ba[2] * ldd(s, 1);
        acc = acc + c.ba[3] * y1o;
        acc = acc + (-c.ba[4]) * y0o;
        const double y0 = c.ba[4] * y1o + c.ba[3] * y0o;
        s[2] = s[0], s[3] = s[1];
        std_(s, 0, x0);
        std_(s, 2, y0);
        std_(s, 3, acc);
        return y0;
    }
};

// ------------------------------------------------------------ processors
// LDS-DMA ring depth of a one-section processor: the section's LDS_RING if it names one, else 8
template <class Sec, class = void>
struct SecRing {
    static constexpr int value = 8;
};
template <class Sec>
struct SecRing<Sec, std::void_t<decltype(Sec::LDS_RING)>> {
    static constexpr int value = Sec::LDS_RING;
};

template <class Sec, class = void>
struct SecRun {
    static constexpr bool value = false;
};
template <class Sec>
struct SecRun<Sec, std::void_t<decltype(Sec::LDS_RUN)>> {
    static constexpr bool value = Sec::LDS_RUN;
};

// Section count up to which a chain of this section type runs faster on the LDS-DMA kernel than on the register-window
// kernel at the C2 shape, both at their worst output placement (tools/tune_lds.hip): i32 DF1 x3 0.77 vs 0.62, x4 0.61 vs
// 0.68; wide x2 0.73 vs 0.65; f32 DF1 / DF2T x2 0.75-0.77 vs 0.60-0.69, x3 0.70 vs 0.71, x4 0.54-0.57 vs 0.57-0.62;
// Normal x2 0.77-0.81 vs 0.66-0.70, x4 0.51 vs 0.53.
template <class Sec, class = void>
struct SecLdsMaxN {
    static constexpr int value = 0;  // f64 sections: 8-byte samples never take the LDS-DMA kernel
};
template <class Sec>
struct SecLdsMaxN<Sec, std::void_t<decltype(Sec::LDS_MAX_N)>> {
    static constexpr int value = Sec::LDS_MAX_N;
};

// N independent sections in series (`[C] x [S]`, compose.rs:43-77).
// largest number of sub-blocks per workgroup a single section is instantiated with on the sweep kernel: 16 (2^20 lanes in one sweep) for the
// biquad sections proper, Sec::SWEEP_LPT for the others (Normal: 8)
template <class Sec, class = void>
struct SecSweepLpt {
    static constexpr int value = 16;
};
template <class Sec>
struct SecSweepLpt<Sec, std::void_t<decltype(Sec::SWEEP_LPT)>> {
    static constexpr int value = Sec::SWEEP_LPT;
};
// sections whose single-section processors run 8 / 16 blocks per workgroup of the sweep kernel on its two-barrier schedule (fm_sweep.h)
template <class Sec, class = void>
struct SecBigTwoBarrier : std::false_type {};
template <class Sec>
struct SecBigTwoBarrier<Sec, std::void_t<decltype(Sec::SWEEP_BIG_TWO_BARRIER)>> : std::integral_constant<bool, Sec::SWEEP_BIG_TWO_BARRIER> {};
// sections whose full blocks the sweep kernel runs WITHOUT the `s_sleep` pacing although they are cheap: the clamped ones and `Normal` (65536 lanes: f32 DF1 clamp
// 0.67 of the peak paced, 0.76 unpaced; f32 DF2T clamp 0.69 / 0.75; i32 DF1 clamp 0.73 / 0.75; Normal f32 0.73 / 0.75 — the unclamped ones 0.77-0.80 / 0.76)
template <class Sec, class = void>
struct SecUnpaced : std::integral_constant<bool, Sec::kClamp> {};
template <class Sec>
struct SecUnpaced<Sec, std::void_t<decltype(Sec::SWEEP_UNPACED)>> : std::integral_constant<bool, Sec::SWEEP_UNPACED || Sec::kClamp> {};
template <class Sec, class = void>
struct SecHasTile : std::false_type {};
template <class Sec>
struct SecHasTile<Sec, std::void_t<decltype(Sec::HAS_TILE)>> : std::integral_constant<bool, Sec::HAS_TILE> {};

template <class Sec, int N>
struct Chain {
    using In = typename Sec::T;
    using Out = typename Sec::T;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 0;
    static constexpr int IN_DIV = 1;
    static constexpr int COST = N * Sec::COST;
    static constexpr int LDS_RING = N == 1 ? SecRing<Sec>::value : 4;  // N >= 2: tools/tune_lds.hip (worst placement 0.75-0.77 against 0.68 at 8 tiles for N = 2)
    static constexpr bool LDS_RUN = N == 1 && SecRun<Sec>::value;
    static constexpr bool LDS_ELIGIBLE = N <= SecLdsMaxN<Sec>::value;
    static constexpr int SWEEP_MAX_LPT = N == 1 ? SecSweepLpt<Sec>::value : N == 2 ? 8 : 2;  // sub-blocks per workgroup on the sweep kernel (fm_sweep.h)
    static constexpr bool SWEEP_BIG_TWO_BARRIER = N == 1 && SecBigTwoBarrier<Sec>::value;
    static constexpr bool SWEEP_UNPACED = SecUnpaced<Sec>::value;
    using Params = ChainParams<typename Sec::Sec, N>;
    uint32_t s[N][Sec::W];

    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t lanes, size_t lane)
    {
#pragma unroll
        for (int k = 0; k < N; k++)
#pragma unroll
            for (int w = 0; w < Sec::W; w++) s[k][w] = st[size_t(k * Sec::W + w) * lanes + lane];
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane)
    {
#pragma unroll
        for (int k = 0; k < N; k++)
#pragma unroll
            for (int w = 0; w < Sec::W; w++) st[size_t(k * Sec::W + w) * lanes + lane] = s[k][w];
    }
    __device__ __forceinline__ Out step(const Params &p, In x)
    {
#pragma unroll
        for (int k = 0; k < N; k++) x = Sec::step(p.sec[k], s[k], x);
        return x;
    }
    // tile form of a single section that has one (Sec::tile; lane_stream.h: HasTileOf): `p` points at SB chains of independent lanes
    static constexpr bool HAS_TILE = N == 1 && SecHasTile<Sec>::value;
    template <int SB, int R>
    static __device__ __forceinline__ void tile(const Params &prm, Chain *p, const In (&x)[SB * R], Out (&y)[SB * R])
    {
        if constexpr (HAS_TILE) {
            Sec::template tile<SB, R>(prm.sec[0], [&](int l) -> uint32_t(&)[Sec::W] { return p[l].s[0]; }, x, y);
        } else {
#pragma unroll
            for (int g = 0; g < SB * R; g++) y[g] = p[g % SB].step(prm, x[g]);
        }
    }
};

// `Cascade<[Biquad<C>; N]>` x `DirectForm<T, N>` (biquad.rs:339-364): the input
// history of section k is the output history of section k-1.
// Values: h[0..1] = x, h[2+2k..3+2k] = y[k]; sizeof(T)/4 state words per value.
template <class T, int N>
struct CascadeDf1 {
    using In = T;
    using Out = T;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 0;
    static constexpr int IN_DIV = 1;
    static constexpr bool kFloat = std::is_floating_point<T>::value;
    static constexpr int COST = N * (std::is_same<T, float>::value ? 24 : (kFloat ? 60 : 50));
    static constexpr int LDS_RING = 4;  // tools/tune_lds.hip: worst placement 0.76-0.78 (N = 2) against 0.67-0.68 at 8 tiles
    // LDS-DMA kernel against register window at the C2 shape: i32 x3 0.78 vs 0.59, x4 0.69 vs 0.65, x8 0.42 vs 0.50;
    // f32 x2 0.76 vs 0.70, x4 0.59 vs 0.62
    static constexpr bool LDS_ELIGIBLE = std::is_same<T, int32_t>::value ? N <= 4 : (std::is_same<T, float>::value && N <= 2);
    static constexpr int SWEEP_MAX_LPT = N == 1 ? 4 : 2;
    using SecT = typename std::conditional<std::is_same<T, float>::value, SecF32,
                                           typename std::conditional<kFloat, SecF64, SecI32>::type>::type;
    using Params = ChainParams<SecT, N>;
    static constexpr int V = 2 + 2 * N;       // values
    static constexpr int VW = sizeof(T) / 4;  // words per value
    T h[V];

    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t lanes, size_t lane)
    {
#pragma unroll
        for (int v = 0; v < V; v++) {
            uint32_t w[VW];
#pragma unroll
            for (int k = 0; k < VW; k++) w[k] = st[size_t(v * VW + k) * lanes + lane];
            h[v] = words_to<T>(w);
        }
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane)
    {
#pragma unroll
        for (int v = 0; v < V; v++) {
            uint32_t w[VW];
            to_words<T>(h[v], w);
#pragma unroll
            for (int k = 0; k < VW; k++) st[size_t(v * VW + k) * lanes + lane] = w[k];
        }
    }
    __device__ __forceinline__ Out step(const Params &p, In x0)
    {
#pragma unroll
        for (int k = 0; k < N; k++) {
            T *xh = &h[2 * k];
            const T *yh = &h[2 * k + 2];
            T y0;
            if constexpr (kFloat) {
                const SecT &c = p.sec[k];
                T acc = c.ba[0] * x0;
                acc = acc + c.ba[1] * xh[0];
                acc = acc + c.ba[2] * xh[1];
                acc = acc + c.ba[3] * yh[0];
                acc = acc + c.ba[4] * yh[1];
                y0 = acc;
            } else {
                const SecI32 &c = p.sec[k];
                y0 = shr_lo(sum5(c, x0, xh[0], xh[1], yh[0], yh[1]), c.frac);
            }
            xh[1] = xh[0];
            xh[0] = x0;
            x0 = y0;
        }
        h[2 * N + 1] = h[2 * N];
        h[2 * N] = x0;
        return x0;
    }
};

// `ByLane<[C; lanes]>` (dsp-process/src/compose.rs:363-390): every lane filters with its
// own coefficients.  Coefficient planes are lane-contiguous like the state planes:
// value v of section k of lane l is coef[(k * CV + v) * lanes + l] (an i32/f32/f64),
// CV = 5 (ba) or 8 (ba, u, min, max); they are read once into registers.
template <class S, class T>
__device__ __forceinline__ void unpack_sec(S &c, const T *v, int nv, T lo, T hi)
{
#pragma unroll
    for (int i = 0; i < 5; i++) c.ba[i] = v[i];
    c.u = nv == 8 ? v[5] : T(0);
    c.mn = nv == 8 ? v[6] : lo;
    c.mx = nv == 8 ? v[7] : hi;
}
__device__ __forceinline__ void unpack_sec(SecI32 &c, const int32_t *v, int nv, int32_t frac)
{
    unpack_sec(c, v, nv, INT32_MIN, INT32_MAX);
    c.frac = frac;
}
__device__ __forceinline__ void unpack_sec(SecF32 &c, const float *v, int nv, int32_t)
{
    unpack_sec(c, v, nv, -__builtin_inff(), __builtin_inff());
}
__device__ __forceinline__ void unpack_sec(SecF64 &c, const double *v, int nv, int32_t)
{
    unpack_sec(c, v, nv, -__builtin_inf(), __builtin_inf());
}

struct ByLaneParams {
    const void *coef;
    int32_t frac;
};

template <class Sec, int N>
struct ChainByLane {
    using In = typename Sec::T;
    using Out = typename Sec::T;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 0;
    static constexpr int IN_DIV = 1;
    static constexpr int COST = N * Sec::COST;
    static constexpr int LDS_RING = N == 1 ? SecRing<Sec>::value : 4;  // N >= 2: tools/tune_lds.hip (worst placement 0.75-0.77 against 0.68 at 8 tiles for N = 2)
    static constexpr bool LDS_RUN = N == 1 && SecRun<Sec>::value;
    static constexpr bool LDS_ELIGIBLE = N <= SecLdsMaxN<Sec>::value;
    static constexpr int CV = Sec::kClamp ? 8 : 5;
    // sub-blocks per workgroup on the sweep kernel (fm_sweep.h): every lane carries its own coefficients in registers beside its state
    static constexpr int SWEEP_MAX_LPT = N == 1 ? 8 : 2;
    static constexpr bool SWEEP_BIG_TWO_BARRIER = N == 1 && SecBigTwoBarrier<Sec>::value && std::is_same<typename Sec::T, float>::value;  // (i32 DF1 by lane: one barrier)
    // never paced: with the coefficient registers the paced schedule is bimodal on some boxes (65536 lanes, i32 DF1 / f32 DF2T by lane: launches of 0.345 and of
    // 0.41-0.43 ms in one process, median 0.41; unpaced 0.35-0.36 every time; other boxes 0.34 paced)
    static constexpr bool SWEEP_UNPACED = true;
    using Params = ByLaneParams;
    uint32_t s[N][Sec::W];
    typename Sec::Sec c[N];

    __device__ __forceinline__ void load(const Params &p, const uint32_t *st, size_t lanes, size_t lane)
    {
#pragma unroll
        for (int k = 0; k < N; k++) {
#pragma unroll
            for (int w = 0; w < Sec::W; w++) s[k][w] = st[size_t(k * Sec::W + w) * lanes + lane];
            In cv[CV];
#pragma unroll
            for (int v = 0; v < CV; v++) cv[v] = static_cast<const In *>(p.coef)[size_t(k * CV + v) * lanes + lane];
            unpack_sec(c[k], cv, CV, p.frac);
        }
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane)
    {
#pragma unroll
        for (int k = 0; k < N; k++)
#pragma unroll
            for (int w = 0; w < Sec::W; w++) st[size_t(k * Sec::W + w) * lanes + lane] = s[k][w];
    }
    __device__ __forceinline__ Out step(const Params &, In x)
    {
#pragma unroll
        for (int k = 0; k < N; k++) x = Sec::step(c[k], s[k], x);
        return x;
    }
};

// ------------------------------------------------------------ host dispatch
// Row pitches of the `_pitch` entry points (include/idsp_hip.h): elements between the starts of consecutive lanes
// (LANE_MAJOR) or consecutive frames (FRAME_MAJOR); 0 = dense.  A pitch must cover a row, and an in-place call must
// use the same pitch on both sides.
inline int check_pitch(const void *x, const void *y, size_t lanes, size_t frames, int layout, Pitch pitch)
{
    const size_t row = layout == IDSP_LANE_MAJOR ? frames : lanes;
    if ((pitch.x && pitch.x < row) || (pitch.y && pitch.y < row))
        return fail(IDSP_EINVAL, "pitch (%zu, %zu) shorter than a row of %zu elements", pitch.x, pitch.y, row);
    if (x == y && (pitch.x ? pitch.x : row) != (pitch.y ? pitch.y : row))
        return fail(IDSP_EINVAL, "in-place call with different x and y pitches (%zu, %zu)", pitch.x, pitch.y);
    return IDSP_OK;
}

template <class T>
int copy_through(const T *x, T *y, size_t lanes, size_t frames, int layout, hipStream_t s, Pitch pitch)
{
    if (x == y) return IDSP_OK;
    const size_t row = layout == IDSP_LANE_MAJOR ? frames : lanes, rows = layout == IDSP_LANE_MAJOR ? lanes : frames;
    const size_t xp = pitch.x ? pitch.x : row, yp = pitch.y ? pitch.y : row;
    IDSP_HIP_TRY(hipMemcpy2DAsync(y, yp * sizeof(T), x, xp * sizeof(T), row * sizeof(T), rows, hipMemcpyDeviceToDevice, s));
    return IDSP_OK;
}

template <class Sec, int N, class CfgFill>
int run_chain_n(CfgFill fill, size_t first, void *state, const typename Sec::T *x, typename Sec::T *y,
                size_t lanes, size_t frames, int layout, hipStream_t s, Pitch pitch = {})
{
    typename Chain<Sec, N>::Params prm;
    for (int k = 0; k < N; k++) fill(prm.sec[k], first + k);
    uint32_t *st = static_cast<uint32_t *>(state) + first * Sec::W * lanes;
    return launch_stream<Chain<Sec, N>>(prm, st, x, y, lanes, frames, layout, s, pitch);
}

// Sections first .. first + NA + NB - 1 on the two-wave kernel: wave 0 runs Chain<Sec, NA>, wave 1 Chain<Sec, NB> one tile behind
template <class Sec, int NA, int NB, class CfgFill>
int run_chain_duo(CfgFill fill, size_t first, void *state, const typename Sec::T *x, typename Sec::T *y, size_t lanes, size_t frames,
                  hipStream_t s, Pitch pitch)
{
    typename Chain<Sec, NA>::Params pa;
    typename Chain<Sec, NB>::Params pb;
    for (int k = 0; k < NA; k++) fill(pa.sec[k], first + k);
    for (int k = 0; k < NB; k++) fill(pb.sec[k], first + NA + k);
    uint32_t *st = static_cast<uint32_t *>(state) + first * Sec::W * lanes;
    return launch_duo<Chain<Sec, NA>, Chain<Sec, NB>>(pa, pb, st, st + size_t(NA) * Sec::W * lanes, x, y, lanes, frames, s, pitch);
}
// section types that have the two-wave instantiations (4-byte samples)
template <class Sec>
struct DuoSec : std::integral_constant<bool, sizeof(typename Sec::T) == 4> {};
constexpr int kMaxChainDuo = 8;

// n sections in passes of <= kMaxChain (<= kMaxChainDuo on the two-wave kernel): the first pass reads x, later passes
// run in place on y — literally the reference's slice composition.
template <class Sec, class CfgFill>
int run_chain(CfgFill fill, size_t n, void *state, const typename Sec::T *x, typename Sec::T *y,
              size_t lanes, size_t frames, int layout, hipStream_t s, Pitch pitch = {})
{
    using T = typename Sec::T;
    if (lanes == 0 || frames == 0) return IDSP_OK;
    if (n == 0) return copy_through<T>(x, y, lanes, frames, layout, s, pitch);  // empty slice: y.copy_from_slice(x), compose.rs:63-65
    size_t done = 0;
    const T *src = x;
    while (done < n) {
        if constexpr (DuoSec<Sec>::value) {
            const size_t md = n - done < size_t(kMaxChainDuo) ? n - done : size_t(kMaxChainDuo);
            // (four sections: +9 % for plain i32 DF1, +20 % for the f32 forms, -1 % / -5 % for the clamp / wide forms, which stay)
            if (duo_wanted(md, lanes, layout) && (md >= 5 || (!Sec::kClamp && Sec::W <= 4))) {
                int rc;
                switch (md) {
                    case 4: rc = run_chain_duo<Sec, 2, 2>(fill, done, state, src, y, lanes, frames, s, pitch); break;
                    case 5: rc = run_chain_duo<Sec, 3, 2>(fill, done, state, src, y, lanes, frames, s, pitch); break;
                    case 6: rc = run_chain_duo<Sec, 3, 3>(fill, done, state, src, y, lanes, frames, s, pitch); break;
                    case 7: rc = run_chain_duo<Sec, 4, 3>(fill, done, state, src, y, lanes, frames, s, pitch); break;
                    default: rc = run_chain_duo<Sec, 4, 4>(fill, done, state, src, y, lanes, frames, s, pitch); break;
                }
                if (rc) return rc;
                done += md;
                src = y;
                pitch.x = pitch.y;
                continue;
            }
        }
        const size_t m = n - done < size_t(kMaxChain) ? n - done : size_t(kMaxChain);
        int rc;
        switch (m) {
            case 1: rc = run_chain_n<Sec, 1>(fill, done, state, src, y, lanes, frames, layout, s, pitch); break;
            case 2: rc = run_chain_n<Sec, 2>(fill, done, state, src, y, lanes, frames, layout, s, pitch); break;
            case 3: rc = run_chain_n<Sec, 3>(fill, done, state, src, y, lanes, frames, layout, s, pitch); break;
            default: rc = run_chain_n<Sec, 4>(fill, done, state, src, y, lanes, frames, layout, s, pitch); break;
        }
        if (rc) return rc;
        done += m;
        src = y;
        pitch.x = pitch.y;  // later passes run in place on y
    }
    return IDSP_OK;
}

inline int check_frac(int frac, size_t k)
{
    // `const { assert!(F >= 0 && F < 32) }` biquad.rs:448-450,513-515
    if (frac < 0 || frac > 31) return fail(IDSP_EINVAL, "section %zu: frac = %d not in 0..31", k, frac);
    return IDSP_OK;
}

constexpr int kMaxChainByLane = 2;  // coefficient registers come on top of the state

// n per-lane sections in passes of <= kMaxChainByLane (stage-major like run_chain)
template <class Sec>
int run_chain_bylane(const void *coef, int frac, size_t n, void *state, const typename Sec::T *x, typename Sec::T *y,
                     size_t lanes, size_t frames, int layout, hipStream_t s, Pitch pitch = {})
{
    using T = typename Sec::T;
    if (lanes == 0 || frames == 0) return IDSP_OK;
    if (n == 0) return copy_through<T>(x, y, lanes, frames, layout, s, pitch);
    constexpr int CV = ChainByLane<Sec, 1>::CV;
    size_t done = 0;
    const T *src = x;
    while (done < n) {
        if constexpr (sizeof(T) == 4) {
            // three or four sections of a bank in one pass on the two-wave kernel (two sections per wave; coefficient and state
            // planes of the second wave start two sections further into the records)
            const size_t md = n - done < size_t(2 * kMaxChainByLane) ? n - done : size_t(2 * kMaxChainByLane);
            if (md >= 3 && duo_wanted(5, lanes, layout)) {
                ByLaneParams pa{static_cast<const T *>(coef) + done * CV * lanes, frac}, pb{static_cast<const T *>(coef) + (done + 2) * CV * lanes, frac};
                uint32_t *sa = static_cast<uint32_t *>(state) + done * Sec::W * lanes, *sb = sa + 2 * Sec::W * lanes;
                const int rc = md == 3 ? launch_duo<ChainByLane<Sec, 2>, ChainByLane<Sec, 1>>(pa, pb, sa, sb, src, y, lanes, frames, s, pitch)
                                       : launch_duo<ChainByLane<Sec, 2>, ChainByLane<Sec, 2>>(pa, pb, sa, sb, src, y, lanes, frames, s, pitch);
                if (rc) return rc;
                done += md;
                src = y;
                pitch.x = pitch.y;
                continue;
            }
        }
        const size_t m = n - done < size_t(kMaxChainByLane) ? n - done : size_t(kMaxChainByLane);
        ByLaneParams prm{static_cast<const T *>(coef) + done * CV * lanes, frac};
        uint32_t *st = static_cast<uint32_t *>(state) + done * Sec::W * lanes;
        const int rc = m == 1 ? launch_stream<ChainByLane<Sec, 1>>(prm, st, src, y, lanes, frames, layout, s, pitch)
                              : launch_stream<ChainByLane<Sec, 2>>(prm, st, src, y, lanes, frames, layout, s, pitch);
        if (rc) return rc;
        done += m;
        src = y;
        pitch.x = pitch.y;
    }
    return IDSP_OK;
}

template <class Sec>
int entry_bylane(const void *coef, int frac, size_t n, void *state, const typename Sec::T *x, typename Sec::T *y,
                 size_t lanes, size_t frames, int layout, void *stream, Pitch pitch = {})
{
    int rc = check_stream_args(coef, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if ((rc = check_pitch(x, y, lanes, frames, layout, pitch))) return rc;
    if (std::is_same<typename Sec::T, int32_t>::value && (rc = check_frac(frac, 0))) return rc;
    return run_chain_bylane<Sec>(coef, frac, n, state, x, y, lanes, frames, layout, as_stream(stream), pitch);
}

template <class T, int N, class CfgFill>
int run_cascade_n(CfgFill fill, void *state, const T *x, T *y, size_t lanes, size_t frames, int layout, hipStream_t s, Pitch pitch)
{
    typename CascadeDf1<T, N>::Params prm;
    for (int k = 0; k < N; k++) fill(prm.sec[k], size_t(k));
    return launch_stream<CascadeDf1<T, N>>(prm, state, x, y, lanes, frames, layout, s, pitch);
}

// `Cascade` of NA + NB sections on the two-wave kernel.  Section k's input history is section k - 1's output history, so wave 1 is
// the same processor type started 2 NA values into the state record: its "x history" is y[NA - 1]'s, which it keeps up to date
// from the samples wave 0 hands over (both waves write that pair back: the same values).
template <class T, int NA, int NB, class CfgFill>
int run_cascade_duo(CfgFill fill, void *state, const T *x, T *y, size_t lanes, size_t frames, hipStream_t s, Pitch pitch)
{
    typename CascadeDf1<T, NA>::Params pa;
    typename CascadeDf1<T, NB>::Params pb;
    for (int k = 0; k < NA; k++) fill(pa.sec[k], size_t(k));
    for (int k = 0; k < NB; k++) fill(pb.sec[k], size_t(NA + k));
    uint32_t *st = static_cast<uint32_t *>(state);
    return launch_duo<CascadeDf1<T, NA>, CascadeDf1<T, NB>>(pa, pb, st, st + size_t(2 * NA) * (sizeof(T) / 4) * lanes, x, y, lanes, frames, s, pitch);
}

template <class T, class CfgFill>
int run_cascade(CfgFill fill, size_t n, void *state, const T *x, T *y, size_t lanes, size_t frames, int layout,
                hipStream_t s, Pitch pitch = {})
{
    if (int rc = check_pitch(x, y, lanes, frames, layout, pitch)) return rc;
    if (n < 1 || n > size_t(kMaxCascade)) return fail(IDSP_EINVAL, "cascade sections n = %zu not in 1..%d", n, kMaxCascade);
    if (lanes == 0 || frames == 0) return IDSP_OK;
    if constexpr (sizeof(T) == 4) {
        if (n >= 5 && duo_wanted(n, lanes, layout)) {  // (4 sections stay on the LDS-DMA kernel: 0.69 against 0.62)
            switch (n) {
                case 5: return run_cascade_duo<T, 3, 2>(fill, state, x, y, lanes, frames, s, pitch);
                case 6: return run_cascade_duo<T, 3, 3>(fill, state, x, y, lanes, frames, s, pitch);
                case 7: return run_cascade_duo<T, 4, 3>(fill, state, x, y, lanes, frames, s, pitch);
                default: return run_cascade_duo<T, 4, 4>(fill, state, x, y, lanes, frames, s, pitch);
            }
        }
    }
    switch (n) {
        case 1: return run_cascade_n<T, 1>(fill, state, x, y, lanes, frames, layout, s, pitch);
        case 2: return run_cascade_n<T, 2>(fill, state, x, y, lanes, frames, layout, s, pitch);
        case 3: return run_cascade_n<T, 3>(fill, state, x, y, lanes, frames, layout, s, pitch);
        case 4: return run_cascade_n<T, 4>(fill, state, x, y, lanes, frames, layout, s, pitch);
        case 5: return run_cascade_n<T, 5>(fill, state, x, y, lanes, frames, layout, s, pitch);
        case 6: return run_cascade_n<T, 6>(fill, state, x, y, lanes, frames, layout, s, pitch);
        case 7: return run_cascade_n<T, 7>(fill, state, x, y, lanes, frames, layout, s, pitch);
        default: return run_cascade_n<T, 8>(fill, state, x, y, lanes, frames, layout, s, pitch);
    }
}

struct FillI32 {
    const idsp_biquad_i32 *c;
    void operator()(SecI32 &d, size_t k) const
    {
        for (int i = 0; i < 5; i++) d.ba[i] = c[k].ba[i];
        d.frac = c[k].frac;
        d.u = 0;
        d.mn = INT32_MIN;
        d.mx = INT32_MAX;
    }
};
struct FillClampI32 {
    const idsp_biquad_clamp_i32 *c;
    void operator()(SecI32 &d, size_t k) const
    {
        for (int i = 0; i < 5; i++) d.ba[i] = c[k].ba[i];
        d.frac = c[k].frac;
        d.u = c[k].u;
        d.mn = c[k].min;
        d.mx = c[k].max;
    }
};
struct FillF32 {
    const idsp_biquad_f32 *c;
    void operator()(SecF32 &d, size_t k) const
    {
        for (int i = 0; i < 5; i++) d.ba[i] = c[k].ba[i];
        d.u = 0.f;
        d.mn = -__builtin_inff();
        d.mx = __builtin_inff();
    }
};
struct FillClampF32 {
    const idsp_biquad_clamp_f32 *c;
    void operator()(SecF32 &d, size_t k) const
    {
        for (int i = 0; i < 5; i++) d.ba[i] = c[k].ba[i];
        d.u = c[k].u;
        d.mn = c[k].min;
        d.mx = c[k].max;
    }
};

struct FillF64 {
    const idsp_biquad_f64 *c;
    void operator()(SecF64 &d, size_t k) const
    {
        for (int i = 0; i < 5; i++) d.ba[i] = c[k].ba[i];
        d.u = 0.0;
        d.mn = -__builtin_inf();
        d.mx = __builtin_inf();
    }
};
struct FillClampF64 {
    const idsp_biquad_clamp_f64 *c;
    void operator()(SecF64 &d, size_t k) const
    {
        for (int i = 0; i < 5; i++) d.ba[i] = c[k].ba[i];
        d.u = c[k].u;
        d.mn = c[k].min;
        d.mx = c[k].max;
    }
};

template <class Sec, class Cfg, class Fill>
int entry_i32(const Cfg *cfg, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames,
              int layout, void *stream, Pitch pitch = {})
{
    int rc = check_stream_args(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if ((rc = check_pitch(x, y, lanes, frames, layout, pitch))) return rc;
    for (size_t k = 0; k < n; k++)
        if ((rc = check_frac(cfg[k].frac, k))) return rc;
    return run_chain<Sec>(Fill{cfg}, n, state, x, y, lanes, frames, layout, as_stream(stream), pitch);
}

template <class Sec, class Cfg, class Fill>
int entry_f64(const Cfg *cfg, size_t n, void *state, const double *x, double *y, size_t lanes, size_t frames,
              int layout, void *stream, Pitch pitch = {})
{
    int rc = check_stream_args(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if ((rc = check_pitch(x, y, lanes, frames, layout, pitch))) return rc;
    return run_chain<Sec>(Fill{cfg}, n, state, x, y, lanes, frames, layout, as_stream(stream), pitch);
}

template <class Sec, class Cfg, class Fill>
int entry_f32(const Cfg *cfg, size_t n, void *state, const float *x, float *y, size_t lanes, size_t frames,
              int layout, void *stream, Pitch pitch = {})
{
    int rc = check_stream_args(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if ((rc = check_pitch(x, y, lanes, frames, layout, pitch))) return rc;
    return run_chain<Sec>(Fill{cfg}, n, state, x, y, lanes, frames, layout, as_stream(stream), pitch);
}

}  // namespace bq
}  // namespace idsp

// `<entry>_pitch` twin of a shared-coefficient entry point (include/idsp_hip.h): same call with explicit row pitches.
#define IDSP_PITCH_TWIN(name, Cfg, T, ENTRY, ...)                                                                      \
    int name##_pitch(const Cfg *cfg, size_t n, void *state, const T *x, size_t x_pitch, T *y, size_t y_pitch,          \
                     size_t lanes, size_t frames, int layout, void *stream)                                            \
    {                                                                                                                  \
        return ENTRY<__VA_ARGS__>(cfg, n, state, x, y, lanes, frames, layout, stream, ::idsp::Pitch{x_pitch, y_pitch}); \
    }
