// lockin_waves_biquad.hip — `Lockin<[Biquad<C>; n]>` on the multi-wave lock-in kernel of lockin_waves.h: the biquad chain is one more
// arm functor (`Bank`) beside `[Lowpass<N>; K]`.  Phase form (src/lockin.rs:30-39 over src/iir/biquad.rs:366-383) for i32, and the
// external-oscillator form (src/lockin.rs:17-27) for i32 and f32 — the latter is the mix -> lowpass graph of
// examples/ddc_lockin.rs:35-42.  Round 4; the one-thread-per-lane forms (lockin_generic.hip) stay for the shapes this does not take.
#include "biquad_sections.h"
#include "lockin_waves.h"

namespace idsp {
namespace {

// n serial DF1 sections, state words {x0, x1, y0, y1} per section at `word0` (the record of idsp_lockin_biquad_state_words)
template <int NS>
struct BqBank {
    using Params = bq::ChainParams<bq::SecI32, NS>;
    static constexpr int kArmWords = 4 * NS;
    static constexpr bool kSixWaves = false;  // four waves per 64 lanes only: the arm waves are the long path with biquad arms
    static constexpr bool kExtLo = false;
    static __device__ __forceinline__ int32_t mix(int32_t lo, int32_t x) { return __mulhi(lo, x); }
    static const char *name() { return NS == 1 ? "[Biquad; 1]" : NS == 2 ? "[Biquad; 2]" : NS == 3 ? "[Biquad; 3]" : "[Biquad; 4]"; }
    uint32_t s[NS][4];
    __device__ __forceinline__ void load(const uint32_t *st, size_t lanes, size_t lane, int word0)
    {
#pragma unroll
        for (int k = 0; k < NS; k++)
#pragma unroll
            for (int w = 0; w < 4; w++) s[k][w] = st[size_t(word0 + k * 4 + w) * lanes + lane];
    }
    __device__ __forceinline__ void store(uint32_t *st, size_t lanes, size_t lane, int word0) const
    {
#pragma unroll
        for (int k = 0; k < NS; k++)
#pragma unroll
            for (int w = 0; w < 4; w++) st[size_t(word0 + k * 4 + w) * lanes + lane] = s[k][w];
    }
    __device__ __forceinline__ int32_t step(const Params &p, int32_t x)
    {
#pragma unroll
        for (int k = 0; k < NS; k++) x = bq::Df1I32<false>::step(p.sec[k], s[k], x);
        return x;
    }
};

// the same arms fed by a per-sample oscillator
template <int NS>
struct BqLoBank : BqBank<NS> {
    static constexpr bool kExtLo = true;
    static const char *name() { return NS == 1 ? "[Biquad; 1], LO" : NS == 2 ? "[Biquad; 2], LO" : NS == 3 ? "[Biquad; 3], LO" : "[Biquad; 4], LO"; }
};
// `[Biquad<f32>; n]` x `[DirectForm1<f32>; n]` arms on bit patterns: rows, samples and oscillator travel as 32-bit words
template <int NS>
struct BqLoBankF32 {
    using Params = bq::ChainParams<bq::SecF32, NS>;
    static constexpr int kArmWords = 4 * NS;
    static constexpr bool kSixWaves = false, kExtLo = true;
    static const char *name() { return NS == 1 ? "[Biquad<f32>; 1], LO" : NS == 2 ? "[Biquad<f32>; 2], LO" : NS == 3 ? "[Biquad<f32>; 3], LO" : "[Biquad<f32>; 4], LO"; }
    static __device__ __forceinline__ int32_t mix(int32_t lo, int32_t x) { return __float_as_int(__int_as_float(x) * __int_as_float(lo)); }
    uint32_t s[NS][4];
    __device__ __forceinline__ void load(const uint32_t *st, size_t lanes, size_t lane, int word0)
    {
#pragma unroll
        for (int k = 0; k < NS; k++)
#pragma unroll
            for (int w = 0; w < 4; w++) s[k][w] = st[size_t(word0 + k * 4 + w) * lanes + lane];
    }
    __device__ __forceinline__ void store(uint32_t *st, size_t lanes, size_t lane, int word0) const
    {
#pragma unroll
        for (int k = 0; k < NS; k++)
#pragma unroll
            for (int w = 0; w < 4; w++) st[size_t(word0 + k * 4 + w) * lanes + lane] = s[k][w];
    }
    __device__ __forceinline__ int32_t step(const Params &p, int32_t xb)
    {
        float x = __int_as_float(xb);
#pragma unroll
        for (int k = 0; k < NS; k++) x = bq::Df1F32<false>::step(p.sec[k], s[k], x);
        return __float_as_int(x);
    }
};

template <int NS>
int run_lo(const idsp_biquad_i32 *sec, void *state, const int32_t *x, const int32_t *lo, int32_t *y, size_t lanes, size_t frames, int layout,
           hipStream_t s, size_t pitch)
{
    typename BqLoBank<NS>::Params p;
    for (int k = 0; k < NS; k++) {
        for (int i = 0; i < 5; i++) p.sec[k].ba[i] = sec[k].ba[i];
        p.sec[k].frac = sec[k].frac;
        p.sec[k].u = 0, p.sec[k].mn = INT32_MIN, p.sec[k].mx = INT32_MAX;
    }
    return launch_lockin_waves_bank<MODE_IQ, BqLoBank<NS>>(p, static_cast<uint32_t *>(state), x, reinterpret_cast<Cplx *>(y), lanes, frames, layout, 4, s, lo, pitch);
}
template <int NS>
int run_lo_f32(const idsp_biquad_f32 *sec, void *state, const float *x, const float *lo, float *y, size_t lanes, size_t frames, int layout,
               hipStream_t s, size_t pitch)
{
    typename BqLoBankF32<NS>::Params p;
    for (int k = 0; k < NS; k++) {
        for (int i = 0; i < 5; i++) p.sec[k].ba[i] = sec[k].ba[i];
        p.sec[k].u = 0.f, p.sec[k].mn = -__builtin_inff(), p.sec[k].mx = __builtin_inff();
    }
    return launch_lockin_waves_bank<MODE_IQ, BqLoBankF32<NS>>(p, static_cast<uint32_t *>(state), reinterpret_cast<const int32_t *>(x),
                                                              reinterpret_cast<Cplx *>(y), lanes, frames, layout, 4, s,
                                                              reinterpret_cast<const int32_t *>(lo), pitch);
}

template <int NS>
int run(const idsp_biquad_i32 *sec, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, hipStream_t s, size_t pitch)
{
    typename BqBank<NS>::Params p;
    for (int k = 0; k < NS; k++) {
        for (int i = 0; i < 5; i++) p.sec[k].ba[i] = sec[k].ba[i];
        p.sec[k].frac = sec[k].frac;
        p.sec[k].u = 0, p.sec[k].mn = INT32_MIN, p.sec[k].mx = INT32_MAX;
    }
    return launch_lockin_waves_bank<MODE_IQ, BqBank<NS>>(p, static_cast<uint32_t *>(state), x, reinterpret_cast<Cplx *>(y), lanes, frames, layout, 4, s, nullptr, pitch);
}

}  // namespace

// the caller (lockin_generic.hip) has validated the arguments and asked lockin_waves_for() whether the shape is the kernel's
int lockin_waves_biquad_iq(const idsp_biquad_i32 *sec, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                           int layout, hipStream_t s, size_t pitch)
{
    switch (n) {
        case 1: return run<1>(sec, state, x, y, lanes, frames, layout, s, pitch);
        case 2: return run<2>(sec, state, x, y, lanes, frames, layout, s, pitch);
        case 3: return run<3>(sec, state, x, y, lanes, frames, layout, s, pitch);
        default: return run<4>(sec, state, x, y, lanes, frames, layout, s, pitch);
    }
}

int lockin_waves_biquad_lo(const idsp_biquad_i32 *sec, size_t n, void *state, const int32_t *x, const int32_t *lo, int32_t *y, size_t lanes,
                           size_t frames, int layout, hipStream_t s, size_t pitch)
{
    switch (n) {
        case 1: return run_lo<1>(sec, state, x, lo, y, lanes, frames, layout, s, pitch);
        case 2: return run_lo<2>(sec, state, x, lo, y, lanes, frames, layout, s, pitch);
        case 3: return run_lo<3>(sec, state, x, lo, y, lanes, frames, layout, s, pitch);
        default: return run_lo<4>(sec, state, x, lo, y, lanes, frames, layout, s, pitch);
    }
}
int lockin_waves_biquad_lo_f32(const idsp_biquad_f32 *sec, size_t n, void *state, const float *x, const float *lo, float *y, size_t lanes,
                               size_t frames, int layout, hipStream_t s, size_t pitch)
{
    switch (n) {
        case 1: return run_lo_f32<1>(sec, state, x, lo, y, lanes, frames, layout, s, pitch);
        case 2: return run_lo_f32<2>(sec, state, x, lo, y, lanes, frames, layout, s, pitch);
        case 3: return run_lo_f32<3>(sec, state, x, lo, y, lanes, frames, layout, s, pitch);
        default: return run_lo_f32<4>(sec, state, x, lo, y, lanes, frames, layout, s, pitch);
    }
}

}  // namespace idsp
