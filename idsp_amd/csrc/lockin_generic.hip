// lockin_generic.hip — `Lockin<C>` as the reference defines it (src/lockin.rs:11-39): arm filters other than `[Lowpass<N>; K]`
// and the external-LO form `(x, Complex<U>) -> Complex<X>` (:17-27), on the one-thread-per-lane stream kernels of
// lane_stream.h.  (The phase form with lowpass arms — the C4 configuration — has its own multi-wave kernels, lockin_waves.h.)
//
//   idsp_lockin_i32_biquad_process      phase form, arms `[Biquad<Q32<F>>; n]` x `[DirectForm1<i32>; n]`
//   idsp_lockin_i32_lo_process          external LO, arms `[Lowpass<N>; K]`
//   idsp_lockin_i32_biquad_lo_process   external LO, biquad arms (i32)
//   idsp_lockin_f32_biquad_lo_process   external LO, `[Biquad<f32>; n]` arms: the mix -> lowpass graph of examples/ddc_lockin.rs
//
// External LO: the stream kernel carries the LO samples (8 bytes, like the 8-byte output); the 4-byte x sample of the
// same (frame, lane) is read by the processor itself in its pre-stage (four frames ahead of the recurrence), through a
// pointer that walks the lane's x samples.  The kernels split lanes only for 4-byte-in / 4-byte-out processors, so the
// lane index a processor sees here is the caller's.
#include "biquad_sections.h"
#include "dds_dev.h"

namespace idsp {

// lockin_waves_biquad.hip: the multi-wave lock-in kernel with the biquad chain as its arm functor
int lockin_waves_biquad_iq(const idsp_biquad_i32 *sec, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                           int layout, hipStream_t s, size_t pitch = 0);
int lockin_waves_biquad_lo(const idsp_biquad_i32 *sec, size_t n, void *state, const int32_t *x, const int32_t *lo, int32_t *y, size_t lanes,
                           size_t frames, int layout, hipStream_t s, size_t pitch = 0);
int lockin_waves_lowpass_lo(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, const int32_t *lo, int32_t *y, size_t lanes, size_t frames,
                            int layout, hipStream_t s, size_t pitch = 0);  // lockin_waves_lo.hip
int lockin_waves_biquad_lo_f32(const idsp_biquad_f32 *sec, size_t n, void *state, const float *x, const float *lo, float *y, size_t lanes,
                               size_t frames, int layout, hipStream_t s, size_t pitch = 0);

namespace {

typedef int32_t cplx_i32 __attribute__((ext_vector_type(2)));
typedef float cplx_f32 __attribute__((ext_vector_type(2)));

// Shapes the multi-wave kernel takes: FrameMajor always, LaneMajor for whole 16-frame batches on 16-byte aligned rows (as for
// the lowpass arms, dds.hip lockin_waves_for).  IDSP_DIAG=1 IDSP_LOCKIN_NO_WAVES=1: the one-thread-per-lane kernels of this file.
inline bool waves_take(const void *x, const void *y, size_t lanes, size_t frames, int layout)
{
    static const bool no_waves = diag_env("IDSP_LOCKIN_NO_WAVES") != nullptr;
    const bool lm_ok = frames % 16 == 0 && (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 == 0;
    return !no_waves && lanes < (size_t(1) << 28) && (layout == IDSP_FRAME_MAJOR || lm_ok);
}

// LaneMajor rows of 32 + 4 k frames that are not whole batches (round 4): the frames of the part the multi-wave kernel takes at the call's row pitch
// (the last frames % 16 follow on a stream kernel, same stream), 0 = the call is not of that kind
inline size_t lm_body(const void *x, const void *y, size_t lanes, size_t frames, int layout)
{
    if (layout != IDSP_LANE_MAJOR || frames < 32 || frames % 4 != 0 || frames % 16 == 0) return 0;
    return waves_take(x, y, lanes, frames - frames % 16, layout) ? frames - frames % 16 : 0;
}

// n serial sections on one arm, state words at `word0` (section-major, {x0,x1,y0,y1} each)
template <class Sec, int NS>
struct Arm {
    uint32_t s[NS][Sec::W];
    __device__ __forceinline__ void load(const uint32_t *st, size_t lanes, size_t lane, int word0)
    {
#pragma unroll
        for (int k = 0; k < NS; k++)
#pragma unroll
            for (int w = 0; w < Sec::W; w++) s[k][w] = st[size_t(word0 + k * Sec::W + w) * lanes + lane];
    }
    __device__ __forceinline__ void store(uint32_t *st, size_t lanes, size_t lane, int word0) const
    {
#pragma unroll
        for (int k = 0; k < NS; k++)
#pragma unroll
            for (int w = 0; w < Sec::W; w++) st[size_t(word0 + k * Sec::W + w) * lanes + lane] = s[k][w];
    }
    template <class P>
    __device__ __forceinline__ typename Sec::T step(const P &p, typename Sec::T x)
    {
#pragma unroll
        for (int k = 0; k < NS; k++) x = Sec::step(p.sec[k], s[k], x);
        return x;
    }
};

// phase form, biquad arms (src/lockin.rs:30-39 -> :17-27)
template <int NS>
struct LockinBiquadProc {
    using In = int32_t;
    using Out = Cplx;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 1 << kCossinDepth;
    static constexpr int IN_DIV = 1;
    static constexpr int COST = 110 + 100 * NS;
    using Params = bq::ChainParams<bq::SecI32, NS>;
    const uint32_t *lut;
    uint32_t acc, inc;
    Arm<bq::Df1I32<false>, NS> bi, bqarm;
    static __device__ __forceinline__ void fill_shared(uint32_t *sh, int tid, int n) { fill_cossin(sh, tid, n); }
    __device__ __forceinline__ void set_shared(const uint32_t *sh) { lut = sh; }
    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t lanes, size_t lane)
    {
        acc = st[lane];
        inc = st[lanes + lane];
        bi.load(st, lanes, lane, 2);
        bqarm.load(st, lanes, lane, 2 + 4 * NS);
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane)
    {
        st[lane] = acc;
        bi.store(st, lanes, lane, 2);
        bqarm.store(st, lanes, lane, 2 + 4 * NS);
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
        return Cplx{bi.step(p, __mulhi(lo.re, x)), bqarm.step(p, __mulhi(lo.im, x))};
    }
};

// where the x samples of the external-LO forms are, beside the kernel's own (LO) input stream
struct XWalk {
    const void *x;
    uint32_t frame_major;
    size_t frames;
};

// external LO, lowpass arms
struct LpLoParams {
    LpParams lp;
    XWalk xw;
};
template <int N, int K>
struct LockinLoProc {
    using In = cplx_i32;
    using Out = Cplx;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 0;
    static constexpr int IN_DIV = 1;
    static constexpr int COST = 20 + 80 * N * K;
    using Params = LpLoParams;
    const int32_t *xp;
    size_t xstride;
    LpBank<N, K> bi, bq_;
    __device__ __forceinline__ void load(const Params &p, const uint32_t *st, size_t lanes, size_t lane)
    {
        xp = static_cast<const int32_t *>(p.xw.x) + (p.xw.frame_major ? lane : lane * p.xw.frames);
        xstride = p.xw.frame_major ? lanes : 1;
        bi.load(st, lanes, lane, 0);
        bq_.load(st, lanes, lane, 2 * N * K);
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane)
    {
        bi.store(st, lanes, lane, 0);
        bq_.store(st, lanes, lane, 2 * N * K);
    }
    static constexpr int BATCH = 4;
    using Pre = int32_t;  // the x sample of the frame
    __device__ __forceinline__ Pre pre(const Params &)
    {
        const int32_t v = *xp;
        xp += xstride;
        return v;
    }
    __device__ __forceinline__ Out step(const Params &p, In lo, const Pre &x)
    {
        return Cplx{bi.step(p.lp, __mulhi(lo.x, x)), bq_.step(p.lp, __mulhi(lo.y, x))};
    }
};

// external LO, biquad arms; T = int32_t (Q32<32> LO) or float
template <class Sec, int NS>
struct BqLoParams {
    typename Sec::Sec sec[NS];
    XWalk xw;
};
template <class Sec, int NS>
struct LockinBiquadLoProc {
    using T = typename Sec::T;
    typedef T In __attribute__((ext_vector_type(2)));
    typedef T Out __attribute__((ext_vector_type(2)));
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 0;
    static constexpr int IN_DIV = 1;
    static constexpr int COST = 20 + 2 * NS * Sec::COST;
    // 2 x NS sections of state beside a deep register window or the staged kernel's staging registers spill (tools/check_scratch.py)
    static constexpr int MAX_U = NS >= 3 ? 8 : 24;
    static constexpr bool LM_STAGED = NS <= 2;
    using Params = BqLoParams<Sec, NS>;
    const T *xp;
    size_t xstride;
    Arm<Sec, NS> bi, bq_;
    __device__ __forceinline__ void load(const Params &p, const uint32_t *st, size_t lanes, size_t lane)
    {
        xp = static_cast<const T *>(p.xw.x) + (p.xw.frame_major ? lane : lane * p.xw.frames);
        xstride = p.xw.frame_major ? lanes : 1;
        bi.load(st, lanes, lane, 0);
        bq_.load(st, lanes, lane, 4 * NS);
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane)
    {
        bi.store(st, lanes, lane, 0);
        bq_.store(st, lanes, lane, 4 * NS);
    }
    static constexpr int BATCH = 4;
    using Pre = T;
    __device__ __forceinline__ Pre pre(const Params &)
    {
        const T v = *xp;
        xp += xstride;
        return v;
    }
    static __device__ __forceinline__ int32_t mix(int32_t x, int32_t lo) { return __mulhi(lo, x); }
    static __device__ __forceinline__ float mix(float x, float lo) { return x * lo; }
    __device__ __forceinline__ Out step(const Params &p, In lo, const Pre &x)
    {
        return Out{bi.step(p, mix(x, lo.x)), bq_.step(p, mix(x, lo.y))};
    }
};

bool sections_ok(const void *sections, size_t n) { return sections && n >= 1 && n <= IDSP_LOCKIN_MAX_SECTIONS; }

template <int NS>
void fill_i32(const idsp_biquad_i32 *sec, bq::SecI32 (&out)[NS])
{
    for (int k = 0; k < NS; k++) {
        for (int i = 0; i < 5; i++) out[k].ba[i] = sec[k].ba[i];
        out[k].frac = sec[k].frac;
        out[k].u = 0, out[k].mn = INT32_MIN, out[k].mx = INT32_MAX;
    }
}
template <int NS>
void fill_f32(const idsp_biquad_f32 *sec, bq::SecF32 (&out)[NS])
{
    for (int k = 0; k < NS; k++) {
        for (int i = 0; i < 5; i++) out[k].ba[i] = sec[k].ba[i];
        out[k].u = 0.f, out[k].mn = -__builtin_inff(), out[k].mx = __builtin_inff();
    }
}

int check_lo_args(const void *cfg, const void *state, const void *x, const void *lo, const void *y, size_t lanes, size_t frames, int layout)
{
    int rc = check_stream_args(cfg, 1, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if (lanes && frames && !lo) return fail(IDSP_EINVAL, "lo is NULL");
    // lo and y hold `Complex` pairs and are read / written 8 bytes at a time (i32x2 in lockin_waves.h, cplx_i32 / cplx_f32 in the stream
    // processors): an address that is only 4-byte aligned is legal by the C signature but would rely on the unaligned-access mode
    if (reinterpret_cast<uintptr_t>(lo) % 8 || reinterpret_cast<uintptr_t>(y) % 8)
        return fail(IDSP_EINVAL, "lo and y hold Complex pairs: both must be 8-byte aligned");
    // three separate buffers (include/idsp_hip.h): x is walked by a side pointer while lo streams through the kernel, no in-place form
    const uintptr_t n = uintptr_t(lanes) * frames, xb = reinterpret_cast<uintptr_t>(x), lb = reinterpret_cast<uintptr_t>(lo),
                    yb = reinterpret_cast<uintptr_t>(y);
    auto meet = [](uintptr_t a, uintptr_t an, uintptr_t b, uintptr_t bn) { return a < b + bn && b < a + an; };
    if (n && (meet(xb, n * 4, yb, n * 8) || meet(lb, n * 8, yb, n * 8) || meet(xb, n * 4, lb, n * 8)))
        return fail(IDSP_EINVAL, "x, lo and y must not overlap");
    return IDSP_OK;
}

template <int NS>
int run_biquad_phase(const idsp_biquad_i32 *sec, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, hipStream_t s,
                     size_t pitch = 0)
{
    typename LockinBiquadProc<NS>::Params p;
    fill_i32<NS>(sec, p.sec);
    return launch_stream<LockinBiquadProc<NS>>(p, state, x, reinterpret_cast<Cplx *>(y), lanes, frames, layout, s, Pitch{pitch, pitch});
}
template <int NS>
int run_biquad_lo_i32(const idsp_biquad_i32 *sec, void *state, const int32_t *x, const int32_t *lo, int32_t *y, size_t lanes, size_t frames,
                      int layout, hipStream_t s, size_t pitch = 0)
{
    using P = LockinBiquadLoProc<bq::Df1I32<false>, NS>;
    typename P::Params p;
    fill_i32<NS>(sec, p.sec);
    p.xw = XWalk{x, layout == IDSP_FRAME_MAJOR ? 1u : 0u, pitch ? pitch : frames};
    return launch_stream<P>(p, state, reinterpret_cast<const typename P::In *>(lo), reinterpret_cast<typename P::Out *>(y), lanes, frames, layout, s,
                            Pitch{pitch, pitch});
}
template <int NS>
int run_biquad_lo_f32(const idsp_biquad_f32 *sec, void *state, const float *x, const float *lo, float *y, size_t lanes, size_t frames, int layout,
                      hipStream_t s, size_t pitch = 0)
{
    using P = LockinBiquadLoProc<bq::Df1F32<false>, NS>;
    typename P::Params p;
    fill_f32<NS>(sec, p.sec);
    p.xw = XWalk{x, layout == IDSP_FRAME_MAJOR ? 1u : 0u, pitch ? pitch : frames};
    return launch_stream<P>(p, state, reinterpret_cast<const typename P::In *>(lo), reinterpret_cast<typename P::Out *>(y), lanes, frames, layout, s, Pitch{pitch, pitch});
}

}  // namespace
}  // namespace idsp

using namespace idsp;

extern "C" {

size_t idsp_lockin_biquad_state_words(size_t n, int with_accu)
{
    return n >= 1 && n <= IDSP_LOCKIN_MAX_SECTIONS ? size_t(with_accu ? 2 : 0) + 8 * n : 0;
}

#define IDSP_BY_SECTIONS(call)                                     \
    switch (n) {                                                   \
        case 1: return call(1);                                    \
        case 2: return call(2);                                    \
        case 3: return call(3);                                    \
        default: return call(4);                                   \
    }

int idsp_lockin_i32_biquad_process(const idsp_biquad_i32 *sections, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes,
                                   size_t frames, int layout, void *stream)
{
    if (!sections_ok(sections, n)) return fail(IDSP_EINVAL, "sections is NULL or n = %zu not in 1..%d", n, IDSP_LOCKIN_MAX_SECTIONS);
    int rc = check_stream_args(sections, 1, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    for (size_t k = 0; k < n; k++)
        if (sections[k].frac < 0 || sections[k].frac > 31) return fail(IDSP_EINVAL, "section %zu: frac = %d not in 0..31", k, sections[k].frac);
    if (lanes == 0 || frames == 0) return IDSP_OK;
    if (waves_take(x, y, lanes, frames, layout)) return lockin_waves_biquad_iq(sections, n, state, x, y, lanes, frames, layout, as_stream(stream));
    // LaneMajor rows of 32 + 4 k frames that are not whole batches: the whole batches of every row on the multi-wave kernel at the call's row
    // pitch, the last frames % 16 on the stream kernel behind it (as for the lowpass arms, dds.hip lockin_lm_body)
    if (const size_t body = lm_body(x, y, lanes, frames, layout)) {
        if ((rc = lockin_waves_biquad_iq(sections, n, state, x, y, lanes, body, layout, as_stream(stream), frames))) return rc;
#define IDSP_CALL(NS) run_biquad_phase<NS>(sections, state, x + body, y + 2 * body, lanes, frames - body, layout, as_stream(stream), frames)
        rc = [&]() -> int { IDSP_BY_SECTIONS(IDSP_CALL) }();
#undef IDSP_CALL
        if (rc == IDSP_OK) note_kernel("lockin_waves_kernel + stream kernel (last frames % 16)", "[Biquad; n]");
        return rc;
    }
#define IDSP_CALL(NS) run_biquad_phase<NS>(sections, state, x, y, lanes, frames, layout, as_stream(stream))
    IDSP_BY_SECTIONS(IDSP_CALL)
#undef IDSP_CALL
}

int idsp_lockin_i32_lo_process(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, const int32_t *lo, int32_t *y, size_t lanes,
                               size_t frames, int layout, void *stream)
{
    if (!cfg || (cfg->order != 1 && cfg->order != 2) || cfg->cascade < 1 || cfg->cascade > IDSP_LOCKIN_MAX_CASCADE)
        return fail(IDSP_EINVAL, "lock-in configuration: order 1..2, cascade 1..%d", IDSP_LOCKIN_MAX_CASCADE);
    int rc = check_lo_args(cfg, state, x, lo, y, lanes, frames, layout);
    if (rc) return rc;
    if (lanes == 0 || frames == 0) return IDSP_OK;
    if (waves_take(x, y, lanes, frames, layout)) return lockin_waves_lowpass_lo(cfg, state, x, lo, y, lanes, frames, layout, as_stream(stream));
    size_t pitch = 0;
    if (const size_t body = lm_body(x, y, lanes, frames, layout)) {
        if ((rc = lockin_waves_lowpass_lo(cfg, state, x, lo, y, lanes, body, layout, as_stream(stream), frames))) return rc;
        pitch = frames, x += body, lo += 2 * body, y += 2 * body, frames -= body;
    }
    LpLoParams p;
    p.lp = lp_params(cfg);
    p.xw = XWalk{x, layout == IDSP_FRAME_MAJOR ? 1u : 0u, pitch ? pitch : frames};
    const cplx_i32 *l2 = reinterpret_cast<const cplx_i32 *>(lo);
    Cplx *y2 = reinterpret_cast<Cplx *>(y);
#define IDSP_CASE(N, K) \
    if (cfg->order == N && cfg->cascade == K) return launch_stream<LockinLoProc<N, K>>(p, state, l2, y2, lanes, frames, layout, as_stream(stream), Pitch{pitch, pitch})
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

int idsp_lockin_i32_biquad_lo_process(const idsp_biquad_i32 *sections, size_t n, void *state, const int32_t *x, const int32_t *lo, int32_t *y,
                                      size_t lanes, size_t frames, int layout, void *stream)
{
    if (!sections_ok(sections, n)) return fail(IDSP_EINVAL, "sections is NULL or n = %zu not in 1..%d", n, IDSP_LOCKIN_MAX_SECTIONS);
    int rc = check_lo_args(sections, state, x, lo, y, lanes, frames, layout);
    if (rc) return rc;
    for (size_t k = 0; k < n; k++)
        if (sections[k].frac < 0 || sections[k].frac > 31) return fail(IDSP_EINVAL, "section %zu: frac = %d not in 0..31", k, sections[k].frac);
    if (lanes == 0 || frames == 0) return IDSP_OK;
    if (waves_take(x, y, lanes, frames, layout)) return lockin_waves_biquad_lo(sections, n, state, x, lo, y, lanes, frames, layout, as_stream(stream));
    size_t pitch = 0;
    if (const size_t body = lm_body(x, y, lanes, frames, layout)) {
        if ((rc = lockin_waves_biquad_lo(sections, n, state, x, lo, y, lanes, body, layout, as_stream(stream), frames))) return rc;
        pitch = frames, x += body, lo += 2 * body, y += 2 * body, frames -= body;
    }
#define IDSP_CALL(NS) run_biquad_lo_i32<NS>(sections, state, x, lo, y, lanes, frames, layout, as_stream(stream), pitch)
    IDSP_BY_SECTIONS(IDSP_CALL)
#undef IDSP_CALL
}

int idsp_lockin_f32_biquad_lo_process(const idsp_biquad_f32 *sections, size_t n, void *state, const float *x, const float *lo, float *y,
                                      size_t lanes, size_t frames, int layout, void *stream)
{
    if (!sections_ok(sections, n)) return fail(IDSP_EINVAL, "sections is NULL or n = %zu not in 1..%d", n, IDSP_LOCKIN_MAX_SECTIONS);
    int rc = check_lo_args(sections, state, x, lo, y, lanes, frames, layout);
    if (rc) return rc;
    if (lanes == 0 || frames == 0) return IDSP_OK;
    if (waves_take(x, y, lanes, frames, layout)) return lockin_waves_biquad_lo_f32(sections, n, state, x, lo, y, lanes, frames, layout, as_stream(stream));
    size_t pitch = 0;
    if (const size_t body = lm_body(x, y, lanes, frames, layout)) {
        if ((rc = lockin_waves_biquad_lo_f32(sections, n, state, x, lo, y, lanes, body, layout, as_stream(stream), frames))) return rc;
        pitch = frames, x += body, lo += 2 * body, y += 2 * body, frames -= body;
    }
#define IDSP_CALL(NS) run_biquad_lo_f32<NS>(sections, state, x, lo, y, lanes, frames, layout, as_stream(stream), pitch)
    IDSP_BY_SECTIONS(IDSP_CALL)
#undef IDSP_CALL
}

}  // extern "C"
