// cic.hip — `Cic<T, N, M>` decimator / interpolator over many lanes (reference: src/cic.rs),
// T = i32 or i64, wrapping arithmetic, bit-exact.
//
// One lane per thread: the N integrators are a serial recurrence at the high rate, so the
// whole state (zoh, combs, integrators) lives in registers for the call.  A thread's chunk
// `[T; R]` is contiguous in both layouts; it is moved with 16-byte accesses when R allows.
// Roofline: HBM, (R + 1) * sizeof(T) bytes per low-rate frame.
#include "common.h"

namespace idsp {
namespace {

constexpr int kBlock = 256;

template <class T>
struct Vec16 {
    typedef T type __attribute__((ext_vector_type(16 / sizeof(T))));
    static constexpr int n = 16 / sizeof(T);
};

template <class T>
__device__ __forceinline__ T wadd(T a, T b)
{
    using U = typename std::make_unsigned<T>::type;
    return T(U(a) + U(b));
}
template <class T>
__device__ __forceinline__ T wsub(T a, T b)
{
    using U = typename std::make_unsigned<T>::type;
    return T(U(a) - U(b));
}

template <class T, int N>
struct CicRegs {
    static constexpr int VW = sizeof(T) / 4;
    T zoh, comb[N][IDSP_CIC_MAX_DELAY], integ[N];

    __device__ __forceinline__ T ldv(const uint32_t *st, size_t lanes, size_t lane, int v) const
    {
        if constexpr (VW == 1) {
            return T(st[size_t(v) * lanes + lane]);
        } else {
            return T(uint64_t(st[size_t(2 * v) * lanes + lane]) | (uint64_t(st[size_t(2 * v + 1) * lanes + lane]) << 32));
        }
    }
    __device__ __forceinline__ void stv(uint32_t *st, size_t lanes, size_t lane, int v, T x) const
    {
        if constexpr (VW == 1) {
            st[size_t(v) * lanes + lane] = uint32_t(x);
        } else {
            st[size_t(2 * v) * lanes + lane] = uint32_t(uint64_t(x));
            st[size_t(2 * v + 1) * lanes + lane] = uint32_t(uint64_t(x) >> 32);
        }
    }
    __device__ __forceinline__ void load(const uint32_t *st, size_t lanes, size_t lane, int m)
    {
        zoh = ldv(st, lanes, lane, 0);
#pragma unroll
        for (int n = 0; n < N; n++)
#pragma unroll
            for (int j = 0; j < IDSP_CIC_MAX_DELAY; j++) comb[n][j] = j < m ? ldv(st, lanes, lane, 1 + n * m + j) : T(0);
#pragma unroll
        for (int n = 0; n < N; n++) integ[n] = ldv(st, lanes, lane, 1 + N * m + n);
    }
    __device__ __forceinline__ void store(uint32_t *st, size_t lanes, size_t lane, int m) const
    {
        stv(st, lanes, lane, 0, zoh);
#pragma unroll
        for (int n = 0; n < N; n++)
#pragma unroll
            for (int j = 0; j < IDSP_CIC_MAX_DELAY; j++)
                if (j < m) stv(st, lanes, lane, 1 + n * m + j, comb[n][j]);
#pragma unroll
        for (int n = 0; n < N; n++) stv(st, lanes, lane, 1 + N * m + n, integ[n]);
    }
    // src/cic.rs:166-171,197-203: y = x - c[0]; c.copy_within(1.., 0); c[M-1] = x
    __device__ __forceinline__ T combs(T x, int m)
    {
#pragma unroll
        for (int n = 0; n < N; n++) {
            const T y = wsub(x, comb[n][0]);
#pragma unroll
            for (int j = 0; j < IDSP_CIC_MAX_DELAY - 1; j++)
                if (j + 1 < m) comb[n][j] = comb[n][j + 1];
#pragma unroll
            for (int j = 0; j < IDSP_CIC_MAX_DELAY; j++)
                if (j == m - 1) comb[n][j] = x;
            x = y;
        }
        return x;
    }
    // src/cic.rs:174-180,189-193
    __device__ __forceinline__ T integrate(T x)
    {
#pragma unroll
        for (int n = 0; n < N; n++) {
            integ[n] = wadd(integ[n], x);
            x = integ[n];
        }
        return x;
    }
};

// Decimator: per chunk, sample 0 integrates, ticks (index 0 -> rate) and runs the combs; samples
// 1..R-1 only integrate (src/cic.rs:186-207 driven by adapters.rs:158-167).
template <class T, int N>
__global__ __launch_bounds__(kBlock) void cic_dec_kernel(const idsp_cic cfg, uint32_t *st, const T *x, T *y, const size_t lanes,
                                                          const size_t frames, const int layout)
{
    const size_t lane = size_t(blockIdx.x) * kBlock + threadIdx.x;
    if (lane >= lanes) return;
    const size_t R = size_t(cfg.rate) + 1;
    const int m = cfg.comb_delay;
    CicRegs<T, N> c;
    c.load(st, lanes, lane, m);
    const bool fm = layout == IDSP_FRAME_MAJOR;
    const T *hp = x + (fm ? lane * R : lane * frames * R);
    const size_t hstride = fm ? lanes * R : R;
    T *lp = y + (fm ? lane : lane * frames);
    const size_t lstride = fm ? lanes : 1;
    using V = Vec16<T>;
    const bool vec = R % V::n == 0 && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
    for (size_t f = 0; f < frames; f++) {
        const T *row = hp + f * hstride;
        T out;
        if (vec) {
            const typename V::type *vr = reinterpret_cast<const typename V::type *>(row);
            typename V::type v = vr[0];
            out = c.integrate(v[0]);
            c.zoh = c.combs(out, m);
#pragma unroll
            for (int k = 1; k < V::n; k++) c.integrate(v[k]);
            const size_t nv = R / V::n;
#pragma unroll 4
            for (size_t i = 1; i < nv; i++) {
                v = vr[i];
#pragma unroll
                for (int k = 0; k < V::n; k++) c.integrate(v[k]);
            }
        } else {
            out = c.integrate(row[0]);
            c.zoh = c.combs(out, m);
            for (size_t r = 1; r < R; r++) c.integrate(row[r]);
        }
        lp[f * lstride] = c.zoh;
    }
    c.store(st, lanes, lane, m);
}

// Interpolator: per input sample the combs run once (index = rate), then R integrator passes over
// the held comb output emit the chunk (src/cic.rs:160-182 driven by adapters.rs:27-35).
template <class T, int N>
__global__ __launch_bounds__(kBlock) void cic_int_kernel(const idsp_cic cfg, uint32_t *st, const T *x, T *y, const size_t lanes,
                                                          const size_t frames, const int layout)
{
    const size_t lane = size_t(blockIdx.x) * kBlock + threadIdx.x;
    if (lane >= lanes) return;
    const size_t R = size_t(cfg.rate) + 1;
    const int m = cfg.comb_delay;
    CicRegs<T, N> c;
    c.load(st, lanes, lane, m);
    const bool fm = layout == IDSP_FRAME_MAJOR;
    T *hp = y + (fm ? lane * R : lane * frames * R);
    const size_t hstride = fm ? lanes * R : R;
    const T *lp = x + (fm ? lane : lane * frames);
    const size_t lstride = fm ? lanes : 1;
    using V = Vec16<T>;
    const bool vec = R % V::n == 0 && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
    for (size_t f = 0; f < frames; f++) {
        c.zoh = c.combs(lp[f * lstride], m);
        T *row = hp + f * hstride;
        if (vec) {
            typename V::type *vr = reinterpret_cast<typename V::type *>(row);
            const size_t nv = R / V::n;
#pragma unroll 4
            for (size_t i = 0; i < nv; i++) {
                typename V::type v;
#pragma unroll
                for (int k = 0; k < V::n; k++) v[k] = c.integrate(c.zoh);
                vr[i] = v;
            }
        } else {
            for (size_t r = 0; r < R; r++) row[r] = c.integrate(c.zoh);
        }
    }
    c.store(st, lanes, lane, m);
}

int cfg_check(const idsp_cic *c)
{
    if (!c) return fail(IDSP_EINVAL, "cfg is NULL");
    if (c->order < 1 || c->order > IDSP_CIC_MAX_ORDER) return fail(IDSP_EINVAL, "Cic order N = %d not in 1..%d", c->order, IDSP_CIC_MAX_ORDER);
    if (c->comb_delay < 1 || c->comb_delay > IDSP_CIC_MAX_DELAY)
        return fail(IDSP_EINVAL, "Cic comb delay M = %d not in 1..%d (src/cic.rs:36: must be non-zero)", c->comb_delay, IDSP_CIC_MAX_DELAY);
    return IDSP_OK;
}

template <class T, bool DEC>
int run(const idsp_cic *cfg, void *state, const T *x, T *y, size_t lanes, size_t frames, int layout, void *stream)
{
    int rc = cfg_check(cfg);
    if (rc) return rc;
    if ((rc = check_stream_args(cfg, 1, state, x, y, lanes, frames, layout))) return rc;
    if (lanes == 0 || frames == 0) return IDSP_OK;
    const dim3 grid(unsigned((lanes + kBlock - 1) / kBlock)), block(kBlock);
    uint32_t *st = static_cast<uint32_t *>(state);
    hipStream_t s = as_stream(stream);
#define IDSP_CIC_CASE(NN)                                                                                       \
    case NN:                                                                                                    \
        if (DEC)                                                                                                \
            hipLaunchKernelGGL((cic_dec_kernel<T, NN>), grid, block, 0, s, *cfg, st, x, y, lanes, frames, layout); \
        else                                                                                                    \
            hipLaunchKernelGGL((cic_int_kernel<T, NN>), grid, block, 0, s, *cfg, st, x, y, lanes, frames, layout); \
        break;
    switch (cfg->order) {
        IDSP_CIC_CASE(1)
        IDSP_CIC_CASE(2)
        IDSP_CIC_CASE(3)
        IDSP_CIC_CASE(4)
        IDSP_CIC_CASE(5)
        IDSP_CIC_CASE(6)
    }
#undef IDSP_CIC_CASE
    return launch_status();
}

}  // namespace
}  // namespace idsp

using namespace idsp;

extern "C" {

int64_t idsp_cic_gain(const idsp_cic *cfg)
{
    if (cfg_check(cfg)) return 0;
    // (M * (rate + 1)).pow(N) in i64, wrapping (src/cic.rs:103-105)
    const uint64_t b = uint64_t(cfg->comb_delay) * (uint64_t(cfg->rate) + 1);
    uint64_t g = 1;
    for (int i = 0; i < cfg->order; i++) g *= b;
    return int64_t(g);
}

int idsp_cic_gain_log2(const idsp_cic *cfg)
{
    const int rc = cfg_check(cfg);
    if (rc) return rc;
    // (u32::BITS - (M * rate + (M - 1)).leading_zeros()) * N  (src/cic.rs:111-113; u32 arithmetic)
    const uint32_t v = uint32_t(cfg->comb_delay) * cfg->rate + uint32_t(cfg->comb_delay - 1);
    const int bits = v ? 32 - __builtin_clz(v) : 0;
    return bits * cfg->order;
}

size_t idsp_cic_response_length(const idsp_cic *cfg)
{
    if (cfg_check(cfg)) return 0;
    return size_t(cfg->rate) * size_t(cfg->order);  // src/cic.rs:116-118
}

size_t idsp_cic_state_words(const idsp_cic *cfg, int bits)
{
    if (cfg_check(cfg) || (bits != 32 && bits != 64)) return 0;
    return size_t(1 + cfg->order * cfg->comb_delay + cfg->order) * size_t(bits / 32);
}

int idsp_cic_dec_i32(const idsp_cic *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout,
                     void *stream)
{
    return run<int32_t, true>(cfg, state, x, y, lanes, frames, layout, stream);
}
int idsp_cic_dec_i64(const idsp_cic *cfg, void *state, const int64_t *x, int64_t *y, size_t lanes, size_t frames, int layout,
                     void *stream)
{
    return run<int64_t, true>(cfg, state, x, y, lanes, frames, layout, stream);
}
int idsp_cic_int_i32(const idsp_cic *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout,
                     void *stream)
{
    return run<int32_t, false>(cfg, state, x, y, lanes, frames, layout, stream);
}
int idsp_cic_int_i64(const idsp_cic *cfg, void *state, const int64_t *x, int64_t *y, size_t lanes, size_t frames, int layout,
                     void *stream)
{
    return run<int64_t, false>(cfg, state, x, y, lanes, frames, layout, stream);
}

}  // extern "C"
