// cic_kernels.h — `Cic<T, N, M>` decimator / interpolator over many lanes (reference: src/cic.rs),
// T = i32 or i64, wrapping arithmetic, bit-exact.  Included by one translation unit per (direction, T)
// so that the 72 kernel instantiations of each compile in parallel.
//
// One lane per thread: the N integrators are a serial recurrence at the high rate, so the
// whole state (zoh, combs, integrators) lives in registers for the call.  A thread's chunk
// `[T; R]` is contiguous in both layouts; it is moved with 16-byte accesses when R allows.
// Roofline: HBM, (R + 1) * sizeof(T) bytes per low-rate frame.
#pragma once

#include "common.h"
#include "cic_ring.h"

namespace idsp {
namespace cic {


// one wave per workgroup: at 16384 lanes that is 256 workgroups = every CU gets one (256-thread
// blocks would fill only 64 CUs, and a CU's L1 moves ~10 B/cycle: 1.3 TB/s measured that way)
constexpr int kBlock = 64;

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

// Chunk widths with a dedicated instantiation: VPC 16-byte vectors per chunk `[T; R]` (VPC = 0: any
// other width, scalar loop).  At 16384 lanes there are only 256 waves on the chip, so every wave has to
// keep tens of KiB in flight by itself (tools/ubench_pattern.hip: 2.0 / 3.9 / 5.0 TB/s with 1 / 4 / 16
// chunk rows in flight per thread at 256 waves): the chunk stream runs through a register ring of
// kRingVecs vectors.  Plain loads — a 128-byte line is shared by two lanes and touched by up to four
// instructions, and nontemporal loads re-fetched it every time (measured: 4x traffic).
constexpr int kRingVecs = 64;  // at most 16 frames

// Decimator: per chunk, sample 0 integrates, ticks (index 0 -> rate) and runs the combs; samples
// 1..R-1 only integrate (src/cic.rs:186-207 driven by adapters.rs:158-167).
template <class T, int N, int VPC>
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
    if constexpr (VPC > 0) {
        using V = Vec16<T>;
        using VT = typename V::type;
        constexpr int U = kRingVecs / VPC < 2 ? 2 : (kRingVecs / VPC > 16 ? 16 : kRingVecs / VPC);
        VT ring[U][VPC];
        auto fetch = [&](int u, size_t f) {
            const VT *vr = reinterpret_cast<const VT *>(hp + f * hstride);
#pragma unroll
            for (int i = 0; i < VPC; i++) ring[u][i] = vr[i];
        };
        // integrates the chunk in ring slot u; the combs take the integrator output of sample 0 and touch
        // only comb state, so they run after the whole chunk without changing any value
        auto frame = [&](int u, size_t f) {
            T out{};
#pragma unroll
            for (int i = 0; i < VPC; i++)
#pragma unroll
                for (int k = 0; k < V::n; k++) {
                    const T v = c.integrate(ring[u][i][k]);
                    if (i == 0 && k == 0) out = v;
                }
            c.zoh = c.combs(out, m);
            lp[f * lstride] = c.zoh;
        };
#pragma unroll
        for (int u = 0; u < U; u++)
            if (size_t(u) < frames) fetch(u, size_t(u));
        size_t f = 0;
        for (; f + 2 * U <= frames; f += U) {  // every refill of this trip is in range
#pragma unroll
            for (int u = 0; u < U; u++) {
                frame(u, f + u);
                fetch(u, f + u + U);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (f + u < frames) {
                frame(u, f + u);
                if (f + u + U < frames) fetch(u, f + u + U);
            }
        }
        f += U;
#pragma unroll
        for (int u = 0; u < U; u++)
            if (f + u < frames) frame(u, f + u);
    } else {
        for (size_t f = 0; f < frames; f++) {
            const T *row = hp + f * hstride;
            const T out = c.integrate(row[0]);
            c.zoh = c.combs(out, m);
            for (size_t r = 1; r < R; r++) c.integrate(row[r]);
            lp[f * lstride] = c.zoh;
        }
    }
    c.store(st, lanes, lane, m);
}

// Interpolator: per input sample the combs run once (index = rate), then R integrator passes over
// the held comb output emit the chunk (src/cic.rs:160-182 driven by adapters.rs:27-35).  The
// low-rate inputs are read kInAhead frames ahead so that waiting for one does not drain the chunk
// stores issued since (vmcnt is in-order).
constexpr int kInAhead = 16;

template <class T, int N, int VPC>
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
    // FRAME_MAJOR, whole wave: the 64 chunks of a frame are one contiguous run of 64 * VPC vectors.  Writing
    // each thread's own chunk directly means 16-byte pieces at a VPC*16-byte lane stride (<= 3.7 TB/s,
    // tools/ubench_pattern.hip); instead the wave transposes the frame through a padded LDS tile and every
    // store instruction writes 1 KiB of whole lines (6.6 TB/s for that shape).
    using VT = typename Vec16<T>::type;
    __shared__ VT tile[VPC > 0 ? kBlock * (VPC + 1) : 1];
    const bool transposed = VPC > 0 && fm && (size_t(blockIdx.x) + 1) * kBlock <= lanes;
    const int lid = threadIdx.x;
    auto chunk = [&](size_t f, T xin) {
        c.zoh = c.combs(xin, m);
        T *row = hp + f * hstride;
        if constexpr (VPC > 0) {
            using V = Vec16<T>;
            typename V::type *vr = reinterpret_cast<typename V::type *>(row);
            typename V::type v[VPC];
#pragma unroll
            for (int i = 0; i < VPC; i++)
#pragma unroll
                for (int k = 0; k < V::n; k++) v[i][k] = c.integrate(c.zoh);
            if (transposed) {
#pragma unroll
                for (int i = 0; i < VPC; i++) tile[lid * (VPC + 1) + i] = v[i];
                lds_wave_sync();
                VT *run = reinterpret_cast<VT *>(y + (f * lanes + size_t(blockIdx.x) * kBlock) * R);
#pragma unroll
                for (int k = 0; k < VPC; k++) {
                    const int j = k * kBlock + lid;  // vector j of the run = vector j % VPC of lane j / VPC
                    run[j] = tile[(j / VPC) * (VPC + 1) + (j % VPC)];
                }
                lds_wave_sync();
            } else {
#pragma unroll
                for (int i = 0; i < VPC; i++) vr[i] = v[i];
            }
        } else {
            for (size_t r = 0; r < R; r++) row[r] = c.integrate(c.zoh);
        }
    };
    T ring[kInAhead];
#pragma unroll
    for (int u = 0; u < kInAhead; u++)
        if (size_t(u) < frames) ring[u] = lp[size_t(u) * lstride];
    size_t f = 0;
    for (; f + 2 * kInAhead <= frames; f += kInAhead) {
#pragma unroll
        for (int u = 0; u < kInAhead; u++) {
            const T xin = ring[u];
            ring[u] = lp[(f + u + kInAhead) * lstride];
            chunk(f + u, xin);
        }
    }
#pragma unroll
    for (int u = 0; u < kInAhead; u++) {
        if (f + u < frames) {
            const T xin = ring[u];
            if (f + u + kInAhead < frames) ring[u] = lp[(f + u + kInAhead) * lstride];
            chunk(f + u, xin);
        }
    }
    f += kInAhead;
#pragma unroll
    for (int u = 0; u < kInAhead; u++)
        if (f + u < frames) chunk(f + u, ring[u]);
    c.store(st, lanes, lane, m);
}

// ------------------------------------------------------------ LANE_MAJOR tiles
// In LANE_MAJOR a lane's chunk stream is one contiguous row, rows are frames*R elements apart: per-thread
// 16-byte pieces touch 64 different lines per instruction (dec 2.8-3.1, int 3.1 TB/s at 16384 lanes).  A wave
// therefore moves tiles of 64 lanes x kTileVecs vectors (= F = kTileVecs / VPC frames): every global
// instruction covers 4 lanes x 256 contiguous bytes, and a padded LDS tile (row stride kTileVecs + 1
// vectors: conflict-free b128 on both sides) hands each thread its own row.  Whole waves only; frames
// beyond the last whole tile take the per-thread path.
constexpr int kTileVecs = 16;
constexpr int kLmTilesAhead = 4;  // dec: tiles in flight per wave (register ring, 64 KiB)

template <class T, int N, int VPC>
__global__ __launch_bounds__(kBlock) void cic_dec_lm_kernel(const idsp_cic cfg, uint32_t *st, const T *x, T *y, const size_t lanes,
                                                             const size_t frames)
{
    using V = Vec16<T>;
    using VT = typename V::type;
    constexpr int F = kTileVecs / VPC;  // frames per tile
    __shared__ VT tile[kBlock * (kTileVecs + 1)];
    const int lid = threadIdx.x;
    const size_t lane0 = size_t(blockIdx.x) * kBlock, lane = lane0 + lid;
    const size_t R = size_t(cfg.rate) + 1;
    const int m = cfg.comb_delay;
    CicRegs<T, N> c;
    c.load(st, lanes, lane, m);
    const size_t ntiles = frames / F;
    // instruction k of a tile: lanes 4k .. 4k+3, this thread moves vector lid % 16 of lane 4k + lid / 16
    const VT *src = reinterpret_cast<const VT *>(x + (lane0 + size_t(lid / kTileVecs)) * frames * R) + lid % kTileVecs;
    const size_t lane4 = 4 * frames * R / V::n;  // vectors between lane groups
    VT ring[kLmTilesAhead][kTileVecs];
    auto fetch = [&](int u, size_t t) {
#pragma unroll
        for (int k = 0; k < kTileVecs; k++) ring[u][k] = src[size_t(k) * lane4 + t * kTileVecs];
    };
    T *yrow = y + lane * frames;
    const bool vec_out = (frames * sizeof(T)) % 16 == 0;  // rows of y start 16-byte aligned (y itself is, checked by the host)
    auto process = [&](int u, size_t t) {
#pragma unroll
        for (int k = 0; k < kTileVecs; k++) tile[(4 * k + lid / kTileVecs) * (kTileVecs + 1) + lid % kTileVecs] = ring[u][k];
        lds_wave_sync();
        T out[F];
#pragma unroll
        for (int f = 0; f < F; f++) {
            T first{};
#pragma unroll
            for (int i = 0; i < VPC; i++) {
                const VT v = tile[lid * (kTileVecs + 1) + f * VPC + i];
#pragma unroll
                for (int e = 0; e < V::n; e++) {
                    const T w = c.integrate(v[e]);
                    if (i == 0 && e == 0) first = w;
                }
            }
            c.zoh = c.combs(first, m);
            out[f] = c.zoh;
        }
        lds_wave_sync();
        // the F outputs of a tile are contiguous in this lane's row: 16-byte stores when they fill vectors
        if constexpr ((F * sizeof(T)) % 16 == 0) {
            if (vec_out) {
#pragma unroll
                for (int f = 0; f < F; f += V::n) {
                    VT v;
#pragma unroll
                    for (int e = 0; e < V::n; e++) v[e] = out[f + e];
                    *reinterpret_cast<VT *>(yrow + t * F + f) = v;
                }
                return;
            }
        }
#pragma unroll
        for (int f = 0; f < F; f++) yrow[t * F + f] = out[f];
    };
#pragma unroll
    for (int u = 0; u < kLmTilesAhead; u++)
        if (size_t(u) < ntiles) fetch(u, size_t(u));
    size_t t = 0;
    for (; t + 2 * kLmTilesAhead <= ntiles; t += kLmTilesAhead) {
#pragma unroll
        for (int u = 0; u < kLmTilesAhead; u++) {
            process(u, t + u);
            fetch(u, t + u + kLmTilesAhead);
        }
    }
#pragma unroll
    for (int u = 0; u < kLmTilesAhead; u++) {
        if (t + u < ntiles) {
            process(u, t + u);
            if (t + u + kLmTilesAhead < ntiles) fetch(u, t + u + kLmTilesAhead);
        }
    }
    t += kLmTilesAhead;
#pragma unroll
    for (int u = 0; u < kLmTilesAhead; u++)
        if (t + u < ntiles) process(u, t + u);
    // frames beyond the last whole tile: per-thread pieces
    const T *hp = x + lane * frames * R;
    for (size_t f = ntiles * F; f < frames; f++) {
        const T *row = hp + f * R;
        const T first = c.integrate(row[0]);
        c.zoh = c.combs(first, m);
        for (size_t r = 1; r < R; r++) c.integrate(row[r]);
        yrow[f] = c.zoh;
    }
    c.store(st, lanes, lane, m);
}

template <class T, int N, int VPC>
__global__ __launch_bounds__(kBlock) void cic_int_lm_kernel(const idsp_cic cfg, uint32_t *st, const T *x, T *y, const size_t lanes,
                                                             const size_t frames)
{
    using V = Vec16<T>;
    using VT = typename V::type;
    constexpr int F = kTileVecs / VPC;
    __shared__ VT tile[kBlock * (kTileVecs + 1)];
    const int lid = threadIdx.x;
    const size_t lane0 = size_t(blockIdx.x) * kBlock, lane = lane0 + lid;
    const size_t R = size_t(cfg.rate) + 1;
    const int m = cfg.comb_delay;
    CicRegs<T, N> c;
    c.load(st, lanes, lane, m);
    const size_t ntiles = frames / F;
    VT *dst = reinterpret_cast<VT *>(y + (lane0 + size_t(lid / kTileVecs)) * frames * R) + lid % kTileVecs;
    const size_t lane4 = 4 * frames * R / V::n;
    const T *xrow = x + lane * frames;
    // inputs of kInTiles tiles ahead, so that waiting for one does not drain the row stores issued since
    // (64-bit samples: 128-byte tiles keep 4, and order 6 at 64-byte tiles 6, so that the ring stays in registers —
    // 8 tiles spilled 0.15-1.2 KB per thread to scratch there, tools/check_scratch.py)
    constexpr int kInTiles = sizeof(T) * F >= 128 ? (N >= 6 ? 3 : 4) : (sizeof(T) * F >= 64 && sizeof(T) == 8 && N >= 6 ? 6 : 8);
    T ring[kInTiles][F];
    auto fetch = [&](int u, size_t t) {
#pragma unroll
        for (int f = 0; f < F; f++) ring[u][f] = xrow[t * F + f];
    };
    auto process = [&](int u, size_t t) {
#pragma unroll
        for (int f = 0; f < F; f++) {
            c.zoh = c.combs(ring[u][f], m);
#pragma unroll
            for (int i = 0; i < VPC; i++) {
                VT v;
#pragma unroll
                for (int e = 0; e < V::n; e++) v[e] = c.integrate(c.zoh);
                tile[lid * (kTileVecs + 1) + f * VPC + i] = v;
            }
        }
        lds_wave_sync();
#pragma unroll
        for (int k = 0; k < kTileVecs; k++)
            dst[size_t(k) * lane4 + t * kTileVecs] = tile[(4 * k + lid / kTileVecs) * (kTileVecs + 1) + lid % kTileVecs];
        lds_wave_sync();
    };
#pragma unroll
    for (int u = 0; u < kInTiles; u++)
        if (size_t(u) < ntiles) fetch(u, size_t(u));
    size_t t = 0;
    for (; t + 2 * kInTiles <= ntiles; t += kInTiles) {
#pragma unroll
        for (int u = 0; u < kInTiles; u++) {
            process(u, t + u);
            fetch(u, t + u + kInTiles);
        }
    }
#pragma unroll
    for (int u = 0; u < kInTiles; u++) {
        if (t + u < ntiles) {
            process(u, t + u);
            if (t + u + kInTiles < ntiles) fetch(u, t + u + kInTiles);
        }
    }
    t += kInTiles;
#pragma unroll
    for (int u = 0; u < kInTiles; u++)
        if (t + u < ntiles) process(u, t + u);
    T *hp = y + lane * frames * R;
    for (size_t f = ntiles * F; f < frames; f++) {
        c.zoh = c.combs(xrow[f], m);
        for (size_t r = 0; r < R; r++) hp[f * R + r] = c.integrate(c.zoh);
    }
    c.store(st, lanes, lane, m);
}

inline bool cic_no_lm_tiles()
{
    static const bool v = diag_env("IDSP_CIC_NO_LM_TILES") != nullptr;
    return v;
}

inline int cfg_check(const idsp_cic *c)
{
    if (!c) return fail(IDSP_EINVAL, "cfg is NULL");
    if (c->order < 1 || c->order > IDSP_CIC_MAX_ORDER) return fail(IDSP_EINVAL, "Cic order N = %d not in 1..%d", c->order, IDSP_CIC_MAX_ORDER);
    if (c->comb_delay < 1 || c->comb_delay > IDSP_CIC_MAX_DELAY)
        return fail(IDSP_EINVAL, "Cic comb delay M = %d not in 1..%d (src/cic.rs:36: must be non-zero)", c->comb_delay, IDSP_CIC_MAX_DELAY);
    return IDSP_OK;
}

// Orders NLO .. NLO+2; the two halves (NLO = 1, 4) are instantiated in separate translation units.
template <class T, bool DEC, int NLO>
int run_orders(const idsp_cic *cfg, void *state, const T *x, T *y, size_t lanes, size_t frames, int layout, void *stream)
{
    int rc = cfg_check(cfg);
    if (rc) return rc;
    if ((rc = check_stream_args(cfg, 1, state, x, y, lanes, frames, layout))) return rc;
    if (lanes == 0 || frames == 0) return IDSP_OK;
    const dim3 grid(unsigned((lanes + kBlock - 1) / kBlock)), block(kBlock);
    uint32_t *st = static_cast<uint32_t *>(state);
    hipStream_t s = as_stream(stream);
    // whole 16-byte pieces per chunk and at least 64 chunks: one wave per lane (cic_ring.h; the decimator in LANE_MAJOR only)
    if (layout == IDSP_LANE_MAJOR || !DEC) {
        int rr;
        if constexpr (DEC)
            rr = cic_ring_dec(cfg, st, x, y, lanes, frames, s);
        else
            rr = cic_ring_int(cfg, st, x, y, lanes, frames, layout == IDSP_LANE_MAJOR, s);
        if (rr == 0) return launch_status();
        if (rr == 2) return IDSP_EHIP;
    }
    // vectors per chunk when the chunk is a whole number of 16-byte vectors and the rows are aligned
    using V = Vec16<T>;
    const size_t R = size_t(cfg->rate) + 1;
    int vpc = 0;
    if (R % V::n == 0 && reinterpret_cast<uintptr_t>(DEC ? static_cast<const void *>(x) : static_cast<const void *>(y)) % 16 == 0) {
        const size_t v = R / V::n;
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) vpc = int(v);
    }
    // LANE_MAJOR tile kernels: whole waves, at least one whole tile
    const bool lm_tiles = layout == IDSP_LANE_MAJOR && vpc > 0 && lanes % kBlock == 0 && frames >= size_t(kTileVecs / (vpc ? vpc : 1)) &&
                          reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
                          (frames * R * sizeof(T)) % 16 == 0 && !cic_no_lm_tiles();
#define IDSP_CIC_LAUNCH(NN, VV)                                                                                       \
    do {                                                                                                              \
        note_kernel(DEC ? "cic_dec_kernel" : "cic_int_kernel");                                                                                                              \
        if constexpr (VV > 0) {                                                                                       \
            if (lm_tiles) {                                                                                           \
                if constexpr (DEC)                                                                                    \
                    hipLaunchKernelGGL((cic_dec_lm_kernel<T, NN, VV>), grid, block, 0, s, *cfg, st, x, y, lanes, frames); \
                else                                                                                                  \
                    hipLaunchKernelGGL((cic_int_lm_kernel<T, NN, VV>), grid, block, 0, s, *cfg, st, x, y, lanes, frames); \
                break;                                                                                                \
            }                                                                                                         \
        }                                                                                                             \
        if constexpr (DEC)                                                                                            \
            hipLaunchKernelGGL((cic_dec_kernel<T, NN, VV>), grid, block, 0, s, *cfg, st, x, y, lanes, frames, layout); \
        else                                                                                                          \
            hipLaunchKernelGGL((cic_int_kernel<T, NN, VV>), grid, block, 0, s, *cfg, st, x, y, lanes, frames, layout); \
    } while (0)
#define IDSP_CIC_CASE(NN)                            \
    case NN:                                         \
        switch (vpc) {                               \
            case 1: IDSP_CIC_LAUNCH(NN, 1); break;   \
            case 2: IDSP_CIC_LAUNCH(NN, 2); break;   \
            case 4: IDSP_CIC_LAUNCH(NN, 4); break;   \
            case 8: IDSP_CIC_LAUNCH(NN, 8); break;   \
            case 16: IDSP_CIC_LAUNCH(NN, 16); break; \
            default: IDSP_CIC_LAUNCH(NN, 0); break;  \
        }                                            \
        break;
    switch (cfg->order) {
        IDSP_CIC_CASE(NLO)
        IDSP_CIC_CASE(NLO + 1)
        IDSP_CIC_CASE(NLO + 2)
    }
#undef IDSP_CIC_CASE
#undef IDSP_CIC_LAUNCH
    return launch_status();
}

template <class T, bool DEC>
int run(const idsp_cic *cfg, void *state, const T *x, T *y, size_t lanes, size_t frames, int layout, void *stream)
{
    const int rc = cfg_check(cfg);
    if (rc) return rc;
    return cfg->order <= 3 ? run_orders<T, DEC, 1>(cfg, state, x, y, lanes, frames, layout, stream)
                           : run_orders<T, DEC, 4>(cfg, state, x, y, lanes, frames, layout, stream);
}


}  // namespace cic
}  // namespace idsp
