// cic_ring.h — `Cic<T, N, M>` decimator (src/cic.rs:186-207 under `Decimator`, adapters.rs:158-167), LANE_MAJOR, one WAVE per
// lane on an LDS-DMA input ring.  Round 4; cic_kernels.h (one lane per THREAD) stays for every shape this does not take.
//
// Why: with one lane per thread 16384 lanes are 256 waves — one per CU — and each has to keep tens of KiB in flight from
// registers (0.47 of the HBM peak, spilling).  The integrators are a serial recurrence only as the reference writes them:
// they wrap (cic.rs:191), so they are linear maps over Z / 2^W and any re-association is exact.  With z the N integrators,
//     z' = A z + b x,   A = lower-triangular ones, b = ones        (z0 += x; z1 += z0; ... : cic.rs:189-193)
// a frame of R samples maps z -> A^R z + L, L = the frame's samples integrated from z = 0.  A^n is Toeplitz,
// (A^n)[i][j] = g_{i-j}(n) = C(n + i - j - 1, i - j), and Toeplitz triangles multiply like truncated polynomials, so
// the host gets g(R 2^k) by repeated squaring in u64.
//
// One block = 64 frames, thread t = frame t:
//   1. t integrates its R samples serially from zero (thread 0: from the block's incoming state S): L_t, and u_t = the last
//      integrator after sample 0 (the sample the decimator ticks on);
//   2. six Hillis-Steele steps over the wave, step k: V_t += A^(R 2^k) V_(t - 2^k) — afterwards V_t is the TRUE integrator
//      state behind frame t; V_(t-1) is what frame t started from, and its contribution to the ticked output is just the
//      sum of its entries (one step of A, last row): out_t = u_t + sum(V_(t-1));
//   3. the combs run at the low rate across threads: y_t = c_t - c_(t-M), the first M threads reading the previous
//      block's values (src/cic.rs:197-203).
// About 130 VALU + 4 LDS reads + 30 lane permutes per 64 frames of 16 i32 samples, against 1024 dependent adds per lane.
//
// Input: `global_load_lds_dwordx4`, a ring of two blocks (2 x PPT KiB per wave, PPT = 16-byte pieces per frame); the
// requests of block c + 2 are issued when block c has been read.  A thread needs the PPT pieces of ITS frame — 64-byte
// strides, a four-way bank conflict for `ds_read_b128` — so the request permutes on the global side instead: lane j of
// request k fetches piece PPT (j % TPR) + j / TPR of the request's KiB (TPR = 64 / PPT frames per request), which puts
// piece i of frame u at u + TPR i: consecutive threads read consecutive pieces.  (PPT = 8: i ^ ((k >> 1) & 1) instead of
// i, else the two requests inside one 16-lane read group would collide.)  Every request still covers one whole KiB.
#pragma once

#include <type_traits>

#include "common.h"
#include "idsp_hip.h"
#include "lds_dma.h"

namespace idsp {
namespace cicr {

constexpr int kW = 64;
constexpr int kSteps = 6;  // log2(kW)

template <class T, int N>
struct ScanCoef {
    typename std::make_unsigned<T>::type g[kSteps][N > 1 ? N - 1 : 1];  // g[k][d - 1] = g_d(R 2^k), d = 1 .. N-1
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int I, int E, class F>
__device__ __forceinline__ void static_for_k(F &&f)
{
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for_k<I + 1, E>(f);
    }
}

__device__ __forceinline__ uint32_t lane_get(uint32_t v, int src) { return uint32_t(__builtin_amdgcn_ds_bpermute((src & 63) * 4, int(v))); }
__device__ __forceinline__ uint64_t lane_get(uint64_t v, int src)
{
    const uint32_t lo = lane_get(uint32_t(v), src), hi = lane_get(uint32_t(v >> 32), src);
    return uint64_t(lo) | (uint64_t(hi) << 32);
}
__device__ __forceinline__ uint32_t lane_read(uint32_t v, int src) { return uint32_t(__builtin_amdgcn_readlane(int(v), src)); }
__device__ __forceinline__ uint64_t lane_read(uint64_t v, int src)
{
    return uint64_t(lane_read(uint32_t(v), src)) | (uint64_t(lane_read(uint32_t(v >> 32), src)) << 32);
}

template <class T, int N, int PPT>
__global__ __launch_bounds__(kW) void cic_dec_ring_lm(const ScanCoef<T, N> coef, const int m, uint32_t *st, const T *x, T *y,
                                                       const size_t lanes, const size_t frames)
{
    using UT = typename std::make_unsigned<T>::type;
    constexpr int VW = sizeof(T) / 4;
    constexpr int SPP = 16 / int(sizeof(T));  // samples per piece
    constexpr int R = PPT * SPP;              // samples per frame
    constexpr int TPR = kW / PPT;             // frames per request
    constexpr int HALFB = PPT * 1024;         // bytes per ring half = one block
    static_assert(PPT == 1 || PPT == 2 || PPT == 4 || PPT == 8, "16-byte pieces per frame");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_cic[];
    char *const ring = reinterpret_cast<char *>(smem_cic);
    const int lid = threadIdx.x;
    const size_t lane = blockIdx.x;

    auto ldv = [&](int v) -> UT {
        if constexpr (VW == 1)
            return UT(st[size_t(v) * lanes + lane]);
        else
            return UT(uint64_t(st[size_t(2 * v) * lanes + lane]) | (uint64_t(st[size_t(2 * v + 1) * lanes + lane]) << 32));
    };
    auto stv = [&](int v, UT val) {
        if constexpr (VW == 1) {
            st[size_t(v) * lanes + lane] = uint32_t(val);
        } else {
            st[size_t(2 * v) * lanes + lane] = uint32_t(val);
            st[size_t(2 * v + 1) * lanes + lane] = uint32_t(uint64_t(val) >> 32);
        }
    };
    // state (include/idsp_hip.h): zoh, combs[N][M] oldest first, integrators[N]
    UT S[N], zoh = ldv(0);
#pragma unroll
    for (int n = 0; n < N; n++) S[n] = ldv(1 + N * m + n);
    // comb n's delay line rides in the top M threads of "the block before": thread 64 - M + j holds combs[n][j]
    UT cprev[N], cold[N];
#pragma unroll
    for (int n = 0; n < N; n++) {
        cprev[n] = lid >= kW - m ? ldv(1 + n * m + (lid - (kW - m))) : UT(0);
        cold[n] = cprev[n];
    }

    const size_t npieces = frames * PPT;
    const char *const xl = reinterpret_cast<const char *>(x + lane * frames * R);
    T *const yl = y + lane * frames;
    const size_t rounds = (frames + kW - 1) / kW;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)ring;

    // mover role: lane j of request k moves piece PPT u + (i ^ flip(k)) of the request's KiB, u = j % TPR, i = j / TPR
    const int mu = lid % TPR, mi = lid / TPR;
    const uint32_t voff0 = uint32_t(PPT * mu + mi) * 16, voff1 = uint32_t(PPT * mu + (mi ^ 1)) * 16;
    auto flip_of = [](int k) { return PPT == 8 ? (k >> 1) & 1 : 0; };
    // owner role: piece i of this thread's frame sits at request lid / TPR, position (lid % TPR) + TPR (i ^ flip)
    const int ok = lid / TPR;
    const uint32_t rd_base = uint32_t(ok) * 1024 + uint32_t(lid % TPR) * 16;
    const int rflip = flip_of(ok);
    const uint32_t rd_even = rd_base + (rflip ? TPR * 16 : 0), rd_odd = rd_base - (rflip ? TPR * 16 : 0);

    auto request_safe = [&](size_t b) {  // block b into ring half b % 2, addresses clamped to the row
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const size_t p = b * size_t(kW * PPT) + size_t(k * kW) + (flip_of(k) ? voff1 : voff0) / 16;
            glds16(xl + (p < npieces ? p * 16 : 0), ring_lds + uint32_t(b & 1) * HALFB + k * 1024);
        }
    };
    auto request_fast = [&](size_t b, auto half_) {  // whole block inside the row: SGPR base, immediate offsets
        constexpr int HALF = decltype(half_)::value;
        const char *xb = uniform_ptr(xl + b * size_t(HALFB));
        static_for_k<0, PPT>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            // the instruction offset moves the global AND the LDS address (lds_dma.h): (k % 4) KiB on both sides
            glds16_si<(k % 4) * 1024>(xb + (k / 4) * 4096, (PPT == 8 && ((k >> 1) & 1)) ? voff1 : voff0,
                                      ring_lds + HALF * HALFB + (k / 4) * 4096);
        });
    };

    auto block = [&](auto fast_, auto half_, size_t c) {
        constexpr bool FAST = decltype(fast_)::value;
        constexpr int HALF = decltype(half_)::value;
        const int nlast = FAST ? kW : int(frames - c * kW < size_t(kW) ? frames - c * kW : size_t(kW));
        // requests of block c: issued two blocks ago; behind them in the queue: store(c-2), the PPT requests of block c+1, store(c-1)
        if constexpr (FAST)
            wait_vmcnt<PPT + 2>();
        else
            wait_vmcnt<0>();
        UT xs[R];
        static_for_k<0, PPT>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            const u32x4 pc = *reinterpret_cast<const u32x4 *>(ring + HALF * HALFB + ((i & 1) ? rd_odd : rd_even) + i * TPR * 16);
            if constexpr (VW == 1) {
                xs[4 * i] = pc.x, xs[4 * i + 1] = pc.y, xs[4 * i + 2] = pc.z, xs[4 * i + 3] = pc.w;
            } else {
                xs[2 * i] = UT(uint64_t(pc.x) | (uint64_t(pc.y) << 32));
                xs[2 * i + 1] = UT(uint64_t(pc.z) | (uint64_t(pc.w) << 32));
            }
        });
        lds_wave_sync();  // the pieces are in registers: the half can be refilled
        if constexpr (FAST)
            request_fast(c + 2, half_);
        else
            request_safe(c + 2);

        // 1. the frame from zero (thread 0: from the incoming state)
        UT z[N], u;
#pragma unroll
        for (int n = 0; n < N; n++) z[n] = lid == 0 ? S[n] : UT(0);
#pragma unroll
        for (int s = 0; s < R; s++) {
            UT v = xs[s];
#pragma unroll
            for (int n = 0; n < N; n++) {
                z[n] += v;
                v = z[n];
            }
            if (s == 0) u = v;
        }
        // 2. scan: z_t <- sum_j A^(R (t - j)) L_j
#pragma unroll
        for (int k = 0; k < kSteps; k++) {
            const int d = 1 << k;
            UT w[N];
#pragma unroll
            for (int n = 0; n < N; n++) {
                w[n] = lane_get(z[n], lid - d);
                if (lid < d) w[n] = 0;
            }
#pragma unroll
            for (int i = N - 1; i >= 0; i--) {
                UT acc = w[i];
#pragma unroll
                for (int j = 0; j < i; j++) acc += coef.g[k][i - j - 1] * w[j];
                z[i] += acc;
            }
        }
        UT q = 0;
#pragma unroll
        for (int n = 0; n < N; n++) {
            const UT p = lane_get(z[n], lid - 1);
            q += lid == 0 ? UT(0) : p;
            S[n] = lane_read(z[n], nlast - 1);
        }
        u += q;
        // 3. combs
#pragma unroll
        for (int n = 0; n < N; n++) {
            const UT a = lane_get(u, lid - m), b = lane_get(cprev[n], lid - m);
            cold[n] = cprev[n];
            cprev[n] = u;
            u -= lid >= m ? a : b;
        }
        if (FAST || lid < nlast) yl[c * kW + lid] = T(u);
        zoh = lane_read(u, nlast - 1);
    };

    request_safe(0);
    request_safe(1);
    size_t c = 0;
    int nlast = 0;
    for (; c < rounds; c++) {
        const bool fast = c >= 2 && (c + 3) * size_t(kW) <= frames;
        nlast = int(frames - c * kW < size_t(kW) ? frames - c * kW : size_t(kW));
        if (c & 1) {
            if (fast)
                block(std::true_type{}, std::integral_constant<int, 1>{}, c);
            else
                block(std::false_type{}, std::integral_constant<int, 1>{}, c);
        } else {
            if (fast)
                block(std::true_type{}, std::integral_constant<int, 0>{}, c);
            else
                block(std::false_type{}, std::integral_constant<int, 0>{}, c);
        }
    }
    wait_vmcnt<0>();  // the requests past the end are still landing: they must not outlive the wave's LDS

    // write-back.  Comb n's delay line = the last M inputs of comb n, oldest first: thread nlast - M + j of the last block,
    // or — negative — thread 64 + that of the block before (the incoming delay line if there was none).
#pragma unroll
    for (int n = 0; n < N; n++) {
        const int p = nlast + lid - m;
        const UT a = lane_get(cprev[n], p), b = lane_get(cold[n], p);
        if (lid < m) stv(1 + n * m + lid, p >= 0 ? a : b);
    }
    if (lid == 0) {
        stv(0, zoh);
#pragma unroll
        for (int n = 0; n < N; n++) stv(1 + N * m + n, S[n]);
    }
}

// ------------------------------------------------------------------------------------------------ interpolator
// `Cic<T, N, M>` interpolator (src/cic.rs:160-182 under `Interpolator`, adapters.rs:27-35), LANE_MAJOR, one wave per lane.
// The combs run at the low rate across threads first; a frame then integrates R copies of its held value `zoh`, which from
// zero is h * zoh with h = R steps of the chain on input 1 (host, u64).  Same scan as the decimator; thread t then
// replays its R steps from the true state behind frame t - 1 and owns the frame's R outputs = PPT 16-byte pieces.  They go
// through a padded LDS tile (pitch PPT + 1 pieces: conflict-free writes) and leave as PPT stores of one whole KiB each.
template <class T, int N>
struct IntCoef {
    ScanCoef<T, N> scan;
    typename std::make_unsigned<T>::type h[N];  // the chain after R steps on constant input 1 from zero
};

template <class T, int N, int PPT>
__global__ __launch_bounds__(kW) void cic_int_ring_lm(const IntCoef<T, N> coef, const int m, uint32_t *st, const T *x, T *y,
                                                       const size_t lanes, const size_t frames)
{
    using UT = typename std::make_unsigned<T>::type;
    constexpr int VW = sizeof(T) / 4;
    constexpr int SPP = 16 / int(sizeof(T));
    constexpr int R = PPT * SPP;
    constexpr int PITCH = (PPT + 1) * 16;  // bytes per frame in the tile
    static_assert(PPT == 1 || PPT == 2 || PPT == 4 || PPT == 8, "16-byte pieces per frame");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_cic[];
    char *const tile = reinterpret_cast<char *>(smem_cic);
    const int lid = threadIdx.x;
    const size_t lane = blockIdx.x;

    auto ldv = [&](int v) -> UT {
        if constexpr (VW == 1)
            return UT(st[size_t(v) * lanes + lane]);
        else
            return UT(uint64_t(st[size_t(2 * v) * lanes + lane]) | (uint64_t(st[size_t(2 * v + 1) * lanes + lane]) << 32));
    };
    auto stv = [&](int v, UT val) {
        if constexpr (VW == 1) {
            st[size_t(v) * lanes + lane] = uint32_t(val);
        } else {
            st[size_t(2 * v) * lanes + lane] = uint32_t(val);
            st[size_t(2 * v + 1) * lanes + lane] = uint32_t(uint64_t(val) >> 32);
        }
    };
    UT S[N], zoh = ldv(0);
#pragma unroll
    for (int n = 0; n < N; n++) S[n] = ldv(1 + N * m + n);
    UT cprev[N], cold[N];
#pragma unroll
    for (int n = 0; n < N; n++) {
        cprev[n] = lid >= kW - m ? ldv(1 + n * m + (lid - (kW - m))) : UT(0);
        cold[n] = cprev[n];
    }

    const T *const xl = x + lane * frames;
    char *const yl = reinterpret_cast<char *>(y + lane * frames * R);
    const size_t rounds = (frames + kW - 1) / kW;
    // output mover: in store k, thread j moves piece 64 k + j of the block = piece j % PPT of frame (64 k + j) / PPT
    int rd[PPT];
#pragma unroll
    for (int k = 0; k < PPT; k++) rd[k] = ((k * kW + lid) / PPT) * PITCH + ((k * kW + lid) % PPT) * 16;

    UT xn = lid < int(frames < size_t(kW) ? frames : size_t(kW)) ? UT(xl[lid]) : UT(0);
    int nlast = 0;
    for (size_t c = 0; c < rounds; c++) {
        nlast = int(frames - c * kW < size_t(kW) ? frames - c * kW : size_t(kW));
        UT v = xn;
        {  // the next block's input: a block ahead, behind this block's stores in the queue
            const size_t f = (c + 1) * kW + lid;
            xn = f < frames ? UT(xl[f]) : UT(0);
        }
        // combs (src/cic.rs:166-171)
#pragma unroll
        for (int n = 0; n < N; n++) {
            const UT a = lane_get(v, lid - m), b = lane_get(cprev[n], lid - m);
            cold[n] = cprev[n];
            cprev[n] = v;
            v -= lid >= m ? a : b;
        }
        zoh = lane_read(v, nlast - 1);
        // the frame's map: A^R (state) + h v; thread 0 carries the incoming state
        UT z[N];
#pragma unroll
        for (int i = N - 1; i >= 0; i--) {
            UT acc = S[i];
#pragma unroll
            for (int j = 0; j < i; j++) acc += coef.scan.g[0][i - j - 1] * S[j];
            z[i] = (lid == 0 ? acc : UT(0)) + coef.h[i] * v;
        }
#pragma unroll
        for (int k = 0; k < kSteps; k++) {
            const int d = 1 << k;
            UT w[N];
#pragma unroll
            for (int n = 0; n < N; n++) {
                w[n] = lane_get(z[n], lid - d);
                if (lid < d) w[n] = 0;
            }
#pragma unroll
            for (int i = N - 1; i >= 0; i--) {
                UT acc = w[i];
#pragma unroll
                for (int j = 0; j < i; j++) acc += coef.scan.g[k][i - j - 1] * w[j];
                z[i] += acc;
            }
        }
        // the state this frame starts from, and the block's successor state
        UT pz[N];
#pragma unroll
        for (int n = 0; n < N; n++) {
            const UT p = lane_get(z[n], lid - 1);
            pz[n] = lid == 0 ? S[n] : p;
        }
#pragma unroll
        for (int n = 0; n < N; n++) S[n] = lane_read(z[n], nlast - 1);
        // the frame's R outputs (src/cic.rs:174-180), PPT pieces into the tile
        static_for_k<0, PPT>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            UT o[SPP];
#pragma unroll
            for (int e = 0; e < SPP; e++) {
                UT t = v;
#pragma unroll
                for (int n = 0; n < N; n++) {
                    pz[n] += t;
                    t = pz[n];
                }
                o[e] = t;
            }
            u32x4 pc;
            if constexpr (VW == 1)
                pc = u32x4{uint32_t(o[0]), uint32_t(o[1]), uint32_t(o[2]), uint32_t(o[3])};
            else
                pc = u32x4{uint32_t(o[0]), uint32_t(uint64_t(o[0]) >> 32), uint32_t(o[1]), uint32_t(uint64_t(o[1]) >> 32)};
            *reinterpret_cast<u32x4 *>(tile + lid * PITCH + i * 16) = pc;
        });
        lds_wave_sync();
        char *dst = yl + c * size_t(kW * PPT * 16) + size_t(lid) * 16;
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const u32x4 pc = *reinterpret_cast<const u32x4 *>(tile + rd[k]);
            if ((k * kW + lid) / PPT < nlast) __builtin_nontemporal_store(pc, reinterpret_cast<u32x4 *>(dst + k * 1024));
        }
        lds_wave_sync();  // the tile has been read: the next block may overwrite it
    }

#pragma unroll
    for (int n = 0; n < N; n++) {
        const int p = nlast + lid - m;
        const UT a = lane_get(cprev[n], p), b = lane_get(cold[n], p);
        if (lid < m) stv(1 + n * m + lid, p >= 0 ? a : b);
    }
    if (lid == 0) {
        stv(0, zoh);
#pragma unroll
        for (int n = 0; n < N; n++) stv(1 + N * m + n, S[n]);
    }
}

// FRAME_MAJOR interpolator: x[f * lanes + l], y[(f * lanes + l) * R + r].  16 lanes = 16 waves per workgroup; the
// arithmetic is the LANE_MAJOR kernel's.  A block's inputs (64 frames x 16 lanes) arrive as one 4- or 8-byte load per
// thread — 64-byte runs — through a [64][17] LDS tile; the outputs of frame f, all 16 lanes, are 16 R sizeof(T)
// contiguous bytes of y: they are collected in a [64 frames][16 lanes x R + one piece of padding] tile (conflict-free
// 16-byte writes) and leave as whole-KiB stores.  Two workgroup barriers per block.
constexpr int kFmLanes = 16;

template <class T, int N, int PPT>
__global__ __launch_bounds__(kFmLanes *kW) void cic_int_ring_fm(const IntCoef<T, N> coef, const int m, uint32_t *st, const T *x, T *y,
                                                                 const size_t lanes, const size_t frames)
{
    using UT = typename std::make_unsigned<T>::type;
    constexpr int VW = sizeof(T) / 4;
    constexpr int SPP = 16 / int(sizeof(T));
    constexpr int R = PPT * SPP;
    constexpr int ROWP = kFmLanes * PPT;           // pieces per output row
    constexpr int PITCH = (ROWP + 1) * 16;         // bytes per tile row
    constexpr int XP = kFmLanes + 1;               // elements per input-tile row
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_cic[];
    char *const tile = reinterpret_cast<char *>(smem_cic);
    UT *const xin = reinterpret_cast<UT *>(tile + kW * PITCH);
    const int lid = threadIdx.x % kW, w = __builtin_amdgcn_readfirstlane(threadIdx.x / kW);
    const int q = threadIdx.x;
    const size_t ngroups = lanes / kFmLanes, per = (ngroups + 7) / 8;
    const size_t group = (blockIdx.x % 8) * per + blockIdx.x / 8;  // every XCD a contiguous eighth of the lane groups
    if (group >= ngroups) return;
    const size_t lane0 = group * kFmLanes, lane = lane0 + w;

    auto ldv = [&](int v) -> UT {
        if constexpr (VW == 1)
            return UT(st[size_t(v) * lanes + lane]);
        else
            return UT(uint64_t(st[size_t(2 * v) * lanes + lane]) | (uint64_t(st[size_t(2 * v + 1) * lanes + lane]) << 32));
    };
    auto stv = [&](int v, UT val) {
        if constexpr (VW == 1) {
            st[size_t(v) * lanes + lane] = uint32_t(val);
        } else {
            st[size_t(2 * v) * lanes + lane] = uint32_t(val);
            st[size_t(2 * v + 1) * lanes + lane] = uint32_t(uint64_t(val) >> 32);
        }
    };
    UT S[N], zoh = ldv(0);
#pragma unroll
    for (int n = 0; n < N; n++) S[n] = ldv(1 + N * m + n);
    UT cprev[N], cold[N];
#pragma unroll
    for (int n = 0; n < N; n++) {
        cprev[n] = lid >= kW - m ? ldv(1 + n * m + (lid - (kW - m))) : UT(0);
        cold[n] = cprev[n];
    }

    const size_t rounds = (frames + kW - 1) / kW;
    // input mover: thread q fetches frame q / 16, lane q % 16 of the block
    const int xf = q / kFmLanes, xlane = q % kFmLanes;
    auto fetch = [&](size_t c) -> UT {
        const size_t f = c * kW + xf;
        return f < frames ? UT(x[f * lanes + lane0 + xlane]) : UT(0);
    };
    // output mover: in pass k, thread q moves piece k * 1024 + q of the block's [64][ROWP] pieces
    int rrow[PPT], rcol[PPT];
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        rrow[k] = (k * kFmLanes * kW + q) / ROWP;
        rcol[k] = (k * kFmLanes * kW + q) % ROWP;
    }
    const size_t yrow = lanes * size_t(R) * sizeof(T);  // bytes per frame row of y
    char *const ybase = reinterpret_cast<char *>(y) + lane0 * size_t(R) * sizeof(T);

    UT xn = fetch(0);
    int nlast = 0;
    for (size_t c = 0; c < rounds; c++) {
        nlast = int(frames - c * kW < size_t(kW) ? frames - c * kW : size_t(kW));
        xin[xf * XP + xlane] = xn;
        xn = fetch(c + 1);
        lds_barrier();  // the input tile is complete; everybody has finished reading the output tile of the block before
        UT v = xin[lid * XP + w];
#pragma unroll
        for (int n = 0; n < N; n++) {
            const UT a = lane_get(v, lid - m), b = lane_get(cprev[n], lid - m);
            cold[n] = cprev[n];
            cprev[n] = v;
            v -= lid >= m ? a : b;
        }
        zoh = lane_read(v, nlast - 1);
        UT z[N];
#pragma unroll
        for (int i = N - 1; i >= 0; i--) {
            UT acc = S[i];
#pragma unroll
            for (int j = 0; j < i; j++) acc += coef.scan.g[0][i - j - 1] * S[j];
            z[i] = (lid == 0 ? acc : UT(0)) + coef.h[i] * v;
        }
#pragma unroll
        for (int k = 0; k < kSteps; k++) {
            const int d = 1 << k;
            UT ww[N];
#pragma unroll
            for (int n = 0; n < N; n++) {
                ww[n] = lane_get(z[n], lid - d);
                if (lid < d) ww[n] = 0;
            }
#pragma unroll
            for (int i = N - 1; i >= 0; i--) {
                UT acc = ww[i];
#pragma unroll
                for (int j = 0; j < i; j++) acc += coef.scan.g[k][i - j - 1] * ww[j];
                z[i] += acc;
            }
        }
        UT pz[N];
#pragma unroll
        for (int n = 0; n < N; n++) {
            const UT p = lane_get(z[n], lid - 1);
            pz[n] = lid == 0 ? S[n] : p;
        }
#pragma unroll
        for (int n = 0; n < N; n++) S[n] = lane_read(z[n], nlast - 1);
        static_for_k<0, PPT>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            UT o[SPP];
#pragma unroll
            for (int e = 0; e < SPP; e++) {
                UT t = v;
#pragma unroll
                for (int n = 0; n < N; n++) {
                    pz[n] += t;
                    t = pz[n];
                }
                o[e] = t;
            }
            u32x4 pc;
            if constexpr (VW == 1)
                pc = u32x4{uint32_t(o[0]), uint32_t(o[1]), uint32_t(o[2]), uint32_t(o[3])};
            else
                pc = u32x4{uint32_t(o[0]), uint32_t(uint64_t(o[0]) >> 32), uint32_t(o[1]), uint32_t(uint64_t(o[1]) >> 32)};
            *reinterpret_cast<u32x4 *>(tile + lid * PITCH + (w * PPT + i) * 16) = pc;
        });
        lds_barrier();  // the output tile is complete; everybody has read its input
        char *dst = ybase + c * size_t(kW) * yrow;
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const u32x4 pc = *reinterpret_cast<const u32x4 *>(tile + rrow[k] * PITCH + rcol[k] * 16);
            if (rrow[k] < nlast) __builtin_nontemporal_store(pc, reinterpret_cast<u32x4 *>(dst + size_t(rrow[k]) * yrow + rcol[k] * 16));
        }
    }

#pragma unroll
    for (int n = 0; n < N; n++) {
        const int p = nlast + lid - m;
        const UT a = lane_get(cprev[n], p), b = lane_get(cold[n], p);
        if (lid < m) stv(1 + n * m + lid, p >= 0 ? a : b);
    }
    if (lid == 0) {
        stv(0, zoh);
#pragma unroll
        for (int n = 0; n < N; n++) stv(1 + N * m + n, S[n]);
    }
}

}  // namespace cicr

// 0: a ring kernel was launched; 1: shape not covered (the caller falls back to cic_kernels.h); 2: HIP error
int cic_ring_dec(const idsp_cic *cfg, uint32_t *st, const int32_t *x, int32_t *y, size_t lanes, size_t frames, hipStream_t stream);
int cic_ring_dec(const idsp_cic *cfg, uint32_t *st, const int64_t *x, int64_t *y, size_t lanes, size_t frames, hipStream_t stream);
int cic_ring_int(const idsp_cic *cfg, uint32_t *st, const int32_t *x, int32_t *y, size_t lanes, size_t frames, bool lane_major,
                 hipStream_t stream);
int cic_ring_int(const idsp_cic *cfg, uint32_t *st, const int64_t *x, int64_t *y, size_t lanes, size_t frames, bool lane_major,
                 hipStream_t stream);

}  // namespace idsp
