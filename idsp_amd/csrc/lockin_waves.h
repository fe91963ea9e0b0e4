// lockin_waves.h — the lock-in (src/lockin.rs:30-39) with the work of one lane spread over several waves.
//
// One thread per lane runs, per frame, cossin (~20 VALU instructions), two `[Lowpass<N>; K]` arms (~30 each for
// N = K = 2: a mixer `v_mul_hi`, four `v_mad_i64_i32`, eight 64-bit adds) and, for the phase read-out, atan2 (~70),
// many of them multi-pass 64-bit operations.  At the C4 lane counts that is one wave per SIMD or fewer, where a SIMD
// issues an instruction only every 6-10 cycles instead of every 3-5, and the kernel is VALU-bound far below the HBM
// roofline.  Here a workgroup is 64 lanes x W waves with wave-uniform roles: wave 0 runs the I arm, wave 1 the Q arm,
// the other W - 2 "read-out" waves evaluate the LO (cossin) for a share of each 8-frame batch and turn the arm outputs
// of the previous batch into the output element: `Complex<i32>` [re, im] (MODE_IQ), `Complex::arg()` (MODE_ARG,
// src/complex.rs:254-256) or `Complex::norm_sqr()` (MODE_NORM_SQR, src/complex.rs:214-217).  cos/sin and the arm
// outputs travel through double-buffered LDS with one barrier per batch, so arms of batch n, LO of batch n + 1 and
// read-out of batch n - 1 overlap.  Every FrameMajor store instruction writes one contiguous row segment of the 64
// lanes (256 or 512 bytes); LaneMajor (whole batches on 16-byte aligned rows only) moves 16-byte vectors per thread.
//
// Input (IN): the arm waves are a serial recurrence with nothing to hide a global load behind, and a register
// prefetch one batch ahead leaves most of the HBM latency exposed (measured: 0.58 ms with, 0.44 ms without the loads
// at 32768 lanes x 4096 frames).  IN_FM_DMA (whole 64-lane workgroups, 16-byte aligned rows) therefore has each arm
// wave issue one `global_load_lds_dwordx4` per batch -- 4 rows x 256 bytes straight into an LDS ring, three batches
// ahead, no VGPRs -- and waits with `vmcnt(2)` at the end of an interval for the one issued two intervals earlier.
// IN_FM_REG / IN_LM_REG fetch the next batch into registers (any shape the kernel takes; the same DMA on LaneMajor
// rows, 32 lanes x 32 bytes per instruction, re-fetches every 128-byte line four times and was slower: 0.79 vs 0.64 ms;
// so was staging the LaneMajor input through LDS by the read-out waves with cached loads: 0.65 vs 0.58 ms -- the
// LaneMajor form is bound by its 32-byte-per-lane output pieces, not by the input).
#pragma once
#include "dds_dev.h"

namespace idsp {
namespace {

constexpr int kLwB = 8;     // frames per batch, FrameMajor
constexpr int kLwBLm = 16;  // LaneMajor: 16 frames = one whole 128-byte line of 8-byte output elements per lane and batch
enum { MODE_IQ = 0, MODE_ARG = 1, MODE_NORM_SQR = 2 };
enum { IN_FM_REG = 0, IN_FM_DMA = 1, IN_LM_REG = 2, IN_LM_DMA = 3 };
constexpr int kLwRing = 4, kLwAhead = 3;  // LDS input ring slots, batches in flight
#ifndef IDSP_LW_OUT_GROUP
#define IDSP_LW_OUT_GROUP 1
#endif
constexpr int kLwOutGroup = IDSP_LW_OUT_GROUP;  // LaneMajor: batches a read-out thread stores together

template <int MODE>
struct LwOut {
    using type = int32_t;
};
template <>
struct LwOut<MODE_IQ> {
    using type = Cplx;
};
template <>
struct LwOut<MODE_NORM_SQR> {
    using type = int64_t;
};

template <int N, int K, int W, int IN, int MODE, int B>
__global__ __launch_bounds__(W * kWave) __attribute__((amdgpu_waves_per_eu(1, IN == IN_LM_REG || IN == IN_LM_DMA ? 2 : 10))) void lockin_waves_kernel(const LpParams prm, uint32_t *st, const int32_t *x,
                                                                 typename LwOut<MODE>::type *y, const size_t lanes, const size_t frames)
{
    using Out = typename LwOut<MODE>::type;
    constexpr bool LMD = IN == IN_LM_DMA, LM = IN == IN_LM_REG || LMD, DMA = IN == IN_FM_DMA;
    constexpr int kLut = 1 << kCossinDepth;
    constexpr int P = W - 2, C = B / P;  // read-out waves; each takes frames b = r * C + j, j < C, of a batch
    static_assert(B % P == 0 && B % 8 == 0, "batch splits evenly over the read-out waves and into 4-row DMA groups per arm wave");
    __shared__ uint32_t lut[kLut];
    __shared__ uint32_t tab[32];
    __shared__ Cplx lo[2][B][kWave];
    __shared__ int32_t arm[2][2][B][kWave];  // [buffer][I/Q][frame][lane]
    // FM DMA: [slot][frame][lane]; LM DMA: kLwRing slots of 64 lanes x 128 bytes (two batches), rows permuted and pieces swizzled
    __shared__ __attribute__((aligned(16))) int32_t xs[DMA ? kLwRing : LMD ? 2 * kLwRing : 1][B * kWave];
    const int w = threadIdx.x / kWave, lid = threadIdx.x % kWave;
    const bool arm_wave = w < 2;         // wave-uniform role
    const int r = arm_wave ? w : w - 2;  // arm waves: I / Q; read-out waves: frame group
    const size_t lane = size_t(blockIdx.x) * kWave + lid;
    const bool active = lane < lanes;
    const size_t la = active ? lane : lanes - 1;  // idle threads of the last workgroup shadow a valid lane, stores masked
    fill_cossin(lut, threadIdx.x, W * kWave);
    if (MODE == MODE_ARG && threadIdx.x < 32) tab[threadIdx.x] = d_atan2_table[threadIdx.x];
    const uint32_t acc0 = st[la], inc = st[lanes + la];
    LpBank<N, K> bank;
    if (arm_wave) bank.load(st, lanes, la, 2 + (r ? 2 * N * K : 0));
    // FrameMajor row base pointers are wave-uniform and the lane offset is one 32-bit register
    const uint32_t lo32 = uint32_t(la), lane32 = uint32_t(lane);
    uint32_t phase = acc0;  // accumulator before the batch whose LO is produced next
    int32_t xn[B];
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    auto fetch = [&](size_t f0, auto full) {
        if constexpr (LM) {
            const i32x4 *row = reinterpret_cast<const i32x4 *>(x + la * frames + f0);
#pragma unroll
            for (int v = 0; v < B / 4; v++) {
                const i32x4 a = row[v];
                xn[4 * v] = a.x, xn[4 * v + 1] = a.y, xn[4 * v + 2] = a.z, xn[4 * v + 3] = a.w;
            }
        } else {
#pragma unroll
            for (int b = 0; b < B; b++) {
                const int32_t *row = x + (f0 + b) * lanes;
                xn[b] = (decltype(full)::value || f0 + b < frames) ? row[lo32] : 0;
            }
        }
    };
    // arm wave r moves rows r B/2 .. of batch n, 4 rows per instruction: lane l takes the 16 bytes at column (l % 16) * 4 of row l / 16;
    // rows past the end re-read the last frame (never consumed) so that every interval issues exactly one operation
    auto dma = [&](size_t n) {
#pragma unroll
        for (int g = 0; g < B / 8; g++) {
            const int r0 = r * (B / 2) + 4 * g;  // first of the 4 rows this instruction moves
            size_t row = n * B + size_t(r0 + lid / 16);
            row = row < frames ? row : frames - 1;
            glds16(x + row * lanes + size_t(blockIdx.x) * kWave + size_t(lid % 16) * 4,
                   uint32_t(reinterpret_cast<uintptr_t>(&xs[n % kLwRing][r0 * kWave])));
        }
    };
    // LaneMajor input by DMA: a pair of batches = one whole 128-byte line per lane.  Instruction j of a pair fetches the lines
    // of lanes j, j + 8, ... (8 threads per line) so that every line is requested once, by one instruction; thread t takes
    // piece (t % 8) ^ j, which leaves lane l's piece k at slot row (l % 8) 8 + l / 8, offset 16 (k ^ (l % 8)): the arm
    // threads' ds_read_b128 of their own row are conflict free.  Arm wave r issues instructions 4 r .. 4 r + 3; pieces past
    // the end of the row (odd number of batches) and pairs past the end re-read frame 0 (never consumed), so that every
    // pair issues exactly four operations per arm wave.
    auto dma_lm = [&](size_t pair) {
        if constexpr (LMD) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int j = 4 * r + g;
                size_t gl = size_t(blockIdx.x) * kWave + size_t(j + 8 * (lid / 8));
                gl = gl < lanes ? gl : lanes - 1;
                const size_t f = pair * 32 + size_t(((lid % 8) ^ j) * 4);
                glds16(x + gl * frames + (f + 4 <= frames ? f : 0),
                       uint32_t(reinterpret_cast<uintptr_t>(&xs[0][0])) + uint32_t((pair % kLwRing) * 8192 + j * 1024));
            }
        }
    };
    const uint32_t xs_own = uint32_t((lid % 8) * 8 + lid / 8) * 128 + uint32_t(lid % 8) * 16;  // LM DMA: own row, piece k at ^ 16 k
    auto lo_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < C; j++) {
            const int b = r * C + j;
            lo[buf][b][lid] = cossin_dev(int32_t(phase + inc * uint32_t(b + 1)), lut);
        }
        phase += inc * uint32_t(B);
    };
    auto element = [&](int buf, int b, int ln) -> Out {
        const int32_t re = arm[buf][0][b][ln], im = arm[buf][1][b][ln];
        if constexpr (MODE == MODE_IQ)
            return Cplx{re, im};
        else if constexpr (MODE == MODE_ARG)
            return atan2_dev(im, re, tab);
        else
            return int64_t(uint64_t(int64_t(re) * re) + uint64_t(int64_t(im) * im));  // wraps for (MIN, MIN) as in release
    };
    // LaneMajor: a read-out thread keeps its pieces of kLwOutGroup batches in registers and stores them together, so that the
    // wave leaves kLwOutGroup * 16 frames (512 bytes of Complex<i32>) per lane in one burst of store instructions
    // instead of one 128-byte line per lane and batch (run length per lane is what the LaneMajor rate follows, see
    // stream_lane_major_staged)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int GV = int(sizeof(Out)) * C / 16;  // 16-byte vectors a read-out thread writes per batch (LaneMajor)
    u32x4 held[LM ? kLwOutGroup : 1][LM ? GV : 1];
    // `slot` (static): position of the batch inside its output group; `flush`: last batch of the call
    auto out_stage = [&](size_t f0, int buf, int nb, auto full, auto slot_tag, bool flush) {
        if constexpr (LM) {
            // read-out wave r writes lanes r * 64 / P ..: P adjacent threads cover the 16 frames of one lane, so that one store
            // instruction leaves 8 * sizeof(Out) contiguous bytes per lane instead of two (four) pieces at different times
            static_assert((sizeof(Out) * C) % 16 == 0, "a thread's piece of a batch is whole 16-byte vectors");
            const int ll = r * (kWave / P) + lid / P, part = lid % P;
            const size_t gl = size_t(blockIdx.x) * kWave + size_t(ll);
            constexpr int slot = decltype(slot_tag)::value;
            constexpr int OW = int(sizeof(Out)) / 4;
            uint32_t w[C * OW];
#pragma unroll
            for (int j = 0; j < C; j++) {
                const Out e = element(buf, part * C + j, ll);
                if constexpr (OW == 1) {
                    w[j] = __builtin_bit_cast(uint32_t, e);
                } else {
                    const uint64_t u = __builtin_bit_cast(uint64_t, e);
                    w[2 * j] = uint32_t(u), w[2 * j + 1] = uint32_t(u >> 32);
                }
            }
#pragma unroll
            for (int v = 0; v < GV; v++) held[slot][v] = u32x4{w[4 * v], w[4 * v + 1], w[4 * v + 2], w[4 * v + 3]};
            if (slot == kLwOutGroup - 1 || flush) {
                u32x4 *dst = reinterpret_cast<u32x4 *>(y + gl * frames + (f0 - size_t(slot) * B) + part * C);
#pragma unroll
                for (int q = 0; q <= slot; q++)
                    if (gl < lanes) {
#pragma unroll
                        for (int v = 0; v < GV; v++) dst[q * (B * int(sizeof(Out)) / 16) + v] = held[q][v];
                    }
            }
        } else {
#pragma unroll
            for (int j = 0; j < C; j++) {
                const int b = r * C + j;
                if ((decltype(full)::value || b < nb) && active) {
                    Out *row = y + (f0 + b) * lanes;
                    nt_store<true>(row + lane32, element(buf, b, lid));  // 8-byte elements leave as one 2-word vector
                }
            }
        }
    };
    auto iter = [&](size_t n, int nb, auto full, auto first, auto slot_tag) {
        const size_t f0 = n * B;
        const int buf = int(n & 1);
        if (arm_wave) {
            int32_t xv[B];
            if constexpr (DMA) {
                dma(n + kLwAhead);
#pragma unroll
                for (int b = 0; b < B; b++) xv[b] = xs[n % kLwRing][b * kWave + lid];
            } else if constexpr (LMD) {
                static_assert(B == 16, "a pair of batches is one 128-byte line per lane");
                if (n % 2 == 0) dma_lm(n / 2 + kLwAhead);
                const char *slot = reinterpret_cast<const char *>(&xs[0][0]) + ((n / 2) % kLwRing) * 8192;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const i32x4 a = *reinterpret_cast<const i32x4 *>(slot + (xs_own ^ uint32_t(((n % 2) * 4 + v) * 16)));
                    xv[4 * v] = a.x, xv[4 * v + 1] = a.y, xv[4 * v + 2] = a.z, xv[4 * v + 3] = a.w;
                }
            } else {
#pragma unroll
                for (int b = 0; b < B; b++) xv[b] = xn[b];
                if (f0 + 2 * B <= frames)
                    fetch(f0 + B, std::true_type{});
                else if (f0 + B < frames)
                    fetch(f0 + B, std::false_type{});
            }
            const int32_t *lo_mine = reinterpret_cast<const int32_t *>(&lo[buf][0][lid]) + r;  // this arm's LO component
#pragma unroll
            for (int b = 0; b < B; b++)
                if (decltype(full)::value || b < nb) arm[buf][r][b][lid] = bank.step(prm, __mulhi(lo_mine[b * kWave * 2], xv[b]));
            if constexpr (DMA) wait_vmcnt<(kLwAhead - 1) * (B / 8)>();  // batch n + 1 has landed
            if constexpr (LMD) {
                if (n % 2 == 1) wait_vmcnt<(kLwAhead - 1) * 4>();  // the pair of batches n + 1, n + 2 has landed
            }
        } else {
            lo_stage(buf ^ 1);
            if constexpr (!decltype(first)::value) out_stage(f0 - B, buf ^ 1, B, std::true_type{}, slot_tag, false);
        }
        __syncthreads();
    };
    if (arm_wave) {
        if constexpr (DMA) {
            for (int n = 0; n < kLwAhead; n++) dma(size_t(n));
            wait_vmcnt<(kLwAhead - 1) * (B / 8)>();  // batch 0 has landed
        } else if constexpr (LMD) {
            for (int n = 0; n < kLwAhead; n++) dma_lm(size_t(n));
            wait_vmcnt<(kLwAhead - 1) * 4>();  // pair 0 has landed
        } else if (frames >= size_t(B)) {
            fetch(0, std::true_type{});
        } else {
            fetch(0, std::false_type{});
        }
    }
    __syncthreads();  // tables
    if (!arm_wave) lo_stage(0);
    __syncthreads();
    const size_t nfull = frames / B;
    const int tail = int(frames % B);
    using Slot0 = std::integral_constant<int, 0>;
    if constexpr (LM) {
        // whole batches only (launcher): batch n - 1 leaves in iteration n, its group slot (n - 1) % kLwOutGroup is static
        iter(0, B, std::true_type{}, std::true_type{}, Slot0{});
        size_t n = 1;
        while (n < nfull)
            static_for<kLwOutGroup>([&](auto q) {
                if (n < nfull) {
                    iter(n, B, std::true_type{}, std::false_type{}, q);
                    n++;
                }
            });
        if (!arm_wave)
            static_for<kLwOutGroup>([&](auto q) {
                if (int((nfull - 1) % kLwOutGroup) == decltype(q)::value)
                    out_stage((nfull - 1) * B, int((nfull - 1) & 1), B, std::true_type{}, q, true);
            });
    } else {
        if (nfull) {
            iter(0, B, std::true_type{}, std::true_type{}, Slot0{});
            for (size_t n = 1; n < nfull; n++) iter(n, B, std::true_type{}, std::false_type{}, Slot0{});
        }
        if (tail) {
            if (nfull)
                iter(nfull, tail, std::false_type{}, std::false_type{}, Slot0{});
            else
                iter(0, tail, std::false_type{}, std::true_type{}, Slot0{});
            if (!arm_wave) out_stage(nfull * B, int(nfull & 1), tail, std::false_type{}, Slot0{}, true);
        } else if (!arm_wave) {
            out_stage((nfull - 1) * B, int((nfull - 1) & 1), B, std::true_type{}, Slot0{}, true);
        }
    }
    if (active && arm_wave) {
        if (r == 0) st[lane] = acc0 + inc * uint32_t(frames);
        bank.store(st, lanes, lane, 2 + (r ? 2 * N * K : 0));
    }
}

template <int MODE, int N, int K, int IN, int B>
int launch_lockin_waves_in(const LpParams &p, uint32_t *st, const int32_t *x, typename LwOut<MODE>::type *y, size_t lanes,
                           size_t frames, int waves, hipStream_t s)
{
    const dim3 grid(unsigned((lanes + kWave - 1) / kWave));
    note_kernel(waves == 6 ? "lockin_waves_kernel[6 waves per 64 lanes]" : "lockin_waves_kernel[4 waves per 64 lanes]");
    if (waves == 6)
        hipLaunchKernelGGL((lockin_waves_kernel<N, K, 6, IN, MODE, B>), grid, dim3(6 * kWave), 0, s, p, st, x, y, lanes, frames);
    else
        hipLaunchKernelGGL((lockin_waves_kernel<N, K, 4, IN, MODE, B>), grid, dim3(4 * kWave), 0, s, p, st, x, y, lanes, frames);
    return launch_status();
}

template <int MODE, int N, int K>
int launch_lockin_waves_nk(const LpParams &p, void *state, const int32_t *x, void *yv, size_t lanes, size_t frames, int layout,
                           int waves, hipStream_t s)
{
    using Out = typename LwOut<MODE>::type;
    uint32_t *st = static_cast<uint32_t *>(state);
    Out *y = static_cast<Out *>(yv);
    static const bool no_dma = diag_env("IDSP_LOCKIN_NO_DMA") != nullptr;
    // 16-frame batches halve the barriers per frame: 0.37 -> 0.35 ms (Complex<i32>), 0.61 -> 0.59 ms (arg) at 32768 lanes x 4096
    // frames, but 1.06 -> 1.19 ms (arg) at 65536 lanes, where the longer intervals cost more than the barriers
    // (IDSP_LOCKIN_B = 8 / 16 forces one)
    static const int forced_b = [] {
        const char *e = diag_env("IDSP_LOCKIN_B");
        return e ? atoi(e) : 0;
    }();
    const bool b16 = forced_b == 16 || (forced_b != 8 && lanes <= kSplitMaxLanes);
    if (layout == IDSP_LANE_MAJOR) {
        // input by DMA (whole 128-byte lines, each requested once) for the 4-wave I/Q and norm_sqr forms: 0.48 -> 0.43-0.44 ms
        // at 32768 lanes x 4096 frames, 0.94 -> 0.86 at 65536; its 32 KiB ring halves the workgroups a CU can hold, which
        // costs the 6-wave form and the arg read-out more than the input gains (tools/exp_lockin_lm.py)
        if (!no_dma && waves == 4 && MODE != MODE_ARG)
            return launch_lockin_waves_in<MODE, N, K, IN_LM_DMA, kLwBLm>(p, st, x, y, lanes, frames, waves, s);
        return launch_lockin_waves_in<MODE, N, K, IN_LM_REG, kLwBLm>(p, st, x, y, lanes, frames, waves, s);
    }
    if (!no_dma && lanes % kWave == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0) {
        if (b16) return launch_lockin_waves_in<MODE, N, K, IN_FM_DMA, 16>(p, st, x, y, lanes, frames, waves, s);
        return launch_lockin_waves_in<MODE, N, K, IN_FM_DMA, kLwB>(p, st, x, y, lanes, frames, waves, s);
    }
    return launch_lockin_waves_in<MODE, N, K, IN_FM_REG, kLwB>(p, st, x, y, lanes, frames, waves, s);
}

template <int MODE>
int launch_lockin_waves(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, void *y, size_t lanes, size_t frames, int layout,
                        int waves, hipStream_t s)
{
    const LpParams p = lp_params(cfg);
#define IDSP_CASE(N, K) \
    if (cfg->order == N && cfg->cascade == K) return launch_lockin_waves_nk<MODE, N, K>(p, state, x, y, lanes, frames, layout, waves, s)
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
