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
// LaneMajor DMA: ring slots of one PAIR of batches (8 KiB) each, pairs requested ahead.  Three slots keep the workgroup at
// 70 KiB of LDS, so that two of them share a CU (four slots: 78 KiB, measured slower: 0.457 against 0.42-0.44 ms at C4)
constexpr int kLwRing = 3, kLwAhead = 2;
// LaneMajor: batches a read-out thread holds in registers and stores together.  Round 4 (tools/exp_lockin_ablate.hip): with the
// global stores switched off the LaneMajor kernel runs as fast as the FrameMajor one (0.22 ms at C4), with the input requests
// switched off it is the stores that cost (0.38 against 0.265 ms), and they cost nothing extra at a row pitch of 32 KiB + 512 B:
// every lane writes the same offset of its row at the same time, and at a power-of-two pitch those lines meet in the memory
// channels.  Four lines (512 bytes) per lane in one burst: 0.38 -> 0.254 ms without the requests, 0.436 -> 0.357 with them, eight
// 0.348; two, three or six lines per burst change nothing.  (Round 2 had measured "no gain at 4" on the form that left 16-byte pieces.)
#ifndef IDSP_LW_OUT_GROUP
#define IDSP_LW_OUT_GROUP 8
#endif
constexpr int kLwLineGroup = IDSP_LW_OUT_GROUP;  // the whole-line form (8-byte elements, two read-out waves); external LO: 4
constexpr int kLwPieceGroup = 1;                 // the 16-byte-piece form
// The other half of the same finding: workgroups that start together stay in phase, and all of them read line p and write line
// q of their rows at the same time.  A start-up stagger over the CUs of an XCD — workgroup b waits ((b >> 4) % 4) steps of
// `skew` ticks (10 ns each; workgroup b runs on XCD b % 8) — buys more than it costs: 0.357 -> 0.328-0.335 ms at C4 for steps of
// 5-13 us and moduli 3-8, including the 15-40 us the last group waits (shifts 0-2 and 6-8, i.e. phases per XCD or per
// workgroup pair, gain nothing).  Which launches: launch_lockin_waves_in.
constexpr unsigned kLwSkewTicks = thr::kLockinStaggerTicks;
#ifndef IDSP_LW_LM_LINES
#define IDSP_LW_LM_LINES 1  // LaneMajor, 8-byte elements: whole 128-byte lines per store instruction
#endif

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

// Exchange rows.  Everything the waves hand each other per batch is one ROW per lane and component: B words at a pitch
// of B + 4 words (80 or 48 bytes: 16-byte aligned, and 5 resp. 3 bank quartets — odd — so that the 16 lanes of every
// ds_read_b128 group and the 8 lanes of every ds_write_b128 group fall into different banks for any piece).  A thread
// moves its frames of a row as 16-byte vectors: round 2 exchanged single words in [frame][lane] order, 5 LDS
// instructions per lane and frame; this is 1.25.
template <int CNT>
__device__ __forceinline__ void row_load(const int32_t *p, int32_t (&v)[CNT])
{
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    typedef int32_t i32x2 __attribute__((ext_vector_type(2)));
    if constexpr (CNT == 2) {
        const i32x2 a = *reinterpret_cast<const i32x2 *>(p);
        v[0] = a.x, v[1] = a.y;
    } else {
        static_assert(CNT % 4 == 0, "whole 16-byte pieces");
#pragma unroll
        for (int q = 0; q < CNT / 4; q++) {
            const i32x4 a = reinterpret_cast<const i32x4 *>(p)[q];
            v[4 * q] = a.x, v[4 * q + 1] = a.y, v[4 * q + 2] = a.z, v[4 * q + 3] = a.w;
        }
    }
}
template <int CNT>
__device__ __forceinline__ void row_store(int32_t *p, const int32_t (&v)[CNT])
{
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    typedef int32_t i32x2 __attribute__((ext_vector_type(2)));
    if constexpr (CNT == 2) {
        *reinterpret_cast<i32x2 *>(p) = i32x2{v[0], v[1]};
    } else {
#pragma unroll
        for (int q = 0; q < CNT / 4; q++) reinterpret_cast<i32x4 *>(p)[q] = i32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    }
}

// tools/exp_lockin_trace.hip builds this header with IDSP_LW_TRACE: per-wave cycle sums of the phases of an interval
#ifdef IDSP_LW_TRACE
__device__ unsigned long long g_lw_trace[8][8];  // [wave][phase], workgroup 0 only
__device__ unsigned long long g_lw_wg[4096][8][3];  // [workgroup][wave]: s_memrealtime at start, at end, HW_ID | XCC_ID << 32
#define LW_T(i)                                           \
    do {                                                  \
        const unsigned long long t_ = __builtin_readcyclecounter(); \
        tr[i] += t_ - tlast;                              \
        tlast = t_;                                       \
    } while (0)
#else
#define LW_T(i) ((void)0)
#endif

constexpr int kLwFmRing = 5, kLwFmAhead = 4;  // FrameMajor DMA: LDS input ring slots, batches requested ahead

// tools/exp_lockin_ablate.hip: the same kernel without its global stores / without its input requests (conditions that are
// never true at run time, so that the instruction stream stays)
#ifdef IDSP_LW_ABL_NOSTORE
#define LW_ST_ON (frames == 1)
#else
#define LW_ST_ON true
#endif
#ifdef IDSP_LW_ABL_NOLOAD
#define LW_LD_ON (frames == 1)
#else
#define LW_LD_ON true
#endif
#ifdef IDSP_LW_ABL_PLAINSTORE
#define LW_NT_STORE(v, p) (*(p) = (v))
#else
#define LW_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif
#ifdef IDSP_LW_ABL_SKEW
__device__ unsigned g_lw_skew[4096];  // workgroup b starts g_lw_skew[b] ticks of 10 ns late (instead of the `skew` argument's pattern)
#endif

// Bank: the arm filter `C` of `Lockin<C>` (src/lockin.rs:11-15) as a register-resident functor — Params (by value in kernel arguments),
// kArmWords state words per arm, load / store / step; `LpBank<N, K>` (dds_dev.h) for `[Lowpass<N>; K]`, `BqBank<NS>`
// (lockin_waves_biquad.hip) for `[Biquad<Q32<F>>; NS]`.
template <class Bank, int W, int IN, int MODE, int B>
__global__ __launch_bounds__(W * kWave) __attribute__((amdgpu_waves_per_eu(1, IN == IN_LM_REG || IN == IN_LM_DMA ? 2 : 10))) void lockin_waves_kernel(const typename Bank::Params prm, uint32_t *st, const int32_t *x,
                                                                 typename LwOut<MODE>::type *y, const size_t lanes, const size_t frames,
                                                                 const int32_t *lo_ext, const unsigned skew, const size_t pitch)
{
    // `pitch`: elements between LaneMajor rows (x, the LO and y alike) — `frames` for a dense tensor, the call's row length when this
    // launch covers the whole batches of longer rows and a stream kernel takes the rest (round 4; FrameMajor: unused).
    // Bank::kExtLo: the oscillator is not `Accu` -> cossin but a per-sample `Complex` the caller supplies (src/lockin.rs:17-27):
    // lo_ext[index(f, l) * 2 + {re, im}], same layout as x.  The record then has no accumulator words, no table is built, and the
    // read-out waves fetch their share of the next batch's LO one interval ahead into registers.  Bank::mix is the mixer product
    // (i32: the high word, dsp-fixedpoint/src/lib.rs:449-456; f32: one rounded multiply on the bit patterns).
    using Out = typename LwOut<MODE>::type;
    constexpr bool LMD = IN == IN_LM_DMA, LM = IN == IN_LM_REG || LMD, DMA = IN == IN_FM_DMA;
#ifdef IDSP_LW_ABL_SKEW
    if (const long long d = g_lw_skew[blockIdx.x % 4096]) {
#else
    if (const long long d = blockIdx.x < 1024 ? (long long)(skew) * ((blockIdx.x >> 4) & 3u) : 0) {  // the first workgroups only: later rounds start out of phase anyway
#endif
        const long long t0 = wall_clock64();
        // (bounded: an s_sleep(8) is at least 0.2 us = 20 ticks, so d / 8 rounds are more than enough even if the counter stood still)
        for (long long spins = d / 8 + 16; spins > 0 && wall_clock64() - t0 < d; spins--) __builtin_amdgcn_s_sleep(8);
    }
    // Where the mixer multiply `x * lo` (src/lockin.rs:34-37) runs: with the input in LDS (both DMA forms) the read-out waves
    // apply it while they hold cos / sin, and the rows carry the mixed samples — the arm waves are then the two lowpass
    // chains and nothing else (14 VALU instructions per frame for [Lowpass<2>; 2]); with register prefetch the input lives in
    // the arm waves' registers, so the rows carry cos / sin and the arm multiplies.
#ifndef IDSP_LW_MIXR
#define IDSP_LW_MIXR 1
#endif
    constexpr bool MIXR = IDSP_LW_MIXR && (DMA || LMD);
    constexpr int P = W - 2, C = B / P;  // read-out waves; each takes frames r C .. r C + C - 1 of a batch
    constexpr int RS = B + 4;            // row pitch in words
    static_assert(B % P == 0 && B % 8 == 0 && (C == 2 || C % 4 == 0), "batch splits evenly over the read-out waves, into 4-row DMA groups per arm wave, into vectors");
    constexpr bool EXT = Bank::kExtLo;
    constexpr int SB = EXT ? 0 : 2;  // state words before the arms: accu.state, accu.step
    __shared__ __attribute__((aligned(16))) uint32_t ctab[EXT ? 4 : kCosCircleWords];
    __shared__ uint32_t tab[32];
    // rows[buffer][I / Q][lane]: batch n lives in buffer n % 3 — written by the read-out waves (LO or mixed samples) during
    // interval n - 1, turned into the arm outputs IN PLACE by the arm waves during interval n, read back by the read-out waves
    // during interval n + 1
    __shared__ __attribute__((aligned(16))) int32_t rows[3][2][kWave * RS];
    // FM DMA: [slot][frame][lane]; LM DMA: kLwRing slots of 64 lanes x 128 bytes (two batches), rows permuted and pieces swizzled
    __shared__ __attribute__((aligned(16))) int32_t xs[DMA ? kLwFmRing : LMD ? 2 * kLwRing : 1][B * kWave];
    // wave-uniform role, pinned to SGPRs: everything derived from it (row pointers of the output, LDS slots) is scalar
    // Roles by SIMD.  The waves of a workgroup land on the four SIMDs of its CU one each (in the order 0, 2, 1, 3 from a start
    // that moves from workgroup to workgroup: HW_ID dumps of tools/exp_lockin_trace.hip), and with the octant logic of cossin in
    // the table an arm wave issues more than twice what a read-out wave does (~290 against ~125 instructions per 16-frame
    // interval, and its chain is latency-bound on top).  With roles by wave index the arm waves of the two workgroups that share a
    // CU at the C4 lane counts (blockIdx b and b + 256: dispatch order) always met on one SIMD.  So the role follows the SIMD the
    // wave finds itself on: even classes put their arms on SIMDs 0 and 1, odd classes on 2 and 3.  If the four waves are not on
    // four different SIMDs (never observed) the wave index decides, as in the 6-wave form.
#ifndef IDSP_LW_ROT
#define IDSP_LW_ROT 1
#endif
    __shared__ uint32_t wave_simd[4];
    const unsigned prio_class = (blockIdx.x >> 8) & 1u;
    const int lid = int(threadIdx.x) % kWave;
    int w = __builtin_amdgcn_readfirstlane(int(threadIdx.x) / kWave);
    if constexpr (IDSP_LW_ROT && W == 4) {
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const uint32_t simd = (hw >> 4) & 3u;
        if (lid == 0) wave_simd[w] = simd;
        __syncthreads();
        const uint32_t seen = (1u << wave_simd[0]) | (1u << wave_simd[1]) | (1u << wave_simd[2]) | (1u << wave_simd[3]);
        // SIMDs 0 / 2 and 1 / 3 for the arms: with both arm waves on SIMDs 0 and 1 a workgroup alone on its CU ran 7 % slower than
        // with the hardware's own order (IDSP_LW_ROT=2 is that pairing)
        const uint32_t pos = IDSP_LW_ROT == 2 ? simd : ((simd & 1u) << 1) | (simd >> 1);
        if (__builtin_amdgcn_readfirstlane(int(seen)) == 0xF) w = int(pos ^ (prio_class << 1));
    }
    const bool arm_wave = w < 2;
    const int r = arm_wave ? w : w - 2;  // arm waves: I / Q; read-out waves: frame group
    const size_t lane = size_t(blockIdx.x) * kWave + lid;
    const bool active = lane < lanes;
    const size_t la = active ? lane : lanes - 1;  // idle threads of the last workgroup shadow a valid lane, stores masked
    if constexpr (!EXT) fill_cossin_circle(ctab, threadIdx.x, W * kWave);
    if (MODE == MODE_ARG && threadIdx.x < 32) tab[threadIdx.x] = d_atan2_table[threadIdx.x];
    uint32_t acc0 = 0, inc = 0;
    if constexpr (!EXT) acc0 = st[la], inc = st[lanes + la];
    Bank bank;
    if (arm_wave) bank.load(st, lanes, la, SB + (r ? Bank::kArmWords : 0));
    // The state loads must have landed HERE, in a way the compiler's wait-count pass sees: it cannot see the DMA requests, and a
    // first use of a state register inside the steady-state loop would be protected by `s_waitcnt vmcnt(0)` on every interval,
    // draining the input ring each time (lane_stream.h, stream_frame_major_lds).  vmcnt(0), expcnt / lgkmcnt untouched:
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const uint32_t lo32 = uint32_t(la), lane32 = uint32_t(lane);
    uint32_t phase = acc0;  // accumulator before the batch whose LO is produced next
    int32_t xn[DMA || LMD ? 1 : B];
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    auto fetch = [&](size_t f0, auto full) {
        if constexpr (!DMA && !LMD) {
            if constexpr (LM) {
                const i32x4 *row = reinterpret_cast<const i32x4 *>(x + la * pitch + f0);
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
        }
    };
    // FrameMajor DMA.  Arm wave r moves rows r B/2 .. of batch n into ring slot `slot`, 4 rows per instruction: lane l takes
    // the 16 bytes at column (l % 16) * 4 of row l / 16; rows past the end re-read the last frame (never consumed) so that every
    // interval issues exactly B / 8 operations per arm wave
    // (whole groups: wave-uniform row base in SGPRs + this thread's constant 32-bit offset, no per-request address arithmetic)
    const uint32_t dma_off = uint32_t(lid / 16) * uint32_t(lanes) * 4u + uint32_t(lid % 16) * 16u;  // launcher: 3 lanes * 4 < 2^32
    auto dma = [&](size_t n, int slot) {
        if (!LW_LD_ON) return;
#pragma unroll
        for (int g = 0; g < B / 8; g++) {
            const int r0 = r * (B / 2) + 4 * g;  // first of the 4 rows this instruction moves
            const uint32_t dst = uint32_t(reinterpret_cast<uintptr_t>(&xs[slot][r0 * kWave]));
            const size_t row0 = n * B + size_t(r0);
            if (row0 + 4 <= frames) {
                glds16_s(uniform_ptr(x + row0 * lanes + size_t(blockIdx.x) * kWave), dma_off, dst);
            } else {
                size_t row = row0 + size_t(lid / 16);
                row = row < frames ? row : frames - 1;
                glds16(x + row * lanes + size_t(blockIdx.x) * kWave + size_t(lid % 16) * 4, dst);
            }
        }
    };
    // LaneMajor input by DMA: a pair of batches = one whole 128-byte line per lane.  Instruction j of a pair fetches the lines
    // of lanes j, j + 8, ... (8 threads per line) so that every line is requested once, by one instruction; thread t takes
    // piece (t % 8) ^ j, which leaves lane l's piece k at slot row (l % 8) 8 + l / 8, offset 16 (k ^ (l % 8)): a thread's
    // ds_read_b128 of its own row is conflict free.  Arm wave r issues instructions 4 r .. 4 r + 3; pieces past the end of the row
    // (odd number of batches) and pairs past the end re-read frame 0 (never consumed), so that every pair issues exactly four
    // operations per arm wave.
    auto dma_lm = [&](size_t pair) {
        if constexpr (LMD) {
            if (!LW_LD_ON) return;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int j = 4 * r + g;
                size_t gl = size_t(blockIdx.x) * kWave + size_t(j + 8 * (lid / 8));
                gl = gl < lanes ? gl : lanes - 1;
                const size_t f = pair * 32 + size_t(((lid % 8) ^ j) * 4);
                glds16(x + gl * pitch + (f + 4 <= frames ? f : 0),
                       uint32_t(reinterpret_cast<uintptr_t>(&xs[0][0])) + uint32_t((pair % kLwRing) * 8192 + j * 1024));
            }
        }
    };
    const uint32_t xs_own = uint32_t((lid % 8) * 8 + lid / 8) * 128 + uint32_t(lid % 8) * 16;  // LM DMA: own row, piece k at ^ 16 k
#ifdef IDSP_LW_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    const unsigned long long t_begin = tlast, rt_begin = __builtin_amdgcn_s_memrealtime();
#endif
    int slot_issue = 0, slot_read = 0;  // FM DMA ring positions of the next request (arm waves) / the next batch mixed (read-out waves)

    // external LO: frames r C .. r C + C - 1 of batch n for this thread's lane, one interval ahead (frames past the end: zero)
    typedef int32_t i32x2 __attribute__((ext_vector_type(2)));
    i32x2 lon[EXT ? C : 1];
    auto lo_fetch = [&](size_t n) {
        if constexpr (EXT) {
            const i32x2 *l2 = reinterpret_cast<const i32x2 *>(lo_ext);
#pragma unroll
            for (int j = 0; j < C; j++) {
                const size_t f = n * B + size_t(r * C + j);
                lon[j] = f < frames ? (LM ? l2[la * pitch + f] : l2[f * lanes + la]) : i32x2{0, 0};
            }
        }
    };
    // ---- read-out waves, first half of an interval: cos / sin (and, with the input in LDS, the mixer) of batch n into `dstb`
    auto lo_stage = [&](size_t n, int dstb) {
        int32_t re[C], im[C], xv[C];
        if constexpr (!MIXR) {
        } else if constexpr (DMA) {
#pragma unroll
            for (int j = 0; j < C; j++) xv[j] = xs[slot_read][(r * C + j) * kWave + lid];
            slot_read = slot_read + 1 == kLwFmRing ? 0 : slot_read + 1;
        } else if constexpr (LMD) {
            static_assert(!LMD || (B == 16 && C % 4 == 0), "a pair of batches is one 128-byte line per lane; a read-out thread takes whole pieces");
            const char *slot = reinterpret_cast<const char *>(&xs[0][0]) + ((n / 2) % kLwRing) * 8192;
#pragma unroll
            for (int v = 0; v < C / 4; v++) {
                const i32x4 a = *reinterpret_cast<const i32x4 *>(slot + (xs_own ^ uint32_t(((n % 2) * 4 + (r * C) / 4 + v) * 16)));
                xv[4 * v] = a.x, xv[4 * v + 1] = a.y, xv[4 * v + 2] = a.z, xv[4 * v + 3] = a.w;
            }
        }
        LW_T(0);
        if constexpr (EXT) {
#pragma unroll
            for (int j = 0; j < C; j++) {
                re[j] = MIXR ? Bank::mix(lon[j].x, xv[j]) : lon[j].x;
                im[j] = MIXR ? Bank::mix(lon[j].y, xv[j]) : lon[j].y;
            }
            lo_fetch(n + 1);  // this wave's share of the batch after: in flight during the interval
        } else {
            // all table reads of the batch first, behind one wait (see lockin_stages_kernel)
            uint32_t pj[C];
            CosCircleEntry ent[C];
#pragma unroll
            for (int j = 0; j < C; j++) {
                pj[j] = phase + inc * uint32_t(r * C + j + 1);
                ent[j] = cossin_circle_fetch(pj[j], ctab);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < C; j++) {
                const Cplx lo = cossin_circle_finish(pj[j], ent[j]);
                re[j] = MIXR ? Bank::mix(lo.re, xv[j]) : lo.re;
                im[j] = MIXR ? Bank::mix(lo.im, xv[j]) : lo.im;
            }
            phase += inc * uint32_t(B);
        }
        LW_T(1);
        row_store<C>(&rows[dstb][0][lid * RS + r * C], re);
        row_store<C>(&rows[dstb][1][lid * RS + r * C], im);
        LW_T(2);
    };
    auto element = [&](int32_t re, int32_t im) -> Out {
        if constexpr (MODE == MODE_IQ)
            return Cplx{re, im};
        else if constexpr (MODE == MODE_ARG)
            return atan2_dev(im, re, tab);
        else
            return int64_t(uint64_t(int64_t(re) * re) + uint64_t(int64_t(im) * im));  // wraps for (MIN, MIN) as in release
    };
    // LaneMajor: a read-out thread keeps its pieces of kLwOutGroup batches in registers and stores them together
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    constexpr int GV = int(sizeof(Out)) * C / 16;  // 16-byte vectors a read-out thread writes per batch (LaneMajor)
    constexpr bool LINES = LM && sizeof(Out) == 8 && P == 2 && B == 16 && IDSP_LW_LM_LINES;
    // (eight lines: 0.334 -> 0.325 ms at C4 with 224 registers; the external-LO forms also hold the next batch's LO and keep four)
    constexpr int kLwOutGroup = !LINES ? kLwPieceGroup : Bank::kExtLo && kLwLineGroup > 4 ? 4 : kLwLineGroup;
    u32x4 held[LM ? kLwOutGroup : 1][LM ? GV : 1];
    // ---- read-out waves, second half: the arm outputs of the batch that starts at frame f0 (buffer srcb) become output elements.
    // `slot` (static): position of the batch inside its output group; `flush`: last batch of the call
    auto out_stage = [&](size_t f0, int srcb, int nb, auto full, auto slot_tag, bool flush) {
        if constexpr (LINES) {
            // Whole 128-byte lines per store instruction (round 3, later): a batch of 8-byte elements is one line per lane.  Read-out
            // wave r takes lanes 32 r .. 32 r + 31; in instruction v thread t holds frames 2 (t % 8), 2 (t % 8) + 1 of lane
            // 32 r + 8 v + t / 8, so the 8 threads of a lane write its line and one instruction writes 8 whole lines (the form below
            // leaves two 16-byte pieces in each of 32 lines per instruction).  The arm outputs come out of the rows as 8-byte reads.
            constexpr int slot = decltype(slot_tag)::value;
            const int piece = lid % 8;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int ll = r * 32 + v * 8 + lid / 8;
                int32_t re[2], im[2];
                row_load<2>(&rows[srcb][0][ll * RS + 2 * piece], re);
                row_load<2>(&rows[srcb][1][ll * RS + 2 * piece], im);
                const uint64_t u0 = __builtin_bit_cast(uint64_t, element(re[0], im[0])), u1 = __builtin_bit_cast(uint64_t, element(re[1], im[1]));
                held[slot][v] = u32x4{uint32_t(u0), uint32_t(u0 >> 32), uint32_t(u1), uint32_t(u1 >> 32)};
            }
            if (slot == kLwOutGroup - 1 || flush) {
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const size_t gl = size_t(blockIdx.x) * kWave + size_t(r * 32 + v * 8 + lid / 8);
                    u32x4 *dst = reinterpret_cast<u32x4 *>(y + gl * pitch + (f0 - size_t(slot) * B)) + piece;
#pragma unroll
                    for (int q = 0; q <= slot; q++)
                        if (gl < lanes && LW_ST_ON) LW_NT_STORE(held[q][v], dst + q * 8);
                }
            }
        } else if constexpr (LM) {
            // read-out wave r writes lanes r * 64 / P ..: P adjacent threads cover the B frames of one lane, so that one store
            // instruction leaves B * sizeof(Out) contiguous bytes per lane
            static_assert(!LM || (sizeof(Out) * C) % 16 == 0, "a thread's piece of a batch is whole 16-byte vectors");
            const int ll = r * (kWave / P) + lid / P, part = lid % P;
            const size_t gl = size_t(blockIdx.x) * kWave + size_t(ll);
            constexpr int slot = decltype(slot_tag)::value;
            constexpr int OW = int(sizeof(Out)) / 4;
            int32_t re[C], im[C];
            row_load<C>(&rows[srcb][0][ll * RS + part * C], re);
            row_load<C>(&rows[srcb][1][ll * RS + part * C], im);
            uint32_t wd[C * OW];
#pragma unroll
            for (int j = 0; j < C; j++) {
                const Out e = element(re[j], im[j]);
                if constexpr (OW == 1) {
                    wd[j] = __builtin_bit_cast(uint32_t, e);
                } else {
                    const uint64_t u = __builtin_bit_cast(uint64_t, e);
                    wd[2 * j] = uint32_t(u), wd[2 * j + 1] = uint32_t(u >> 32);
                }
            }
#pragma unroll
            for (int v = 0; v < GV; v++) held[slot][v] = u32x4{wd[4 * v], wd[4 * v + 1], wd[4 * v + 2], wd[4 * v + 3]};
            if (slot == kLwOutGroup - 1 || flush) {
                u32x4 *dst = reinterpret_cast<u32x4 *>(y + gl * pitch + (f0 - size_t(slot) * B) + part * C);
#pragma unroll
                for (int q = 0; q <= slot; q++)
                    if (gl < lanes) {
#pragma unroll
                        for (int v = 0; v < GV; v++) dst[q * (B * int(sizeof(Out)) / 16) + v] = held[q][v];
                    }
            }
        } else {
            int32_t re[C], im[C];
            row_load<C>(&rows[srcb][0][lid * RS + r * C], re);
            row_load<C>(&rows[srcb][1][lid * RS + r * C], im);
            LW_T(3);
            // wave-uniform row base (SGPRs) + this thread's 32-bit byte offset (launcher: lanes * sizeof(Out) < 2^32)
            char *row = reinterpret_cast<char *>(uniform_ptr(y + (f0 + size_t(r * C)) * lanes));
            const uint32_t off = lane32 * uint32_t(sizeof(Out));
#pragma unroll
            for (int j = 0; j < C; j++) {
                if ((decltype(full)::value || r * C + j < nb) && active && LW_ST_ON)
                    nt_store<true>(reinterpret_cast<Out *>(row + size_t(off)), element(re[j], im[j]));  // 8-byte elements leave as one 2-word vector
                row += lanes * sizeof(Out);
            }
            LW_T(4);
        }
    };
    // ---- arm waves: batch n (nb frames of it) through this arm's lowpass chain, in place in buffer `buf`
    auto arm_stage = [&](size_t n, int buf, int nb, auto full) {
        const size_t f0 = n * B;
        int32_t xv[MIXR ? 1 : B];
        if constexpr (DMA) {
            dma(n + kLwFmAhead, slot_issue);
            slot_issue = slot_issue + 1 == kLwFmRing ? 0 : slot_issue + 1;
            if constexpr (!MIXR) {
#pragma unroll
                for (int b = 0; b < B; b++) xv[b] = xs[slot_read][b * kWave + lid];
                slot_read = slot_read + 1 == kLwFmRing ? 0 : slot_read + 1;
            }
        } else if constexpr (LMD) {
            if (n % 2 == 0) dma_lm(n / 2 + kLwAhead);
            if constexpr (!MIXR) {
                const char *slot = reinterpret_cast<const char *>(&xs[0][0]) + ((n / 2) % kLwRing) * 8192;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const i32x4 a = *reinterpret_cast<const i32x4 *>(slot + (xs_own ^ uint32_t(((n % 2) * 4 + v) * 16)));
                    xv[4 * v] = a.x, xv[4 * v + 1] = a.y, xv[4 * v + 2] = a.z, xv[4 * v + 3] = a.w;
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < B; b++) xv[b] = xn[b];
            if (f0 + 2 * B <= frames)
                fetch(f0 + B, std::true_type{});
            else if (f0 + B < frames)
                fetch(f0 + B, std::false_type{});
        }
        LW_T(0);
        int32_t v[B];
        int32_t *row = &rows[buf][r][lid * RS];
        row_load<B>(row, v);
        LW_T(1);
#pragma unroll
        for (int b = 0; b < B; b++) {
            if (decltype(full)::value || b < nb) {
                if constexpr (MIXR)
                    v[b] = bank.step(prm, v[b]);
                else
                    v[b] = bank.step(prm, Bank::mix(v[b], xv[b]));
            }
        }
        LW_T(2);
        row_store<B>(row, v);
        LW_T(3);
        // the read-out waves mixed batch n + 1 during this interval; the next interval needs batch n + 2 in LDS
        // (mixer in the read-out waves: they need batch n + 2 during the next interval; in the arm waves: batch n + 1)
        constexpr int kEarly = MIXR ? 2 : 1;
        if constexpr (DMA) wait_vmcnt<(kLwFmAhead - kEarly) * (B / 8)>();  // batches up to n + kEarly have landed, the later ones may be in flight
        if constexpr (LMD) {
            // mixer in the arm waves: batch n + 1 is read during interval n + 1, so at the end of an ODD interval the pair of batch n + 1
            // must have landed and the youngest request (issued one interval ago) may still fly.  Mixer in the read-out waves: batch
            // n + 2 is read during interval n + 1, so the same holds at the end of an EVEN interval, the request issued in this very
            // interval still flying.  (The first version of the read-out mixer waited for vmcnt(0) at odd intervals — every pair only one
            // interval after its request — and exposed the DMA latency.)
            if constexpr (MIXR) {
                if (n % 2 == 0) wait_vmcnt<4>();
            } else {
                if (n % 2 == 1) wait_vmcnt<(kLwAhead - 1) * 4>();
            }
        }
        LW_T(4);
    };
    if (arm_wave) {
        if constexpr (DMA) {
            for (int n = 0; n < kLwFmAhead; n++) {
                dma(size_t(n), slot_issue);
                slot_issue++;
            }
            static_assert(kLwFmAhead < kLwFmRing, "the ring holds the batch being mixed and the ones in flight");
            wait_vmcnt<(kLwFmAhead - (MIXR ? 2 : 1)) * (B / 8)>();  // batch 0 (and 1, if the read-out waves mix) has landed
        } else if constexpr (LMD) {
            for (int n = 0; n < kLwAhead; n++) dma_lm(size_t(n));
            wait_vmcnt<(kLwAhead - (MIXR ? 2 : 1)) * 4>();  // pair 0 (and 1) has landed
        } else if (frames >= size_t(B)) {
            fetch(0, std::true_type{});
        } else {
            fetch(0, std::false_type{});
        }
    }
    __syncthreads();  // tables, first input batches
    if (!arm_wave) lo_fetch(0);
    if (!arm_wave) lo_stage(0, 0);
    __syncthreads();
    const size_t nfull = frames / B;
    const int tail = int(frames % B);
    const size_t nbatch = nfull + (tail ? 1 : 0);
    using Slot0 = std::integral_constant<int, 0>;
    int cur = 0;  // n % 3
    auto next3 = [](int b) { return b == 2 ? 0 : b + 1; };
    auto prev3 = [](int b) { return b == 0 ? 2 : b - 1; };
    // interval n: arm waves batch n; read-out waves LO / mixer of batch n + 1, then the output of batch n - 1
    // Two workgroups share a CU at the C4 lane counts, and VALU issue is arbitrated by priority, then AGE: the older workgroup's
    // waves win every conflict, it runs at nearly its solo speed and the younger one crawls until the older has finished —
    // per-workgroup end times of 285 and 415 us on every CU (tools/exp_lockin_trace.hip), i.e. a third of the launch with half
    // of each CU idle.  The two swap priority every interval instead (the co-resident workgroups of a CU are blockIdx b and
    // b + 256: dispatch order), so that both advance at the same average rate.
#ifndef IDSP_LW_PRIO
#define IDSP_LW_PRIO 3
#endif
    // The two roles run the intervals in loops of their own (round 4): inside one loop the arm state and the read-out waves' held
    // output vectors are all loop-carried values of the same thread and their registers add up; in two loops they overlay.
    auto interval = [&](size_t n, int nb, auto full, auto slot_tag, auto arm_tag) {
#ifdef IDSP_LW_PRIO_BY_ROLE  // experiment: a fixed priority per role (arm waves IDSP_LW_PRIO_BY_ROLE, read-out waves 0) instead of the alternation
        if (n == 0) {
            if (arm_wave)
                __builtin_amdgcn_s_setprio(IDSP_LW_PRIO_BY_ROLE);
            else
                __builtin_amdgcn_s_setprio(0);
        }
#else
        if constexpr (IDSP_LW_PRIO != 0) {
            if ((unsigned(n) ^ prio_class) & 1u)
                __builtin_amdgcn_s_setprio(IDSP_LW_PRIO);
            else
                __builtin_amdgcn_s_setprio(0);
        }
#endif
        if constexpr (decltype(arm_tag)::value) {
            arm_stage(n, cur, nb, full);
        } else {
            if (n + 1 < nbatch) lo_stage(n + 1, next3(cur));
            if (n > 0) out_stage((n - 1) * B, prev3(cur), B, std::true_type{}, slot_tag, false);
        }
        __syncthreads();
        LW_T(5);
        cur = next3(cur);
    };
    auto run_role = [&](auto arm_tag) {
        constexpr bool ARM = decltype(arm_tag)::value;
        if constexpr (LM && ARM) {
            for (size_t n = 0; n < nfull; n++) interval(n, B, std::true_type{}, Slot0{}, arm_tag);  // whole batches only (launcher)
        } else if constexpr (LM) {
            // batch n - 1 leaves in interval n, its group slot (n - 1) % kLwOutGroup is static
            interval(0, B, std::true_type{}, Slot0{}, arm_tag);
            size_t n = 1;
            while (n < nfull)
                static_for<kLwOutGroup>([&](auto q) {
                    if (n < nfull) {
                        interval(n, B, std::true_type{}, q, arm_tag);
                        n++;
                    }
                });
            if constexpr (!ARM)
                static_for<kLwOutGroup>([&](auto q) {
                    if (int((nfull - 1) % kLwOutGroup) == decltype(q)::value) out_stage((nfull - 1) * B, prev3(cur), B, std::true_type{}, q, true);
                });
        } else {
            for (size_t n = 0; n < nfull; n++) interval(n, B, std::true_type{}, Slot0{}, arm_tag);
            if (tail) {
                interval(nfull, tail, std::false_type{}, Slot0{}, arm_tag);
                if constexpr (!ARM) out_stage(nfull * B, prev3(cur), tail, std::false_type{}, Slot0{}, true);
            } else if constexpr (!ARM) {
                out_stage((nfull - 1) * B, prev3(cur), B, std::true_type{}, Slot0{}, true);
            }
        }
    };
    if (arm_wave)
        run_role(std::true_type{});
    else
        run_role(std::false_type{});
#ifdef IDSP_LW_TRACE
    if (lid == 0 && blockIdx.x < 4096) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_lw_wg[blockIdx.x][w][0] = rt_begin, g_lw_wg[blockIdx.x][w][1] = __builtin_amdgcn_s_memrealtime();
        g_lw_wg[blockIdx.x][w][2] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
    }
    if (blockIdx.x == 0 && lid == 0) {
        for (int i = 0; i < 6; i++) g_lw_trace[w][i] = tr[i];
        g_lw_trace[w][6] = __builtin_readcyclecounter() - t_begin;       // s_memtime ticks of the whole kernel
        g_lw_trace[w][7] = __builtin_amdgcn_s_memrealtime() - rt_begin;  // 100 MHz ticks of the whole kernel
    }
#endif
    if (active && arm_wave) {
        if constexpr (!EXT) {
            if (r == 0) st[lane] = acc0 + inc * uint32_t(frames);
        }
        bank.store(st, lanes, lane, SB + (r ? Bank::kArmWords : 0));
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// One wave per cascade STAGE (round 3, later).  What bounds lockin_waves_kernel at the C4 shape is not the number of instructions
// but how fast ONE wave issues a dependent chain: an arm wave runs `[Lowpass<N>; 2]` at ~10 cycles per instruction (a SIMD
// could issue one every 4-5 with three or four waves to pick from), two workgroups per CU are two waves per SIMD, and an
// interval of 16 frames takes 1.4 us where the instruction count needs 0.8.  Here the two stages of each arm are two waves one
// batch apart (stage 1 of batch n - 1 beside stage 0 of batch n: half the chain per wave), the read-out waves also apply the
// mixer, R = 4 read-out waves per lane group take four frames of a batch each, and G = 1 or 2 lane groups of 64 share one workgroup,
// its cossin table and its barrier (two 6- or 8-wave workgroups do not share a CU on this hardware whatever the occupancy API says:
// the second starts when the first has ended).  With G = 2: 8 arm-stage waves + 8 read-out waves on a CU, four per SIMD — by the
// hardware's round-robin placement every SIMD gets two arm-stage waves (~135 instructions per interval each) and two read-out waves.  A batch lives in one of three
// row buffers: mixed by the read-out waves during interval n - 1, stage 0 in place during n, stage 1 in place during n + 1,
// turned into output elements during n + 2 (and overwritten with batch n + 3 by the wave that just read it).  Input by LDS-DMA as in the kernel above: every arm-stage wave moves 4 of the 16 rows
// of a batch, four batches ahead.  FrameMajor, whole 16-frame batches, whole 64 G-lane blocks, K = 2 (the launcher falls back to
// the kernel above for everything else).
#ifndef IDSP_LS_RING
#define IDSP_LS_RING 5
#endif
constexpr int kLsB = 16, kLsRing = IDSP_LS_RING, kLsAhead = IDSP_LS_RING - 1;  // input ring slots / batches requested ahead

template <int N, int MODE, int G, int R = 2>
__global__ __launch_bounds__((4 + R) * G *kWave) void lockin_stages_kernel(const LpParams prm, uint32_t *st, const int32_t *x,
                                                                      typename LwOut<MODE>::type *y, const size_t lanes, const size_t frames)
{
    using Out = typename LwOut<MODE>::type;
    constexpr int B = kLsB, RS = B + 4, C = B / R, W = (4 + R) * G;  // R read-out waves per lane group, C frames of a batch each
    __shared__ __attribute__((aligned(16))) uint32_t ctab[kCosCircleWords];
    __shared__ uint32_t tab[32];
    __shared__ __attribute__((aligned(16))) int32_t rows[G][3][2][kWave * RS];
    __shared__ __attribute__((aligned(16))) int32_t xs[G][kLsRing][B * kWave];
    const int wv = __builtin_amdgcn_readfirstlane(int(threadIdx.x) / kWave), lid = int(threadIdx.x) % kWave;
    // arm-stage waves: wv = 4 g + 2 r + k (lane group, I / Q, stage); read-out waves: wv = 4 G + R g + h (lane group, part of the batch)
    const bool arm_wave = wv < 4 * G;
    const int g = arm_wave ? wv >> 2 : (wv - 4 * G) / R;
    const int r = (wv >> 1) & 1, k = wv & 1, q = wv & 3, h = arm_wave ? 0 : (wv - 4 * G) % R;
    const size_t lane0 = (size_t(blockIdx.x) * G + size_t(g)) * kWave, lane = lane0 + size_t(lid);
    fill_cossin_circle(ctab, threadIdx.x, W * kWave);
    if (MODE == MODE_ARG && threadIdx.x < 32) tab[threadIdx.x] = d_atan2_table[threadIdx.x];
    int64_t s[N];
    uint32_t acc0 = 0, inc = 0;
    const int word0 = 2 + r * 4 * N + k * 2 * N;  // this stage's LowpassState<N> inside [acc, step, I arm [K][N] i64, Q arm]
    if (arm_wave) {
#pragma unroll
        for (int j = 0; j < N; j++)
            s[j] = int64_t(uint64_t(st[size_t(word0 + 2 * j) * lanes + lane]) | (uint64_t(st[size_t(word0 + 2 * j + 1) * lanes + lane]) << 32));
    } else {
        acc0 = st[lane], inc = st[lanes + lane];
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see lockin_waves_kernel
    const uint32_t lane32 = uint32_t(lane);
    // read-out wave h evaluates the LO of frames C h + 1 .. C h + C past the accumulator; `ph` = accumulator + C h steps
    uint32_t ph = acc0 + inc * uint32_t(h * C);
    const uint32_t inc_rest = inc * uint32_t(B - C);
    const uint32_t dma_off = uint32_t(lid / 16) * uint32_t(lanes) * 4u + uint32_t(lid % 16) * 16u;
    int slot_issue = 0, slot_read = 0;
    auto dma = [&](size_t n) {
        const int r0 = 4 * q;  // this wave's 4 rows of the batch
        const uint32_t dst = uint32_t(reinterpret_cast<uintptr_t>(&xs[g][slot_issue][r0 * kWave]));
        const size_t row0 = n * B + size_t(r0);
        if (row0 + 4 <= frames) {
            glds16_s(uniform_ptr(x + row0 * lanes + lane0), dma_off, dst);
        } else {  // past the end: never consumed, but every interval issues exactly one request per arm-stage wave
            size_t row = row0 + size_t(lid / 16);
            row = row < frames ? row : frames - 1;
            glds16(x + row * lanes + lane0 + size_t(lid % 16) * 4, dst);
        }
        slot_issue = slot_issue + 1 == kLsRing ? 0 : slot_issue + 1;
    };
    auto lo_stage = [&](int dstb) {
        int32_t xv[C], re[C], im[C];
#pragma unroll
        for (int j = 0; j < C; j++) xv[j] = xs[g][slot_read][(h * C + j) * kWave + lid];
        slot_read = slot_read + 1 == kLsRing ? 0 : slot_read + 1;
        // all table reads of the batch first, behind one wait: an evaluation is two LDS reads and nine VALU instructions, and
        // evaluated one after the other each pays the LDS round trip
        uint32_t pj[C];
        CosCircleEntry ent[C];
#pragma unroll
        for (int j = 0; j < C; j++) {
            ph += inc;
            pj[j] = ph;
            ent[j] = cossin_circle_fetch(ph, ctab);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < C; j++) {
            const Cplx lo = cossin_circle_finish(pj[j], ent[j]);
            re[j] = __mulhi(lo.re, xv[j]);  // src/lockin.rs:34-37
            im[j] = __mulhi(lo.im, xv[j]);
        }
        ph += inc_rest;
        row_store<C>(&rows[g][dstb][0][lid * RS + h * C], re);
        row_store<C>(&rows[g][dstb][1][lid * RS + h * C], im);
    };
    auto element = [&](int32_t re, int32_t im) -> Out {
        if constexpr (MODE == MODE_IQ)
            return Cplx{re, im};
        else if constexpr (MODE == MODE_ARG)
            return atan2_dev(im, re, tab);
        else
            return int64_t(uint64_t(int64_t(re) * re) + uint64_t(int64_t(im) * im));
    };
    auto out_stage = [&](size_t f0, int srcb) {
        int32_t re[C], im[C];
        row_load<C>(&rows[g][srcb][0][lid * RS + h * C], re);
        row_load<C>(&rows[g][srcb][1][lid * RS + h * C], im);
        char *row = reinterpret_cast<char *>(uniform_ptr(y + (f0 + size_t(h * C)) * lanes));
        const uint32_t off = lane32 * uint32_t(sizeof(Out));
#pragma unroll
        for (int j = 0; j < C; j++) {
            nt_store<true>(reinterpret_cast<Out *>(row + size_t(off)), element(re[j], im[j]));
            row += lanes * sizeof(Out);
        }
    };
    auto arm_stage = [&](int buf) {
        int32_t v[B];
        int32_t *row = &rows[g][buf][r][lid * RS];
        row_load<B>(row, v);
#pragma unroll
        for (int b = 0; b < B; b++) v[b] = lowpass_step<N>(prm.k[k], s, v[b]);
        row_store<B>(row, v);
    };
    const size_t nb = frames / B;
    if (arm_wave) {
        for (int n = 0; n < kLsAhead; n++) dma(size_t(n));
        wait_vmcnt<kLsAhead - 2>();  // batches 0 and 1 have landed
    }
    __syncthreads();  // tables, first input batches
    if (!arm_wave) lo_stage(0);
    __syncthreads();
    // Three row buffers: batch n sits in buffer n % 3.  A read-out wave turns ITS eight frames of batch n - 2 into output elements
    // and then writes the mixed samples of batch n + 1 over exactly those words (same buffer, no other wave touches them).
    int cur = 0;  // n % 3
    auto next3 = [](int b) { return b == 2 ? 0 : b + 1; };
    auto prev3 = [](int b) { return b == 0 ? 2 : b - 1; };
    for (size_t n = 0; n < nb + 2; n++) {
        if (arm_wave) {
            dma(n + kLsAhead);
            if (k == 0) {
                if (n < nb) arm_stage(cur);
            } else {
                if (n >= 1 && n <= nb) arm_stage(prev3(cur));
            }
            wait_vmcnt<kLsAhead - 2>();  // batches up to n + 2 have landed; the read-out waves mix batch n + 2 during the next interval
        } else {
            if (n >= 2) out_stage((n - 2) * B, next3(cur));
            if (n + 1 < nb) lo_stage(next3(cur));
        }
        __syncthreads();
        cur = next3(cur);
    }
    if (arm_wave) {
#pragma unroll
        for (int j = 0; j < N; j++) {
            st[size_t(word0 + 2 * j) * lanes + lane] = uint32_t(uint64_t(s[j]));
            st[size_t(word0 + 2 * j + 1) * lanes + lane] = uint32_t(uint64_t(s[j]) >> 32);
        }
    } else if (h == 0) {
        st[lane] = acc0 + inc * uint32_t(frames);
    }
}

// Which launches the stage-wave kernel takes (0 = none, else lane groups per workgroup): whole 64-lane groups, whole batches,
// aligned rows, K = 2.  Measured at 4096 frames against the 4- / 6-wave kernel with the same table cossin and roles by SIMD
// (tools/exp_lockin_stages.hip, profiles/r03_exp_lockin_stages*.jsonl): `Complex<i32>` / `norm_sqr` read-out 0.207 against 0.232 ms
// up to 16384 lanes (one workgroup per CU or fewer: eight waves per 64 lanes instead of four) but 0.345 against 0.317 at 32768 and
// 0.64 against 0.57 at 49152, so those take it up to 16384 lanes; the `arg` read-out (atan2 on the read-out waves) 0.337 / 0.465 /
// 0.93 against 0.384 / 0.51 / 1.16 ms at 16384 / 32768 / 65536 lanes (16-byte table entries: 0.311 against 0.289 of the HBM peak from 65536 to
// 196608 lanes), so it takes it at every lane count.
// IDSP_DIAG switches: IDSP_LOCKIN_NO_STAGES=1 never, IDSP_LOCKIN_STAGE_GROUPS=1 / 2 always (when the shape allows).
inline int lockin_stage_groups(const void *x, size_t lanes, size_t frames, int layout, int cascade, bool heavy_readout)
{
    static const bool off = diag_env("IDSP_LOCKIN_NO_STAGES") != nullptr;
    static const int forced = [] {
        const char *e = diag_env("IDSP_LOCKIN_STAGE_GROUPS");
        return e ? atoi(e) : 0;
    }();
    if (off || layout != IDSP_FRAME_MAJOR || cascade != 2 || frames == 0 || frames % kLsB != 0 || lanes == 0 || lanes % kWave != 0 ||
        reinterpret_cast<uintptr_t>(x) % 16 != 0)
        return 0;
    const bool pairs = lanes % (2 * kWave) == 0;
    if (forced == 1 || (forced == 2 && pairs)) return forced;
    if (lanes <= 16384) return 1;
    return heavy_readout && pairs ? 2 : 0;
}

template <int MODE, int N>
int launch_lockin_stages(const LpParams &p, void *state, const int32_t *x, void *yv, size_t lanes, size_t frames, int groups, hipStream_t s)
{
    using Out = typename LwOut<MODE>::type;
    uint32_t *st = static_cast<uint32_t *>(state);
    Out *y = static_cast<Out *>(yv);
    if (groups == 2) {
        note_kernel("lockin_stages_kernel[16 waves per 128 lanes]");
        hipLaunchKernelGGL((lockin_stages_kernel<N, MODE, 2, 4>), dim3(unsigned(lanes / (2 * kWave))), dim3(16 * kWave), 0, s, p, st, x, y, lanes, frames);
    } else {
        note_kernel("lockin_stages_kernel[8 waves per 64 lanes]");
        hipLaunchKernelGGL((lockin_stages_kernel<N, MODE, 1, 4>), dim3(unsigned(lanes / kWave)), dim3(8 * kWave), 0, s, p, st, x, y, lanes, frames);
    }
    return launch_status();
}

template <int MODE, class Bank, int IN, int B>
int launch_lockin_waves_in(const typename Bank::Params &p, uint32_t *st, const int32_t *x, typename LwOut<MODE>::type *y, size_t lanes,
                           size_t frames, int waves, hipStream_t s, const int32_t *lo, size_t pitch)
{
    const dim3 grid(unsigned((lanes + kWave - 1) / kWave));
    if (pitch == 0) pitch = frames;
    // start-up stagger (see kLwSkewTicks): LaneMajor launches that fill the chip, long enough for the wait to pay
    // (IDSP_DIAG=1 IDSP_LOCKIN_NO_SKEW=1: none)
    static const bool no_skew = diag_env("IDSP_LOCKIN_NO_SKEW") != nullptr;
    // (one workgroup per CU: 16384 x 4096 0.217 -> 0.225 ms with it; 32768 lanes x 2048 frames 0.187 -> 0.179, x 4096 0.350 -> 0.323,
    // x 16384 1.335 -> 1.358: long calls drift apart by themselves; 65536 x 4096 0.696 -> 0.640)
    const unsigned skew = (IN == IN_LM_REG || IN == IN_LM_DMA) && !no_skew && grid.x >= thr::kStaggerMinWorkgroups && frames >= thr::kStaggerMinFrames && frames <= thr::kStaggerMaxFrames && stagger_tuned_device() ? kLwSkewTicks : 0u;
    note_kernel(waves == 6 ? "lockin_waves_kernel[6 waves per 64 lanes]" : "lockin_waves_kernel[4 waves per 64 lanes]", Bank::name());
    if constexpr (Bank::kSixWaves) {
        if (waves == 6) {
            hipLaunchKernelGGL((lockin_waves_kernel<Bank, 6, IN, MODE, B>), grid, dim3(6 * kWave), 0, s, p, st, x, y, lanes, frames, lo, skew, pitch);
            return launch_status();
        }
    }
    hipLaunchKernelGGL((lockin_waves_kernel<Bank, 4, IN, MODE, B>), grid, dim3(4 * kWave), 0, s, p, st, x, y, lanes, frames, lo, skew, pitch);
    return launch_status();
}

// batch length and input form for a bank on the multi-wave kernel (`waves` from lockin_waves_for, dds.hip)
template <int MODE, class Bank>
int launch_lockin_waves_bank(const typename Bank::Params &p, uint32_t *st, const int32_t *x, typename LwOut<MODE>::type *y, size_t lanes,
                             size_t frames, int layout, int waves, hipStream_t s, const int32_t *lo = nullptr, size_t pitch = 0)
{
    static const bool no_dma = diag_env("IDSP_LOCKIN_NO_DMA") != nullptr;
    // 16-frame batches halve the barriers per frame: 0.37 -> 0.35 ms (Complex<i32>), 0.61 -> 0.59 ms (arg) at 32768 lanes x 4096
    // frames, but 1.06 -> 1.19 ms (arg) at 65536 lanes, where the longer intervals cost more than the barriers
    // (IDSP_LOCKIN_B = 8 / 16 forces one)
    static const int forced_b = [] {
        const char *e = diag_env("IDSP_LOCKIN_B");
        return e ? atoi(e) : 0;
    }();
    // the 6-wave form keeps 8-frame batches: with 16 (66 KiB of LDS) only ONE 6-wave workgroup runs on a CU at a time — every
    // second workgroup started after the first had finished (tools/exp_lockin_trace.hip) although the occupancy API promises two
    // Four workgroups per CU (49152 < lanes <= 65536) also take 16-frame batches: two co-resident workgroups in two rounds, where the 8-frame
    // form's three leave a round of one (0.649 against 0.581 of the HBM peak at 65536 lanes; 49152: 0.557 against 0.654, 98304: 0.649 against 0.666,
    // profiles/r03_exp_lockin_batch_v2.jsonl)
    const bool b16 = forced_b == 16 || (forced_b != 8 && ((waves == 4 && (lanes <= kSplitMaxLanes || (lanes > 49152 && lanes <= 65536))) ||
                                                          (waves == 6 && lanes <= 16384)));  // six waves, one workgroup per CU: dds.hip lockin_waves_for
    if (layout == IDSP_LANE_MAJOR) {
        // input by DMA (whole 128-byte lines, each requested once) for the 4-wave I/Q and norm_sqr forms: 0.48 -> 0.43-0.44 ms
        // at 32768 lanes x 4096 frames, 0.94 -> 0.86 at 65536; its 32 KiB ring halves the workgroups a CU can hold, which
        // costs the 6-wave form and the arg read-out more than the input gains (tools/exp_lockin_lm.py)
        if (!no_dma && waves == 4 && MODE != MODE_ARG)
            return launch_lockin_waves_in<MODE, Bank, IN_LM_DMA, kLwBLm>(p, st, x, y, lanes, frames, waves, s, lo, pitch);
        return launch_lockin_waves_in<MODE, Bank, IN_LM_REG, kLwBLm>(p, st, x, y, lanes, frames, waves, s, lo, pitch);
    }
    if (!no_dma && lanes % kWave == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0) {
        if (b16) return launch_lockin_waves_in<MODE, Bank, IN_FM_DMA, 16>(p, st, x, y, lanes, frames, waves, s, lo, pitch);
        return launch_lockin_waves_in<MODE, Bank, IN_FM_DMA, kLwB>(p, st, x, y, lanes, frames, waves, s, lo, pitch);
    }
    return launch_lockin_waves_in<MODE, Bank, IN_FM_REG, kLwB>(p, st, x, y, lanes, frames, waves, s, lo, pitch);
}

template <int MODE, int N, int K>
int launch_lockin_waves_nk(const LpParams &p, void *state, const int32_t *x, void *yv, size_t lanes, size_t frames, int layout,
                           int waves, hipStream_t s, size_t pitch)
{
    using Out = typename LwOut<MODE>::type;
    if constexpr (K == 2) {
        if (const int groups = lockin_stage_groups(x, lanes, frames, layout, K, MODE == MODE_ARG))
            return launch_lockin_stages<MODE, N>(p, state, x, yv, lanes, frames, groups, s);
    }
    return launch_lockin_waves_bank<MODE, LpBank<N, K>>(p, static_cast<uint32_t *>(state), x, static_cast<Out *>(yv), lanes, frames, layout, waves, s, nullptr, pitch);
}

template <int MODE>
int launch_lockin_waves(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, void *y, size_t lanes, size_t frames, int layout,
                        int waves, hipStream_t s, size_t pitch = 0)
{
    const LpParams p = lp_params(cfg);
#define IDSP_CASE(N, K) \
    if (cfg->order == N && cfg->cascade == K) return launch_lockin_waves_nk<MODE, N, K>(p, state, x, y, lanes, frames, layout, waves, s, pitch)
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
