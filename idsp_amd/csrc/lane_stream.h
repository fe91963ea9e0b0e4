// lane_stream.h — the two streaming kernels every per-lane recurrence of the
// hot path runs through (biquad family, lowpass, DDS, lock-in).
//
// Replaces "L2 block loop x L3 process body x N lanes" of the reference
// (dsp-process/src/process.rs:122-127,137-141 driven by Lanes,
// dsp-process/src/compose.rs:468-494) by ONE launch:
//   * one lane per thread; the serially dependent recurrence state lives in
//     VGPRs for the whole call and is read/written once (word-plane-major
//     state, include/idsp_hip.h);
//   * the shared configuration arrives as a kernarg POD, i.e. in SGPRs —
//     uniform across the wave, no LDS or vector loads on the hot loop;
//   * FRAME_MAJOR: a wave reads one 256-byte row segment per frame through a
//     rotating register window that requests frame f+U before frame f is
//     consumed, so a wave keeps U row segments in flight at all times (HBM
//     latency >> per-sample work, and at 64k lanes there is exactly one wave
//     per SIMD — no other wave hides it);
//   * LANE_MAJOR: adjacent lanes are `frames` elements apart, so a wave moves
//     64-lane x TS-sample tiles through a padded LDS tile: row-contiguous
//     (coalesced) global accesses on one side, conflict-free column walks by
//     the owning thread on the other.
//
// A processor P provides:
//   using In / Out;            element types (Out may be a 2-word struct)
//   static constexpr bool HAS_IN;
//   struct Params;             POD, passed by value as kernarg
//   load(prm, st, lanes, lane) / store(...)   state <-> registers
//   Out step(prm, In x)        one sample of the reference `process()`
//   static constexpr int LDS_WORDS;   read-only table words (0 = none), filled
//   by static fill_shared(uint32_t*) with the whole block and handed to the
//   processor through set_shared(const uint32_t*).
//   static constexpr int BATCH; Pre pre(prm); Out step(prm, x, pre)   (optional,
//   BATCH > 1): the part of a sample's work that does not depend on the
//   recurrence state (the DDS phase -> cossin chain) is evaluated for BATCH
//   consecutive frames back to back, which gives the scheduler independent
//   dependency chains to interleave — at one wave per SIMD the per-frame
//   dependent-chain latency, not throughput, is what bounds these kernels.
//   static constexpr int IN_DIV;      FRAME_MAJOR only: IN_DIV adjacent threads
//   ("virtual lanes") share one input lane — used to split the I and Q arms of
//   the lock-in / DDS over two threads when there are too few lanes to give
//   every SIMD a wave; `lanes` then counts virtual lanes.
#pragma once

#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <typeinfo>

#include "common.h"
#include "dispatch_thresholds.h"
#include "lds_dma.h"

namespace idsp {

template <class P, class = void>
struct BatchOf {
    static constexpr int value = 1;
};
template <class P>
struct BatchOf<P, std::void_t<decltype(P::BATCH)>> {
    static constexpr int value = P::BATCH;
};

// Processors with a large per-sample body (many uniform branches) cap the register-window depth so that the
// unrolled loop stays inside the instruction cache: `static constexpr int MAX_U`.
template <class P, class = void>
struct MaxU {
    static constexpr int value = 24;
};
template <class P>
struct MaxU<P, std::void_t<decltype(P::MAX_U)>> {
    static constexpr int value = P::MAX_U;
};

// BATCH pre-stage values: processors may provide pre_batch(prm, Pre (&)[BATCH]) (e.g. to share
// the work between the IN_DIV threads of a lane); the default evaluates pre() BATCH times.
template <class P, class = void>
struct HasPreBatch : std::false_type {};
template <class P>
struct HasPreBatch<P, std::void_t<decltype(&P::pre_batch)>> : std::true_type {};

// Processors with a tile form — P::tile<SB, R>(prm, P *lanes, x, y): SB independent lanes x R consecutive samples in one call, the
// state-independent part of all samples issued ahead of the per-sample chains (biquad_sections.h, Df1I32::tile) — declare HAS_TILE.
template <class P, class = void>
struct HasTileOf : std::false_type {};
template <class P>
struct HasTileOf<P, std::void_t<decltype(P::HAS_TILE)>> : std::integral_constant<bool, P::HAS_TILE> {};

template <class P, int B>
__device__ __forceinline__ void pre_all(P &p, const typename P::Params &prm, typename P::Pre (&pre)[B])
{
    if constexpr (HasPreBatch<P>::value) {
        p.pre_batch(prm, pre);
    } else {
#pragma unroll
        for (int b = 0; b < B; b++) pre[b] = p.pre(prm);
    }
}

// one sample through processors with or without a state-independent pre-stage
template <class P>
__device__ __forceinline__ typename P::Out step1(P &p, const typename P::Params &prm, typename P::In v)
{
    if constexpr (BatchOf<P>::value > 1)
        return p.step(prm, v, p.pre(prm));
    else
        return p.step(prm, v);
}

// (a function of its own: inside the kernels' generic lambdas a discarded `if constexpr` branch is still checked against P)
template <class P, int SB, int R>
__device__ __forceinline__ void tile_of(const typename P::Params &prm, P *lanes, const typename P::In (&x)[SB * R], typename P::Out (&y)[SB * R])
{
    if constexpr (HasTileOf<P>::value) {
        P::template tile<SB, R>(prm, lanes, x, y);
    } else {
#pragma unroll
        for (int g = 0; g < SB * R; g++) y[g] = step1(lanes[g % SB], prm, x[g]);
    }
}

// words <-> sample helpers (1- or 2-word element types)
template <class T>
__device__ __forceinline__ T words_to(const uint32_t *w)
{
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, w[0]);
    } else {
        const uint64_t u = uint64_t(w[0]) | (uint64_t(w[1]) << 32);
        return __builtin_bit_cast(T, u);
    }
}
template <class T>
__device__ __forceinline__ void to_words(const T &v, uint32_t *w)
{
    if constexpr (sizeof(T) == 4) {
        w[0] = __builtin_bit_cast(uint32_t, v);
    } else {
        const uint64_t u = __builtin_bit_cast(uint64_t, v);
        w[0] = uint32_t(u), w[1] = uint32_t(u >> 32);
    }
}

// streamed-once data: nontemporal loads/stores (struct outputs go out as a 2-word vector).
// NT = false for arithmetic-bound processors, where the longer nontemporal load latency
// showed up as a slowdown (8-section cascade) instead of a bandwidth gain.
// (Both name the GLOBAL address space: a pointer rebuilt from a wave-uniform integer base is a generic one otherwise and the access a
// `flat_*` instruction, which counts on lgkmcnt beside vmcnt — tools/check_flat.py keeps the library free of them.)
// a 4-, 8- or 16-byte object as a vector of words (class types have no copy operations in a named address space)
template <int BYTES>
struct WordsOf;
template <>
struct WordsOf<4> {
    using type = uint32_t;
};
template <>
struct WordsOf<8> {
    typedef uint32_t type __attribute__((ext_vector_type(2)));
};
template <>
struct WordsOf<16> {
    typedef uint32_t type __attribute__((ext_vector_type(4)));
};
template <bool NT, class T>
__device__ __forceinline__ T nt_load(const T *p)
{
#ifdef IDSP_EXP_GENERIC_NT  // A/B: the generic-pointer form of rounds 1-5
    if constexpr (NT)
        return __builtin_nontemporal_load(p);
    else
        return *p;
#endif
    using W = typename WordsOf<int(sizeof(T))>::type;
    const auto *g = reinterpret_cast<const __attribute__((address_space(1))) W *>(reinterpret_cast<uintptr_t>(p));
    if constexpr (NT)
        return __builtin_bit_cast(T, __builtin_nontemporal_load(g));
    else
        return __builtin_bit_cast(T, *g);
}
template <bool NT, class T>
__device__ __forceinline__ void nt_store(T *p, const T &v)
{
#ifdef IDSP_EXP_GENERIC_NT
    if constexpr (!NT) {
        *p = v;
    } else if constexpr (sizeof(T) == 4) {
        __builtin_nontemporal_store(__builtin_bit_cast(uint32_t, v), reinterpret_cast<uint32_t *>(p));
    } else {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(__builtin_bit_cast(u32x2, v), reinterpret_cast<u32x2 *>(p));
    }
    return;
#endif
    using W = typename WordsOf<int(sizeof(T))>::type;
    auto *g = reinterpret_cast<__attribute__((address_space(1))) W *>(reinterpret_cast<uintptr_t>(p));
    if constexpr (NT)
        __builtin_nontemporal_store(__builtin_bit_cast(W, v), g);
    else
        *g = __builtin_bit_cast(W, v);
}

constexpr int kWave = 64;
constexpr int kFmBlock = 256;  // 4 waves: one per SIMD of a CU, 1 KiB row segment per block

// ---------------------------------------------------------------- FRAME_MAJOR
template <class P, int U>
__global__ __launch_bounds__(kFmBlock) void stream_frame_major(
    const typename P::Params prm, uint32_t *st, const typename P::In *x, typename P::Out *y,
    const size_t lanes, const size_t frames, const size_t xl, const size_t yl, const int xcd_contiguous, const size_t slanes_ = 0)
{
    using In = typename P::In;
    using Out = typename P::Out;
    // slanes: pitch of the state (and per-lane coefficient) planes in lanes; 0 = `lanes` (not a lane block of a larger call)
    const size_t slanes = slanes_ ? slanes_ : lanes;
    __shared__ uint32_t ptab[P::LDS_WORDS ? P::LDS_WORDS : 1];
    // launched with 256-thread workgroups when that still gives every CU one, else with single waves
    if constexpr (P::LDS_WORDS > 0) {
        P::fill_shared(ptab, threadIdx.x, int(blockDim.x));
        __syncthreads();
    }
    // xcd_contiguous (rows that do not start on 64-byte boundaries, see stream_frame_major_lds XCDC): workgroup blockIdx runs
    // on XCD blockIdx % 8; give XCD j the j-th contiguous eighth of the lane blocks so that the lines two neighbouring blocks
    // share are fetched and written through one L2
    size_t wg = blockIdx.x;
    if (xcd_contiguous) {
        const size_t q = gridDim.x / 8, r = gridDim.x % 8, j = blockIdx.x % 8;
        wg = j * q + (j < r ? j : r) + blockIdx.x / 8;
    }
    const size_t lane = wg * blockDim.x + threadIdx.x;
    if (lane >= lanes) return;

    P p;
    if constexpr (P::LDS_WORDS > 0) p.set_shared(ptab);
    p.load(prm, st, slanes, lane);

#ifdef IDSP_NO_NT
    constexpr bool kNT = false;
#else
    constexpr bool kNT = P::COST <= 220;
#endif
    // xl / yl: elements between consecutive frames of x / y (dense: lanes / IN_DIV and lanes — IN_DIV virtual
    // lanes share an input lane)
    const In *xp = x + lane / P::IN_DIV;
    Out *yp = y + lane;

    if constexpr (P::HAS_IN) {
        // Rotating register window: frame f + U is requested right before frame
        // f is consumed, so U row segments per wave are in flight at all
        // times (vmcnt is in-order: the wait for frame f tolerates the U-1
        // younger loads and the interleaved stores).  Reads always run ahead of
        // this thread's writes, so y == x is safe.
        In ring[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            if (size_t(u) < frames) ring[u] = nt_load<kNT>(xp + size_t(u) * xl);
        size_t f = 0;
        for (; f + 2 * U <= frames; f += U) {
            const In *xn = xp + (f + U) * xl;
            Out *yn = yp + f * yl;
            constexpr int B = BatchOf<P>::value;
            static_assert(U % B == 0, "window depth must be a multiple of the batch");
#pragma unroll
            for (int u0 = 0; u0 < U; u0 += B) {
                if constexpr (B > 1) {
                    typename P::Pre pre[B];
                    pre_all<P, B>(p, prm, pre);
#pragma unroll
                    for (int b = 0; b < B; b++) {
                        const int u = u0 + b;
                        const In v = ring[u];
                        ring[u] = nt_load<kNT>(xn + size_t(u) * xl);
                        nt_store<kNT>(yn + size_t(u) * yl, p.step(prm, v, pre[b]));
                    }
                } else {
                    const int u = u0;
                    const In v = ring[u];
                    ring[u] = nt_load<kNT>(xn + size_t(u) * xl);
                    nt_store<kNT>(yn + size_t(u) * yl, p.step(prm, v));
                }
            }
        }
        // drain: fewer than 2U frames left, predicates are wave-uniform
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t fr = f + u;
            if (fr < frames) {
                const In v = ring[u];
                if (fr + U < frames) ring[u] = nt_load<kNT>(xp + (fr + U) * xl);
                nt_store<kNT>(yp + fr * yl, step1(p, prm, v));
            }
        }
        f += U;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t fr = f + u;
            if (fr < frames) nt_store<kNT>(yp + fr * yl, step1(p, prm, ring[u]));
        }
    } else {
        constexpr int B = BatchOf<P>::value;
        size_t f = 0;
        if constexpr (B > 1) {
            for (; f + B <= frames; f += B) {
                typename P::Pre pre[B];
                pre_all<P, B>(p, prm, pre);
#pragma unroll
                for (int b = 0; b < B; b++) nt_store<kNT>(yp + size_t(b) * yl, p.step(prm, In{}, pre[b]));
                yp += size_t(B) * yl;
            }
        }
        for (; f < frames; f++) {
            nt_store<kNT>(yp, step1(p, prm, In{}));
            yp += yl;
        }
    }
    p.store(prm, st, slanes, lane);
}

// ------------------------------------------------- FRAME_MAJOR through LDS (DMA)
// At <= 2 waves per SIMD the register-window kernel above is limited by what a
// 4-byte-per-lane access stream can pull from HBM (~5.3-5.5 TB/s measured at 65536
// lanes).  This variant moves the same rows with 16-byte-per-lane accesses while
// keeping one lane per thread (all four SIMDs of a CU compute): a 256-thread
// block owns 256 lanes (1 KiB per frame row); its waves pull whole rows straight
// into an LDS ring with `global_load_lds_dwordx4` (no VGPR staging, NB tiles of T
// rows in flight), every thread then walks its own column with conflict-free
// ds_read_b32, and results leave through a double-buffered LDS tile as
// row-contiguous dwordx4 stores.  +10 % at C2 (6.07 vs 5.52 TB/s, tools/exp_lds.hip).
//
// The DMA loads are issued from inline asm (the compiler neither counts nor
// drains them), so their completion is tracked by hand: vmcnt is in-order, and
// in steady state exactly kYoung younger operations (this wave's later DMA rows
// and its dwordx4 stores) are allowed to remain outstanding when tile i is
// needed; start-up, drain and ragged tiles simply wait for everything.
constexpr int kLdsT = 8;    // frames per tile
// Ring depth (input tiles in flight, 8 KiB each at 256 lanes): 8 unless the processor names another (P::LDS_RING).  It sets
// how far the DMA reads run ahead of the stores.  Measured at C2 (65536 lanes, 256 KiB rows; tools/probe_xy_offset.py over
// output placements y - x = 1 GiB + delta and in place), plain i32 DF1: 8 tiles 0.348-0.412 ms depending on delta (in
// place 0.39), 9 tiles 0.347-0.393, 7 tiles 0.333-0.354 (in place 0.34), 6 tiles 0.343-0.363, 5 tiles 0.340-0.354: with 8
// (and 9) the read and the write stream sit 16 (18) MiB apart and collide on the HBM channels unless y happens to sit at
// a lucky offset from x.  The clamp / wide / multi-section processors are the other way round (8 tiles 0.35-0.40 ms by
// placement, 7 or fewer a flat 0.41-0.46 ms), so only DF1 i32, dither and f32 DF2T without clamp take 7.
constexpr int kLdsNB = 8;
// Largest non-persistent grid of the LDS-DMA kernel; launches with more workgroups walk their lane blocks persistently.
constexpr size_t kLdsGridCap = thr::kLdsGridCap;
// LDS-DMA path or register-window kernel (launch_stream): P::LDS_ELIGIBLE if the processor declares it, else COST <= 120.
template <class P, class = void>
struct LdsEligibleOf {
    static constexpr bool value = P::COST <= 120;
};
template <class P>
struct LdsEligibleOf<P, std::void_t<decltype(P::LDS_ELIGIBLE)>> {
    static constexpr bool value = P::LDS_ELIGIBLE;
};
// Largest lanes-per-thread factor the LDS-DMA kernel is instantiated with for a processor (P::LDS_LPT_MAX; 1 = only
// the one-lane-per-thread form).  Each step doubles the processor's state registers.
template <class P, class = void>
struct LdsLptMaxOf {
    static constexpr int value = 1;
};
template <class P>
struct LdsLptMaxOf<P, std::void_t<decltype(P::LDS_LPT_MAX)>> {
    static constexpr int value = P::LDS_LPT_MAX;
};
template <class P, class = void>
struct LdsRingOf {
    static constexpr int value = kLdsNB;
};
template <class P>
struct LdsRingOf<P, std::void_t<decltype(P::LDS_RING)>> {
    static constexpr int value = P::LDS_RING;
};
// Addressing form of the LDS-DMA kernel per processor (see the kernel): indexed unless the processor asks for the running
// form (P::LDS_RUN).  Ring depth and form per section type come from tools/tune_lds.hip (profiles/r02_tune_lds.jsonl).
template <class P, class = void>
struct LdsRunOf {
    static constexpr bool value = false;
};
template <class P>
struct LdsRunOf<P, std::void_t<decltype(P::LDS_RUN)>> {
    static constexpr bool value = P::LDS_RUN;
};

// tools/exp_lm_ablate.sh: stream_lane_major_staged and stream_frame_major_lds without their loads / stores (conditions never true at run time: timing only)
#ifdef IDSP_EXP_LM_NOLOAD
#define IDSP_EXP_LM_LD_ON (frames == 1)
#else
#define IDSP_EXP_LM_LD_ON true
#endif
#ifdef IDSP_EXP_LM_NOSTORE
#define IDSP_EXP_LM_ST_ON (frames == 1)
#else
#define IDSP_EXP_LM_ST_ON true
#endif
#ifndef IDSP_EXP_LDS_PLAIN_LD  // experiment: plain instead of nontemporal accesses in every instantiation of the LDS-DMA kernel
#define IDSP_EXP_LDS_PLAIN_LD 0
#endif
#ifndef IDSP_EXP_LDS_PLAIN_ST
#define IDSP_EXP_LDS_PLAIN_ST 0
#endif
#ifndef IDSP_XCDC_LOAD_NT
#define IDSP_XCDC_LOAD_NT 1
#endif
#ifndef IDSP_XCDC_STORE_NT
#define IDSP_XCDC_STORE_NT 1
#endif
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// LPT ("lanes per thread") > 1: a workgroup owns 256 * LPT adjacent lanes and thread t runs the LPT independent
// recurrences of lanes t, t + 256, ...  A tile is still TS one-KiB row segments (8 KiB; 16 KiB for LPT = 16) —
// frames [v R, v R + R) x all LPT sub-blocks, R = TS / LPT — so ring bytes, barriers per sample and the vmcnt
// bookkeeping are those of LPT = 1, and every workgroup sweeps whole (LPT KiB) row pieces in address order, so that a
// launch of L lanes needs only L / (256 LPT) workgroups.  MEASURED AND NOT USED BY DEFAULT (no processor declares
// LDS_LPT_MAX > 1): with the output right behind the input in one allocation the form runs 131072 lanes at 0.76-0.78
// of the HBM peak and 262144 at 0.72 (one lane per thread on a persistent grid: 0.70 / 0.67), but over nine placements
// of the output buffer it ranges 0.61-0.78 / 0.62-0.77 (mean 0.68 / 0.67) where the persistent one-lane form ranges
// 0.69-0.75 / 0.65-0.69 (mean 0.72 / 0.68) — profiles/r02_exp_c5_place.jsonl, tools/exp_c5_place.hip.  The template
// parameter stays for those tools and for IDSP_DIAG=1 IDSP_LDS_LPT experiments on processors that opt in.
// XCDC (round 3): rows that do not start on 64-byte boundaries — dense rows of 65000 or 65532 lanes, odd pitches.  The 1 KiB
// row segments of adjacent lane blocks then share the 64-byte pieces at their boundaries, and with consecutive blocks dealt
// round-robin to the 8 XCDs (each with its own L2) both halves of every shared piece are fetched — and partially written —
// through two different L2s: 0.58-0.64 of the HBM peak against 0.77 on aligned rows (tools/exp_fm_pitch.py: same lanes, same
// kernel, pitch 65536 + {4, 8, 16, 32} lanes -> 0.59 / 0.64 / 0.76 / 0.75).  XCDC gives every XCD a CONTIGUOUS eighth of the
// lane blocks, so that neighbours meet in one L2: 65000 dense lanes 0.62 -> 0.71, pitch 65540 0.59 -> 0.67, aligned rows
// 0.78 -> 0.77 (tools/exp_fm_misaligned.hip, profiles/r03_exp_fm_misaligned.jsonl) — used for misaligned rows only.
// Built, bit-exact and removed again: rotating the piece -> thread assignment per row so that every 16-lane quad of a
// `dwordx4` request starts on a 64-byte boundary (the LDS-DMA destination is fixed by the hardware lane, so the row sits
// rotated in LDS and the owning thread reads its column through the rotation) ran at 0.56 on EVERY pitch, aligned ones
// included: the straddling quads were not the cost.
// ILV (round 5, LPT > 1): the LPT sub-blocks of a workgroup are INTERLEAVED over the row — workgroup w of G owns the 256-lane
// blocks w, w + G, w + 2G, ... — instead of adjacent.  One round of G workgroups then requests every row as ONE dense piece in
// the very order the single-round C2 launch does (consecutive 1 KiB segments from consecutive workgroups, i.e. from the 8 XCDs
// in turn), whatever the lane count: the launch sweeps memory front to back instead of visiting a 256 KiB panel of every row
// per round (see launch_stream and profiles/NOTES.md, round 5: the strided panel walk is what holds C5 at 0.67-0.70; dense
// rows run at 0.78 at a 32 GiB footprint as well).
template <class P, int NB = LdsRingOf<P>::value, int LPT = 1, bool RUN = LdsRunOf<P>::value, bool XCDC = false, bool ILV = false>
__global__ __launch_bounds__(kFmBlock) void stream_frame_major_lds(
    const typename P::Params prm, uint32_t *st, const typename P::In *x, typename P::Out *y,
    const size_t lanes, const size_t frames, const size_t xl, const size_t yl, const size_t slanes)
{
    // slanes: pitch of the state (and per-lane coefficient) planes, in lanes — `lanes` unless this launch is a lane block of
    // a larger call (launch_stream splits a call into whole rounds here + a remainder on the staged kernel)
    using In = typename P::In;
    using Out = typename P::Out;
    static_assert(P::HAS_IN && P::IN_DIV == 1 && sizeof(In) == 4, "LDS path: one 4-byte input per lane and frame");
    static_assert(LPT >= 1 && (LPT & (LPT - 1)) == 0, "LPT is a power of two");
    constexpr int OW = sizeof(Out) / 4, B = BatchOf<P>::value;
    constexpr int TS = LPT > kLdsT ? LPT : kLdsT;  // 1 KiB row segments per tile
    constexpr int R = TS / LPT;                    // frames per tile
    constexpr int RPW = TS / 4;                    // segments per wave and tile
    constexpr int kYoung = RPW * OW + (NB - 1) * (RPW + RPW * OW);
    static_assert(kYoung <= 63 && TS % 4 == 0 && (LPT > 1 ? B == 1 : R % B == 0), "vmcnt range / tile shape");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *tin = smem;                               // [NB][TS][256]
    uint32_t *tout = smem + NB * TS * kFmBlock;         // [2][TS][256 * OW]
    uint32_t *ptab = tout + 2 * TS * kFmBlock * OW;     // [P::LDS_WORDS]
    const int tid = threadIdx.x, lid = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)smem;

    P p[LPT];
    if constexpr (P::LDS_WORDS > 0) {
        P::fill_shared(ptab, tid, kFmBlock);  // published by the first tile barrier
#pragma unroll
        for (int s = 0; s < LPT; s++) p[s].set_shared(ptab);
    }
#ifdef IDSP_EXP_LDS_SKEW  // experiment (tools/exp_c5_skew.sh): start-up stagger of persistent launches, ((b >> SHIFT) % MOD) x SKEW ticks of 10 ns
    if ((lanes + size_t(kFmBlock) * LPT - 1) / (size_t(kFmBlock) * LPT) > gridDim.x) {
        const long long d_ = (long long)(IDSP_EXP_LDS_SKEW) * ((blockIdx.x >> IDSP_EXP_LDS_SKEW_SHIFT) % IDSP_EXP_LDS_SKEW_MOD), t0_ = wall_clock64();
        for (long long spins = d_ / 8 + 16; d_ && spins > 0 && wall_clock64() - t0_ < d_; spins--) __builtin_amdgcn_s_sleep(8);
    }
#endif
    // Persistent over lane blocks: workgroup w walks the (256 LPT)-lane blocks w, w + grid, w + 2 grid, ... one after
    // the other (state load, the whole frame walk, state store per block).
    constexpr size_t kBlockLanes = size_t(kFmBlock) * LPT;
    // LPT == 1 with one-word outputs also takes a ragged last block (lanes % 256 != 0, lanes % 4 == 0: whole 16-byte pieces): a thread whose
    // piece of the 1 KiB row segment lies beyond the last lane acts as a CLONE of an in-range thread — it requests
    // that thread's piece again and stores that thread's results to that thread's address (same bytes, twice) — so
    // the steady-state loop carries no predicate and every wave still issues exactly RPW loads and RPW * OW stores
    // per tile (the vmcnt bookkeeping below).  Its own column computes on whatever the clone's piece holds and is
    // dropped; only the state store is predicated.
    const size_t nblocks = (lanes + kBlockLanes - 1) / kBlockLanes;  // LPT > 1: lanes % (256 LPT) == 0 (launcher)
    // (a workgroup that owns ADJACENT blocks instead — blocks [w rounds, (w + 1) rounds) — was measured slower: 0.58 vs
    // 0.68 of peak at 2^20 lanes, profiles/r02_exp_c5_matrix6.jsonl: the panel order keeps the concurrently active
    // columns in one contiguous 256 KiB piece of every row)
    // Order in which a persistent workgroup visits its column panels.  All workgroups advance through the frames in
    // lockstep, so with the plain order (0) the whole launch touches the same few rows of ONE 256 KiB panel at any instant.
    // Order 3 — the upper half of the grid starts half way through its panels — keeps two panels active: over nine
    // placements of the output buffer 0.71-0.77 against 0.68-0.72 at 131072 lanes, 0.67-0.71 against 0.65-0.67 at 262144,
    // 0.61-0.72 against 0.64-0.70 at 2^20 (profiles/r03_exp_c5_order.jsonl; tools/exp_c5_place.hip builds every order).
    // Single-round launches (C2) are the same walk in every order.
#ifndef IDSP_LDS_ORDER
#define IDSP_LDS_ORDER 3
#endif
#if IDSP_LDS_ORDER == 0
    for (size_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
#else
    // tools/exp_c5_place.hip: other orders in which a persistent workgroup visits its column panels (all workgroups advance
    // through the frames in lockstep, so at any instant the launch touches the same few rows of ONE panel; these spread it
    // over several).  1: rotated by XCD (blockIdx % 8); 2: rotated by groups of 32 adjacent workgroups; 3: the upper half of
    // the grid starts half way; 4: odd workgroups walk the panels backwards.
    const size_t rounds_ = (nblocks + gridDim.x - 1) / gridDim.x;
    for (size_t k_ = 0; k_ < rounds_; k_++) {
    size_t rr_ = k_;
    if (IDSP_LDS_ORDER == 1) rr_ = (k_ + (blockIdx.x & 7) * ((rounds_ + 7) / 8)) % rounds_;
    if (IDSP_LDS_ORDER == 2) rr_ = (k_ + (blockIdx.x >> 5) * ((rounds_ + 7) / 8)) % rounds_;
    if (IDSP_LDS_ORDER == 3) rr_ = (k_ + (blockIdx.x >= gridDim.x / 2 ? rounds_ / 2 : 0)) % rounds_;
    if (IDSP_LDS_ORDER == 4) rr_ = (blockIdx.x & 1) ? rounds_ - 1 - k_ : k_;
    // 5: as 3 with an ODD panel offset from 8 rounds up; 6: four quarter grids at offsets g (rounds / 4) + g; 7: eight groups at g (rounds / 8) + g
    if (IDSP_LDS_ORDER == 5) rr_ = (k_ + (blockIdx.x >= gridDim.x / 2 ? rounds_ / 2 + (rounds_ >= 8 ? 1 : 0) : 0)) % rounds_;
    if (IDSP_LDS_ORDER == 6) {
        const size_t g_ = size_t(blockIdx.x) * 4 / gridDim.x;
        rr_ = rounds_ >= 4 ? (k_ + g_ * (rounds_ / 4) + (rounds_ >= 8 ? g_ : 0)) % rounds_ : (k_ + (g_ >= 2 ? rounds_ / 2 : 0)) % rounds_;
    }
    if (IDSP_LDS_ORDER == 7) {
        const size_t g_ = size_t(blockIdx.x) * 8 / gridDim.x;
        rr_ = rounds_ >= 8 ? (k_ + g_ * (rounds_ / 8) + (rounds_ >= 16 ? g_ : 0)) % rounds_ : (k_ + (g_ >= 4 ? rounds_ / 2 : 0)) % rounds_;
    }
    // XCDC: workgroups are dealt to the XCDs round-robin (blockIdx % 8); give XCD j the j-th contiguous eighth of the blocks
    // (XCD j hosts the workgroups blockIdx = j, j + 8, ...: grid / 8 of them, one more on the first grid % 8 XCDs)
    const size_t q_ = gridDim.x / 8, r_ = gridDim.x % 8, j_ = blockIdx.x % 8;
    const size_t wg_ = XCDC ? j_ * q_ + (j_ < r_ ? j_ : r_) + blockIdx.x / 8 : blockIdx.x;
    const size_t blk = wg_ + rr_ * gridDim.x;
    if (blk >= nblocks) continue;
#endif
    static_assert(!ILV || (LPT > 1 && !XCDC), "ILV: interleaved sub-blocks of an LPT > 1 workgroup");
    const size_t lane0 = ILV ? blk * size_t(kFmBlock) : blk * kBlockLanes;
    const size_t sub = ILV ? nblocks * size_t(kFmBlock) : size_t(kFmBlock);  // lanes between the sub-blocks of this workgroup
    const size_t avail = lanes - lane0;  // lanes of this block that exist (>= 4)
    const bool ragged_block = LPT == 1 && OW == 1 && avail < size_t(kFmBlock);
    const int lid4 = ragged_block && size_t(lid * 4) >= avail ? int(size_t(lid * 4) % avail) : lid * 4;  // first lane of this thread's piece
    const bool lane_ok = !ragged_block || size_t(tid) < avail;
#pragma unroll
    for (int s = 0; s < LPT; s++) p[s].load(prm, st, slanes, lane_ok ? lane0 + size_t(s) * sub + tid : lanes - 1);
    // The state loads must have landed HERE, in a way the compiler's wait-count pass sees: it cannot see the DMA
    // requests of glds16(), and if it first needs a state register inside the steady-state loop it protects that use
    // with `s_waitcnt vmcnt(0)` on every iteration — which drains the whole prefetch ring each tile (0.52 instead of
    // 0.34 ms at C2 when a refactoring moved the first use).  vmcnt(0), expcnt / lgkmcnt untouched:
    __builtin_amdgcn_s_waitcnt(0x0F70);

    const size_t ntiles = (frames + R - 1) / R;
    const size_t nfull = frames / R;  // tiles [0, nfull) have all their rows
    // segment g of a tile = frame g / LPT of the tile, sub-block g % LPT; a ragged tile holds nseg(v) segments
    auto nseg = [&](size_t v) { return int((frames - v * R < size_t(R) ? frames - v * R : size_t(R)) * LPT); };
    // Two forms of the per-tile addressing, chosen per processor (P::LDS_RUN) by measurement — these kernels sit on a
    // timing edge where the scalar instructions between the barriers decide how the DMA requests and the stores
    // interleave (profiles/r02_exp_c2_clamp*.jsonl, tools/exp_c2_clamp.sh: C2 shape, four placements of y):
    //   RUN = false  ring slot = tile % NB, addresses from the tile index: plain DF1 / dither / f32 DF2T at ring 7
    //                run 0.796-0.805 of peak this way and 0.71-0.77 the other way;
    //   RUN = true   running ring slot, x / y pointers advanced per tile: the clamp forms (two more dependent
    //                operations per sample) run 0.77-0.78 at ring 5 this way in every placement, against 0.67-0.68
    //                (ring 5-7) or 0.67 / 0.75 by placement (ring 8) the other way.
    const size_t xstep = size_t(R) * xl, ystep = size_t(R) * yl * OW;
    const In *xq = x + lane0 + lid4;                                          // RUN: tile about to be requested
    uint32_t *yq = reinterpret_cast<uint32_t *>(y) + lane0 * OW + lid4;       // RUN: tile about to be stored
    int slot_run = 0;                                                         // RUN: i % NB
    auto issue = [&](size_t v, auto full) {
        constexpr bool FULL = decltype(full)::value;
        const int slot = RUN ? slot_run : int(v % NB), ns = FULL ? TS : nseg(v);
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int g = wave + 4 * j;
            if ((FULL || g < ns) && IDSP_EXP_LM_LD_ON) {
                const In *src = RUN ? xq + size_t(g / LPT) * xl + (g % LPT) * sub : x + (v * R + g / LPT) * xl + lane0 + (g % LPT) * sub + lid4;
                if constexpr ((XCDC && !IDSP_XCDC_LOAD_NT) || IDSP_EXP_LDS_PLAIN_LD)
                    glds16_plain(src, lds_base + uint32_t((slot * TS + g) * kFmBlock * 4));
                else
                    glds16(src, lds_base + uint32_t((slot * TS + g) * kFmBlock * 4));
            }
        }
        if constexpr (RUN) xq += xstep;
    };
    auto store = [&](size_t v, auto full) {
        constexpr bool FULL = decltype(full)::value;
        const uint32_t *o = tout + (v & 1) * TS * kFmBlock * OW;
        const int ns = FULL ? TS : nseg(v);
        uint32_t *yw = reinterpret_cast<uint32_t *>(y);
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int g = wave + 4 * j;
            if ((FULL || g < ns) && IDSP_EXP_LM_ST_ON) {
#pragma unroll
                for (int h = 0; h < OW; h++) {
                    const u32x4 v4 = *reinterpret_cast<const u32x4 *>(o + (g * OW + h) * kFmBlock + lid4);
                    uint32_t *dst = RUN ? yq + (size_t(g / LPT) * yl + (g % LPT) * sub) * OW + h * kFmBlock
                                        : yw + ((v * R + g / LPT) * yl + lane0 + (g % LPT) * sub) * OW + h * kFmBlock + lid4;
                    if constexpr ((XCDC && !IDSP_XCDC_STORE_NT) || IDSP_EXP_LDS_PLAIN_ST)
                        *reinterpret_cast<u32x4 *>(dst) = v4;
                    else
                        __builtin_nontemporal_store(v4, reinterpret_cast<u32x4 *>(dst));
                }
            }
        }
        if constexpr (RUN) {
            yq += ystep;
            slot_run = slot_run + 1 == NB ? 0 : slot_run + 1;  // the tile is done: its slot went to tile v + NB in issue()
        }
    };
    auto compute = [&](size_t v, auto full) {
        constexpr bool FULL = decltype(full)::value;
        const uint32_t *in = tin + (RUN ? slot_run : int(v % NB)) * TS * kFmBlock;
        uint32_t *o = tout + (v & 1) * TS * kFmBlock * OW;
        const int ns = FULL ? TS : nseg(v);
        if constexpr (B > 1) {  // LPT == 1: segments are consecutive frames of one lane
#pragma unroll
            for (int r0 = 0; r0 < TS; r0 += B) {
                if (!FULL && r0 >= ns) break;
                typename P::Pre pre[B];
#pragma unroll
                for (int b = 0; b < B; b++)
                    if (FULL || r0 + b < ns) pre[b] = p[0].pre(prm);
#pragma unroll
                for (int b = 0; b < B; b++) {
                    const int r = r0 + b;
                    if (FULL || r < ns) to_words<Out>(p[0].step(prm, __builtin_bit_cast(In, in[r * kFmBlock + tid]), pre[b]), o + (r * kFmBlock + tid) * OW);
                }
            }
        } else {
            static_for<TS>([&](auto gg) {
                constexpr int g = decltype(gg)::value;  // static: the LPT states stay in registers
                if (FULL || g < ns) to_words<Out>(p[g % LPT].step(prm, __builtin_bit_cast(In, in[g * kFmBlock + tid])), o + (g * kFmBlock + tid) * OW);
            });
        }
    };

    using Full = std::true_type;
    using Ragged = std::false_type;
    for (size_t t = 0; t < size_t(NB) && t < ntiles; t++) {
        slot_run = int(t);
        issue(t, Ragged{});
    }
    slot_run = 0;
    size_t i = 0;
    auto slow_iter = [&]() {  // start-up, drain and ragged tiles: wait for everything
        wait_vmcnt<0>();
        lds_barrier();
        compute(i, Ragged{});
        lds_barrier();
        if (i + NB < ntiles) issue(i + NB, Ragged{});
        store(i, Ragged{});
    };
    for (; i < ntiles && i < size_t(NB); i++) slow_iter();
    // steady state: all tiles involved are full, and every wave has issued exactly RPW loads
    // and RPW*OW stores per past tile, so kYoung younger operations may stay in flight
    for (; i + NB < nfull; i++) {
        wait_vmcnt<kYoung>();
        lds_barrier();  // all four waves' segments of tile i have landed
        compute(i, Full{});
        lds_barrier();  // out tile complete; ring slot i % NB is free again
        issue(i + NB, Full{});
        store(i, Full{});
    }
    for (; i < ntiles; i++) slow_iter();
#pragma unroll
    for (int s = 0; s < LPT; s++)
        if (lane_ok) p[s].store(prm, st, slanes, lane0 + size_t(s) * sub + tid);
    // No barrier needed here: the next block's first lds_barrier() (after each wave's lgkmcnt wait) orders this
    // block's last output-tile reads before the compute() that overwrites the tile, and its first DMA rows only
    // touch input slots whose last readers passed the barrier after the final compute().
    }
}

// ----------------------------------------------------------------- LANE_MAJOR
// One wave per workgroup; tiles are wave-private, so the syncs below only
// order this wave's own LDS traffic (lds_wave_sync: no vmcnt drain, the next
// tile's global loads stay in flight).
template <class P>
__global__ __launch_bounds__(kWave) void stream_lane_major(
    const typename P::Params prm, uint32_t *st, const typename P::In *x, typename P::Out *y,
    const size_t lanes, const size_t frames, const size_t xl, const size_t yl)
{
    // xl / yl: samples between the starts of consecutive lanes of x / y (dense: frames)
    using In = typename P::In;
    using Out = typename P::Out;
    // D = IN_DIV threads ("virtual lanes") share one lane: they read the same input sample and each
    // writes its own OW words of the lane's D*OW-word output sample; `lanes` counts virtual lanes.
    constexpr int D = P::IN_DIV;
    constexpr int IW = sizeof(In) / 4, OW = sizeof(Out) / 4, OWR = OW * D;  // words per input / output sample
    static_assert((IW == 1 || IW == 2) && (OWR == 1 || OWR == 2) && (D == 1 || D == 2), "samples are one or two 32-bit words");
    constexpr int MW = IW > OWR ? IW : OWR;
    constexpr int TS = kWave / MW;             // samples per tile and lane (widest row = 64 words)
    constexpr int RWI = TS * IW, RWO = TS * OWR;  // words per tile row
    constexpr int ROWS = kWave / D;            // lanes per wave
    constexpr int RPI = kWave / RWI;           // tile rows covered by one load instruction
    constexpr int NLD = ROWS / RPI;            // load instructions per tile
    constexpr bool kAlias = P::HAS_IN && RWI == RWO;

    __shared__ uint32_t tin[P::HAS_IN ? ROWS : 1][RWI + 1];
    __shared__ uint32_t tout_[kAlias ? 1 : ROWS][RWO + 1];
    uint32_t(*tout)[RWO + 1] = kAlias ? reinterpret_cast<uint32_t(*)[RWO + 1]>(tin) : tout_;

    const int lid = threadIdx.x;
    const size_t vlane = size_t(blockIdx.x) * kWave + lid;  // virtual lane of this thread
    const bool active = vlane < lanes;
    const size_t lane0 = size_t(blockIdx.x) * ROWS, rlanes = lanes / D;  // first lane of the wave, lane count
    const size_t nrows = rlanes - lane0 < size_t(ROWS) ? rlanes - lane0 : size_t(ROWS);
    const int trow = lid / D, tsub = lid % D;

    __shared__ uint32_t ptab[P::LDS_WORDS ? P::LDS_WORDS : 1];
    if constexpr (P::LDS_WORDS > 0) P::fill_shared(ptab, lid, kWave);  // published by the first tile sync

    P p;
    if constexpr (P::LDS_WORDS > 0) p.set_shared(ptab);
    if (active) p.load(prm, st, lanes, vlane);

    const uint32_t *xw = reinterpret_cast<const uint32_t *>(x);
    uint32_t *yw = reinterpret_cast<uint32_t *>(y);
    // word (row r, col c) handled by this thread in load instruction i: r = i*RPI + lrow, c = lcol
    const int lrow = lid / RWI, lcol = lid % RWI;

    uint32_t stage[NLD];
    auto fetch = [&](size_t t0) {
        if constexpr (P::HAS_IN) {
            const size_t nw = (frames - t0 < size_t(TS) ? frames - t0 : size_t(TS)) * IW;
#pragma unroll
            for (int i = 0; i < NLD; i++) {
                const size_t r = size_t(i) * RPI + lrow;
                stage[i] = (r < nrows && size_t(lcol) < nw) ? xw[((lane0 + r) * xl + t0) * IW + lcol] : 0u;
            }
        }
    };

    auto one = [&](size_t j) {
        In v{};
        if constexpr (P::HAS_IN) v = words_to<In>(&tin[trow][j * IW]);
        const Out o = step1(p, prm, v);
        to_words<Out>(o, &tout[trow][j * OWR + tsub * OW]);
    };

    fetch(0);
    for (size_t t0 = 0; t0 < frames; t0 += TS) {
        const size_t ncols = frames - t0 < size_t(TS) ? frames - t0 : size_t(TS);
        if constexpr (P::HAS_IN) {
#pragma unroll
            for (int i = 0; i < NLD; i++) tin[i * RPI + lrow][lcol] = stage[i];
        }
        lds_wave_sync();
        if (t0 + TS < frames) fetch(t0 + TS);  // next tile in flight during the arithmetic

        if (active) {
            if (ncols == size_t(TS)) {
                constexpr int B = BatchOf<P>::value;
                if constexpr (B > 1 && TS % B == 0) {
#pragma unroll
                    for (int j0 = 0; j0 < TS; j0 += B) {
                        typename P::Pre pre[B];
                        pre_all<P, B>(p, prm, pre);
#pragma unroll
                        for (int b = 0; b < B; b++) {
                            In v{};
                            if constexpr (P::HAS_IN) v = words_to<In>(&tin[trow][(j0 + b) * IW]);
                            to_words<Out>(p.step(prm, v, pre[b]), &tout[trow][(j0 + b) * OWR + tsub * OW]);
                        }
                    }
                } else if constexpr (MaxU<P>::value < 24) {
                    for (int j = 0; j < TS; j++) one(size_t(j));  // large body: keep the tile loop rolled
                } else {
#pragma unroll
                    for (int j = 0; j < TS; j++) one(size_t(j));
                }
            } else {
                for (size_t j = 0; j < ncols; j++) one(j);
            }
        }
        lds_wave_sync();
        // row-contiguous stores: one instruction = up to 64 consecutive words of one lane
        const size_t nw = ncols * OWR;
        for (size_t r = 0; r < nrows; r++) {
            if (size_t(lid) < nw) yw[((lane0 + r) * yl + t0) * OWR + lid] = tout[r][lid];
        }
        lds_wave_sync();
    }
    if (active) p.store(prm, st, lanes, vlane);
}

// ------------------------------------------- LANE_MAJOR, 16-byte pieces (staged)
// The kernel above moves a lane's row as 4-byte pieces (one 256-byte run of ONE lane per instruction): 64 loads, 64
// stores and 256 LDS accesses per 64 x 64 tile, all on the issue path of a single wave, and it asks HBM for 256-byte
// runs per lane.  This form moves 512-byte runs per lane with 16 bytes per thread in both directions:
//   * a tile is 64 lanes x kLmRun bytes (TF = 128 / W frames of W words, PCS = 32 pieces per lane); load instruction j
//     of a tile fetches the runs of the G = 2 lanes j and j + 32 — 32 threads per run — into VGPRs, and the whole tile
//     is handed to one 32 KiB LDS slot with linear ds_write_b128 (thread t's piece of instruction j at 1024 j + 16 t, so
//     lane l sits in slot row (l % PCS) G + l / PCS).  WHICH piece of the run a thread fetches is free: thread t takes
//     piece (t % PCS) ^ (j % 16), so the row of lane l holds piece k at 16 (k ^ (l % 16)) and the owning thread's
//     ds_read_b128 column walk is bank-conflict free without padding;
//   * thread l reads its pieces, runs the steps in registers, writes the results over the same pieces; the wave then
//     re-reads the slot linearly and stores whole runs.  The NEXT tile's loads are issued right after the hand-over,
//     so they are in flight during the arithmetic and the stores (the compiler counts them: plain loads).
//     Per 64 x 128 samples: 32 loads + 128 LDS + 32 stores instead of 128 + 512 + 128.
// What the run length is worth at 65536 lanes x 4096 frames (tools/tune_lm.hip, profiles/r02_tune_lm.jsonl; same
// arithmetic, i32 DF1): 128-byte runs 0.54 of the HBM peak whatever the prefetch depth, 256-byte 0.56-0.58, 512-byte
// 0.65-0.70 (this kernel; the 4-byte tile kernel 0.60-0.64), 1 KiB runs no better at the two waves per CU their LDS
// allows.  An LDS-DMA twin (global_load_lds_dwordx4 rings of 1-4 slots) ran the same rates per run length: the bound
// is the access pattern (65536 concurrent streams of short runs 16 KiB apart), not latency or issue.
// One wave per workgroup, no barriers.  Needs 16-byte aligned rows (base and pitch).  A last partial tile moves its
// whole 16-byte pieces the same way; the final frames % (4 / W) samples of every lane go sample by sample.  x == y is
// safe (a tile is stored after it has been read; the rows of one wave are touched by no other).
constexpr int kLmRun = 512;  // bytes per lane and tile on the wider of the two sides
// Processors the staged kernel takes: one lane per thread (IN_DIV == 1), 4- or 8-byte samples, no input or an input of the
// output's size, and a pre-stage batch of 1 or 4 frames.  The kernel itself also handles samples of different sizes (a tile
// is then 128 / max(words) frames: 512-byte runs on the wider side, 256-byte runs on the narrower, separate LDS slots), but
// its 48 KiB of LDS leave a CU three waves and those processors (fm_disc, the lock-in with Complex output) are VALU-bound:
// fm_disc LaneMajor 1.89 ms staged against 1.46 ms on the tile kernel at 65536 lanes x 4096, so they stay there.
// A processor can opt out (P::LM_STAGED = false): with 128 staging registers next to a large state the kernel spills
// (tools/check_scratch.py is the build check).
template <class P, class = void>
struct LmStagedAllowed : std::true_type {};
template <class P>
struct LmStagedAllowed<P, std::void_t<decltype(P::LM_STAGED)>> : std::integral_constant<bool, P::LM_STAGED> {};
template <class P, class = void>
struct LmStagedOf {
    static constexpr bool value = false;
};
template <class P>
struct LmStagedOf<P, std::enable_if_t<P::IN_DIV == 1 && (!P::HAS_IN || sizeof(typename P::In) == sizeof(typename P::Out)) &&
                                      (sizeof(typename P::Out) == 4 || sizeof(typename P::Out) == 8) &&
                                      (BatchOf<P>::value == 1 || BatchOf<P>::value == 4) && LmStagedAllowed<P>::value>> {
    static constexpr bool value = true;
};
template <class P, class = void>
struct LmOneFormOf : std::false_type {};
template <class P>
struct LmOneFormOf<P, std::void_t<decltype(P::LM_ONE_FORM)>> : std::integral_constant<bool, P::LM_ONE_FORM> {};
// bytes of LDS the kernel needs for P (slots + the processor's table)
template <class P, int LW, int LB = kLmRun>
constexpr size_t lm_staged_lds_bytes()
{
    constexpr int IW = P::HAS_IN ? int(sizeof(typename P::In)) / 4 : 0, OW = int(sizeof(typename P::Out)) / 4;
    constexpr int S = IW > OW ? IW : OW, TF = LB / 4 / S;
    constexpr int IB = TF * IW * 4, OB = TF * OW * 4;
    return size_t(LW) * (IB == OB ? OB : IB + OB) + size_t(P::LDS_WORDS) * 4;
}

// geometry of one side of a tile of LW lanes: RB bytes per lane, moved as PCS 16-byte pieces; one instruction covers G
// lanes, a tile takes NI instructions; instruction j covers lanes j, j + NI, ...; lane l sits in slot row (l % NI) G + l / NI
// with piece k at 16 (k ^ (l % 16))
template <int RB, int LW>
struct LmSide {
    static constexpr int PCS = RB / 16, G = PCS ? kWave / PCS : 1, NI = LW / G;
    static_assert(PCS == 0 || PCS == 16 || PCS == 32 || PCS == 64, "256-, 512- or 1024-byte runs");
};

// LW = lanes per wave (64, 32 or 16).  With fewer than 64 the wave still moves whole runs with all its threads, but only
// the first LW threads own a lane: a launch of few lanes then spreads over LW / 64 times as many waves (SIMDs) — below
// 65536 lanes a 64-lane wave per SIMD leaves most of the chip without a wave, and the per-lane recurrence is a serial
// chain that one wave cannot speed up.
template <class P, int LW = kWave, int LB = kLmRun>
__global__ __launch_bounds__(kWave) void stream_lane_major_staged(
    const typename P::Params prm, uint32_t *st, const typename P::In *x, typename P::Out *y,
    const size_t lanes, const size_t frames, const size_t xl, const size_t yl, const unsigned skew = 0, const unsigned skew_shift = 8)
{
    // Start-up skew (round 5 experiment; IDSP_DIAG=1 IDSP_LM_SKEW=ticks of 10 ns, IDSP_LM_SKEW_SHIFT): every wave of a launch issues its tile's 32
    // loads in one burst and its 32 stores in another, and all waves do so at the same time — a CU's memory pipe then serves a load burst
    // and a store burst one after the other (round 4: the kernel runs as the SUM of its directions).  Workgroups with an odd
    // (blockIdx >> shift) start `skew` ticks late, so that half the waves of a CU store while the other half load.
    // (round 6: `skew_shift` bits 8.. = number of phases, 0 = two: workgroup b of the first 1024 waits ((b >> shift) % phases) x skew ticks)
    const unsigned skew_ph = skew && blockIdx.x < 1024u ? (blockIdx.x >> (skew_shift & 31u)) % ((skew_shift >> 8) ? (skew_shift >> 8) : 2u) : 0u;
    if (skew_ph) {
        const long long d = (long long)(skew) * skew_ph, t0 = wall_clock64();
        for (long long spins = d / 8 + 16; spins > 0 && wall_clock64() - t0 < d; spins--) __builtin_amdgcn_s_sleep(8);
    }
    using In = typename P::In;
    using Out = typename P::Out;
    static_assert(LmStagedOf<P>::value, "one lane per thread, 4- or 8-byte samples, pre-stage batch 1 or 4");
    constexpr bool HAS_IN = P::HAS_IN;
    constexpr int IW = HAS_IN ? int(sizeof(In)) / 4 : 0, OW = int(sizeof(Out)) / 4;  // words per sample
    constexpr int S = IW > OW ? IW : OW;
    constexpr int TF = LB / 4 / S;                 // frames per tile (LB = bytes per lane and tile on the wider side)
    constexpr int IB = TF * IW * 4, OB = TF * OW * 4;  // bytes per lane and tile
    static_assert(LW == 64 || LW == 32 || LW == 16, "lanes per wave");
    using SI = LmSide<IB, LW>;
    using SO = LmSide<OB, LW>;
    constexpr int PI = SI::PCS, PO = SO::PCS;      // pieces per lane and tile
    constexpr int NII = HAS_IN ? SI::NI : 0, NIO = SO::NI;  // load / store instructions per tile
    constexpr bool kAlias = IB == OB;              // results overwrite the input pieces
    constexpr int NS = 16 / S;                     // samples per compute chunk (4 pieces of the wider side)
    constexpr int CI = NS * IW / 4, CO = NS * OW / 4;  // pieces per chunk
    constexpr int B = BatchOf<P>::value;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    char *const slot_in = reinterpret_cast<char *>(smem);
    char *const slot_out = slot_in + (kAlias ? 0 : LW * IB);
    uint32_t *ptab = reinterpret_cast<uint32_t *>(slot_out + LW * OB);  // [P::LDS_WORDS]
    const int lid = threadIdx.x;

    const size_t lane0 = size_t(blockIdx.x) * LW;
    const size_t nrows = lanes - lane0 < size_t(LW) ? lanes - lane0 : size_t(LW);
    const bool active = size_t(lid) < nrows;

    P p;
    if constexpr (P::LDS_WORDS > 0) {
        P::fill_shared(ptab, lid, kWave);
        lds_wave_sync();
        p.set_shared(ptab);
    }
    if (active) p.load(prm, st, lanes, lane0 + lid);

    // Addresses = wave-uniform 64-bit base (the wave's first lane, instruction j, tile: SGPRs) + a 32-bit per-thread
    // byte offset (the lane of the instruction's group and the piece): the global_load / store `saddr` form.  With one
    // 64-bit pointer per instruction in VGPRs the source and destination pointers sat next to the staged pieces and
    // the kernel spilled.  (The launcher checks that 64 rows span less than 4 GiB.)
    const size_t xrowb = xl * sizeof(In), yrowb = yl * sizeof(Out);  // bytes between lanes
    const char *const xbase = reinterpret_cast<const char *>(x) + lane0 * xrowb;
    char *const ybase = reinterpret_cast<char *>(y) + lane0 * yrowb;
    // mover role on either side: in instruction j, lane mq + j of the tile, piece mpc ^ ((mq + j) % 16) of its run
    const int mqi = PI ? (lid / (PI ? PI : 1)) * NII : 0, mpci = PI ? lid % (PI ? PI : 1) : 0;
    const int mqo = (lid / PO) * NIO, mpco = lid % PO;
    const uint32_t xoff = uint32_t(mqi) * uint32_t(xrowb), yoff = uint32_t(mqo) * uint32_t(yrowb);
    // owner role (threads < LW): slot row of this thread's lane, and the byte offset of its piece k = own ^ (16 k)
    const int ol = lid % LW;
    const uint32_t owni = NII ? uint32_t((ol % (NII ? NII : 1)) * SI::G + ol / (NII ? NII : 1)) * IB + uint32_t(ol & 15) * 16 : 0;
    const uint32_t owno = uint32_t((ol % NIO) * SO::G + ol / NIO) * OB + uint32_t(ol & 15) * 16;

    const size_t nfull = frames / TF;
    const int nquad = int((frames - nfull * TF) / 4);  // whole groups of 4 samples of the last, partial tile
    u32x4 stage[NII ? NII : 1];
    // WHOLE: all 64 lanes of the wave exist; FULL: the whole tile exists (else nq groups of 4 samples)
    auto fetch = [&](size_t v, auto whole, auto full, int nq) __attribute__((always_inline)) {
        if constexpr (HAS_IN) {
            const char *src = xbase + v * size_t(IB);
#pragma unroll
            for (int j = 0; j < NII; j++) {
                const int pc = mpci ^ (NII % 16 == 0 ? j & 15 : (mqi + j) & 15);  // mqi is a multiple of NII: a constant when 16 | NII
                if ((decltype(whole)::value || size_t(mqi + j) < nrows) && (decltype(full)::value || pc < nq * IW) && IDSP_EXP_LM_LD_ON)
                    stage[j] = global_ld<u32x4, true>(src + j * xrowb, xoff + uint32_t(pc * 16));
            }
        }
    };
    auto hand_over = [&]() __attribute__((always_inline)) {
        if constexpr (HAS_IN) {
#pragma unroll
            for (int j = 0; j < NII; j++) *reinterpret_cast<u32x4 *>(slot_in + j * 1024 + lid * 16) = stage[j];
        }
    };
    // four consecutive samples: IW input pieces in, OW output pieces out
    auto quad = [&](const u32x4 *in, u32x4 *out) __attribute__((always_inline)) {
        uint32_t wi[IW ? 4 * IW : 1], wo[4 * OW];
        if constexpr (HAS_IN) {
#pragma unroll
            for (int h = 0; h < IW; h++)
#pragma unroll
                for (int e = 0; e < 4; e++) wi[4 * h + e] = in[h][e];
        }
        auto sample = [&](int b) __attribute__((always_inline)) {
            In v{};
            if constexpr (HAS_IN) v = words_to<In>(wi + b * IW);
            return v;
        };
        if constexpr (B == 4) {
            typename P::Pre pre[4];
            pre_all<P, 4>(p, prm, pre);
#pragma unroll
            for (int b = 0; b < 4; b++) to_words<Out>(p.step(prm, sample(b), pre[b]), wo + b * OW);
        } else {
#pragma unroll
            for (int b = 0; b < 4; b++) to_words<Out>(p.step(prm, sample(b)), wo + b * OW);
        }
#pragma unroll
        for (int h = 0; h < OW; h++) out[h] = u32x4{wo[4 * h], wo[4 * h + 1], wo[4 * h + 2], wo[4 * h + 3]};
    };
    auto in_piece = [&](int k) __attribute__((always_inline)) { return *reinterpret_cast<const u32x4 *>(slot_in + (owni ^ uint32_t(k * 16))); };
    auto out_piece = [&](int k, const u32x4 &v) __attribute__((always_inline)) { *reinterpret_cast<u32x4 *>(slot_out + (owno ^ uint32_t(k * 16))) = v; };
    auto compute = [&](auto full, int nq) __attribute__((always_inline)) {
        if (!active) return;
        if constexpr (!decltype(full)::value || MaxU<P>::value < 24) {
            for (int q = 0; q < nq; q++) {  // partial tile, or a large body: keep the loop rolled
                u32x4 in[IW ? IW : 1], out[OW];
#pragma unroll
                for (int h = 0; h < IW; h++) in[h] = in_piece(q * IW + h);
                quad(in, out);
#pragma unroll
                for (int h = 0; h < OW; h++) out_piece(q * OW + h, out[h]);
            }
        } else {
            // chunks of NS samples (4 pieces of the wider side): the next chunk's LDS reads are issued before the
            // current chunk's arithmetic, and a scheduling fence per chunk keeps the compiler from hoisting all reads
            // of the tile above the first step (which, next to the staged pieces of the next tile, overflowed the
            // register file into scratch)
            constexpr int NCH = TF / NS;
            u32x4 cur[CI ? CI : 1], nxt[CI ? CI : 1];
#pragma unroll
            for (int c = 0; c < CI; c++) cur[c] = in_piece(c);
#pragma unroll
            for (int g = 0; g < NCH; g++) {
                if (g + 1 < NCH) {
#pragma unroll
                    for (int c = 0; c < CI; c++) nxt[c] = in_piece((g + 1) * CI + c);
                }
                u32x4 out[CO];
                if constexpr (HasTileOf<P>::value && IW == 1 && OW == 1 && B == 1) {
                    In xin[NS];
                    Out yo[NS];
#pragma unroll
                    for (int q = 0; q < NS; q++) xin[q] = __builtin_bit_cast(In, uint32_t(cur[q / 4][q % 4]));
                    tile_of<P, 1, NS>(prm, &p, xin, yo);
#pragma unroll
                    for (int q = 0; q < NS / 4; q++)
                        out[q] = u32x4{__builtin_bit_cast(uint32_t, yo[4 * q]), __builtin_bit_cast(uint32_t, yo[4 * q + 1]), __builtin_bit_cast(uint32_t, yo[4 * q + 2]),
                                       __builtin_bit_cast(uint32_t, yo[4 * q + 3])};
                } else {
#pragma unroll
                    for (int q = 0; q < NS / 4; q++) quad(cur + q * IW, out + q * OW);
                }
#pragma unroll
                for (int c = 0; c < CO; c++) out_piece(g * CO + c, out[c]);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int c = 0; c < CI; c++) cur[c] = nxt[c];
            }
        }
    };
    auto store = [&](size_t v, auto whole, auto full, int nq) __attribute__((always_inline)) {
        char *dst = ybase + v * size_t(OB);
#pragma unroll
        for (int j = 0; j < NIO; j++) {
            const int pc = mpco ^ (NIO % 16 == 0 ? j & 15 : (mqo + j) & 15);
            const u32x4 v4 = *reinterpret_cast<const u32x4 *>(slot_out + j * 1024 + lid * 16);
            if ((decltype(whole)::value || size_t(mqo + j) < nrows) && (decltype(full)::value || pc < nq * OW) && IDSP_EXP_LM_ST_ON)
                global_st<u32x4, true>(dst + j * yrowb, yoff + uint32_t(pc * 16), v4);
            if (j % 8 == 7) asm volatile("" ::: "memory");  // at most 8 pieces between LDS and the store (the staged tile holds PI)
        }
    };
    auto walk = [&](auto whole) __attribute__((always_inline)) {
        using Full = std::true_type;
        using Part = std::false_type;
        if (nfull > 0)
            fetch(0, whole, Full{}, TF / 4);
        else
            fetch(0, whole, Part{}, nquad);
        for (size_t i = 0; i < nfull; i++) {
            hand_over();
            lds_wave_sync();
            if (i + 1 < nfull)
                fetch(i + 1, whole, Full{}, TF / 4);
            else if (nquad > 0)
                fetch(i + 1, whole, Part{}, nquad);
            compute(Full{}, TF / 4);
            lds_wave_sync();
            store(i, whole, Full{}, TF / 4);
            lds_wave_sync();
        }
        if (nquad > 0) {
            hand_over();
            lds_wave_sync();
            compute(Part{}, nquad);
            lds_wave_sync();
            store(nfull, whole, Part{}, nquad);
        }
    };
    if (nrows == size_t(LW))
        walk(std::true_type{});
    else
        walk(std::false_type{});

    if (active) {  // the last frames % 4 samples of this lane's own row
        const In *xr = x + (lane0 + lid) * xl;
        Out *yr = y + (lane0 + lid) * yl;
        for (size_t f = nfull * TF + size_t(nquad) * 4; f < frames; f++) {
            In v{};
            if constexpr (HAS_IN) v = xr[f];
            yr[f] = step1(p, prm, v);
        }
        p.store(prm, st, lanes, lane0 + lid);
    }
}

// ------------------------------------------- FRAME_MAJOR, few lanes (staged)
// Below ~49152 lanes the FrameMajor kernels above stop being bandwidth-bound: a lane is a serial chain, the LDS-DMA
// kernel pays two workgroup barriers per 8 frames and the register-window kernel a global-load wait per frame, and both
// take the same ~0.33 ms for 4096 frames whatever the lane count (80 ns = 190 cycles per frame and wave), so the rate
// falls with the lanes: 0.61 of the HBM peak at 49152 lanes, 0.41 at 32768, 0.20 at 16384 — while the LaneMajor staged
// kernel, whose threads run 128 frames from LDS without a barrier, needs 36 ns per frame.  This kernel gives FrameMajor
// buffers the same structure: one wave per workgroup owns LW lanes (64 / 32 / 16, as in the LaneMajor kernel); a tile is
// 32 KiB = TF frames x LW lanes, fetched as 16-byte pieces (one instruction = 1 KiB = 1024 / (LW W 4) row pieces — the
// neighbouring waves read the neighbouring pieces of the same rows) into VGPRs one tile ahead, handed to LDS linearly
// (instruction j's 1 KiB at 1024 j: the slot is [frame][lane]), walked by the owning thread down its column (conflict
// free: consecutive lanes, consecutive banks; two frames per ds_read2st64), results written in place, the slot re-read
// linearly and stored as whole row pieces.  Rows must be 16-byte aligned (lanes and pitches multiples of 4 / W).
constexpr int kFmStagedTile = 32768;  // bytes per tile
template <class P, class = void>
struct FmStagedOf {
    static constexpr bool value = false;
};
template <class P>
struct FmStagedOf<P, std::enable_if_t<P::HAS_IN && P::IN_DIV == 1 && sizeof(typename P::In) == sizeof(typename P::Out) &&
                                      (sizeof(typename P::In) == 4 || sizeof(typename P::In) == 8) && BatchOf<P>::value == 1>> {
    static constexpr bool value = true;
};

template <class P, int LW>
__global__ __launch_bounds__(kWave) void stream_frame_major_staged(
    const typename P::Params prm, uint32_t *st, const typename P::In *x, typename P::Out *y,
    const size_t lanes, const size_t frames, const size_t xl, const size_t yl, const size_t slanes, const int xcd_contiguous = 0)
{
    using In = typename P::In;
    using Out = typename P::Out;
    static_assert(FmStagedOf<P>::value && (LW == 64 || LW == 32 || LW == 16), "one 4- or 8-byte input and output per lane and frame");
    constexpr int W = int(sizeof(In)) / 4;          // words per sample
    constexpr int RB = LW * W * 4;                  // bytes of a row piece (one frame of the wave's lanes)
    constexpr int TF = kFmStagedTile / RB;          // frames per tile
    constexpr int PPR = RB / 16;                    // 16-byte pieces per row piece
    constexpr int RPI = kWave / PPR;                // rows one instruction covers
    constexpr int NI = kFmStagedTile / 1024;        // instructions per tile
    constexpr int NS = 16;                          // frames per compute chunk
    static_assert(TF % NS == 0 && NI * RPI == TF, "tile shape");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    using Word = std::conditional_t<W == 1, uint32_t, uint64_t>;

    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    char *const slot = reinterpret_cast<char *>(smem);
    uint32_t *ptab = smem + kFmStagedTile / 4;  // [P::LDS_WORDS]
    const int lid = threadIdx.x;

    // xcd_contiguous (rows off the 64-byte grid): XCD j = blockIdx % 8 takes the j-th contiguous eighth of the lane blocks, so that the
    // 128-byte lines two neighbouring waves share are fetched and written through one L2 (see stream_frame_major_lds, XCDC)
    size_t wg = blockIdx.x;
    if (xcd_contiguous & 1) {
        const size_t q = gridDim.x / 8, r = gridDim.x % 8, j = blockIdx.x % 8;
        wg = j * q + (j < r ? j : r) + blockIdx.x / 8;
    }
    const size_t lane0 = wg * LW;
    const size_t nrows = lanes - lane0 < size_t(LW) ? lanes - lane0 : size_t(LW);  // lanes of this wave (a multiple of 4 / W)
    const bool active = size_t(lid) < nrows;

    P p;
    if constexpr (P::LDS_WORDS > 0) {
        P::fill_shared(ptab, lid, kWave);
        lds_wave_sync();
        p.set_shared(ptab);
    }
    if (active) p.load(prm, st, slanes, lane0 + lid);

    // mover role: in instruction j, row j RPI + mrow of the tile, piece mpc of the row piece; addresses = wave-uniform base
    // (SGPRs, see uniform_ptr) + 32-bit thread offset
    const size_t xrowb = xl * sizeof(In), yrowb = yl * sizeof(Out);  // bytes between frames
    const int mrow = lid / PPR, mpc = lid % PPR;
    const bool mine = size_t(mpc) * 16 < nrows * sizeof(In);  // this thread's piece lies inside the wave's lanes
    const char *const xbase = reinterpret_cast<const char *>(x) + lane0 * sizeof(In);
    char *const ybase = reinterpret_cast<char *>(y) + lane0 * sizeof(Out);
    const uint32_t xoff = uint32_t(mrow) * uint32_t(xrowb) + uint32_t(mpc) * 16, yoff = uint32_t(mrow) * uint32_t(yrowb) + uint32_t(mpc) * 16;

    const size_t nfull = frames / TF;
    const int ntail = int(frames - nfull * TF);  // frames of the last, partial tile
    u32x4 stage[NI];
    // FULL: all TF frames of the tile exist (else nf of them).  `mine` masks the pieces of missing lanes (last wave only; one
    // code path for whole and partial waves: the predicate is a loop-invariant exec mask, and the kernel compiles once)
    auto fetch_as = [&](size_t v, auto full, int nf, auto nt) __attribute__((always_inline)) {
        const char *src = xbase + v * TF * xrowb;
#pragma unroll
        for (int j = 0; j < NI; j++)
            if (mine && (decltype(full)::value || j * RPI + mrow < nf))
                stage[j] = global_ld<u32x4, decltype(nt)::value>(src + size_t(j * RPI) * xrowb, xoff);
    };
    // Rows off the 64-byte grid (xcd_contiguous & 2): the 128-byte lines at both ends of a wave's row piece are shared with the
    // neighbouring waves, and with nontemporal accesses each of the two fetches them from memory and writes its part back alone;
    // plain loads and stores let the second wave hit, and the two parts meet, in the L2 the XCD-contiguous order gives them in
    // common: 32769 lanes x 4096 frames 0.363 -> 0.268 ms, 32772 0.351 -> 0.278, 40001 0.489 -> 0.356 (round 4; plain stores
    // alone 0.341 / 0.312 / 0.418; aligned rows lose 4 % with plain accesses and keep the nontemporal ones).
    const bool plain = (xcd_contiguous & 2) != 0;
    auto fetch = [&](size_t v, auto full, int nf) __attribute__((always_inline)) {
        if (plain)
            fetch_as(v, full, nf, std::false_type{});
        else
            fetch_as(v, full, nf, std::true_type{});
    };
    auto hand_over = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NI; j++) *reinterpret_cast<u32x4 *>(slot + j * 1024 + lid * 16) = stage[j];
    };
    Word *const col = reinterpret_cast<Word *>(slot) + lid;  // this thread's column: frame f at col[f LW]
    auto one = [&](Word w) __attribute__((always_inline)) {
        uint32_t ww[W];
        ww[0] = uint32_t(w);
        if constexpr (W == 2) ww[1] = uint32_t(uint64_t(w) >> 32);
        to_words<Out>(p.step(prm, words_to<In>(ww)), ww);
        if constexpr (W == 2)
            return Word(uint64_t(ww[0]) | (uint64_t(ww[1]) << 32));
        else
            return Word(ww[0]);
    };
    auto compute = [&](auto full, int nf) __attribute__((always_inline)) {
        if (!active) return;
        if constexpr (!decltype(full)::value || MaxU<P>::value < 24) {
            for (int f = 0; f < nf; f++) col[f * LW] = one(col[f * LW]);  // partial tile, or a large body: keep the loop rolled
        } else {
            // chunks of NS frames: the next chunk's LDS reads are issued before the current chunk's arithmetic; the fence
            // keeps the compiler from hoisting all reads of the tile above the first step
            Word cur[NS], nxt[NS];
#pragma unroll
            for (int c = 0; c < NS; c++) cur[c] = col[c * LW];
#pragma unroll
            for (int g = 0; g < TF / NS; g++) {
                if (g + 1 < TF / NS) {
#pragma unroll
                    for (int c = 0; c < NS; c++) nxt[c] = col[((g + 1) * NS + c) * LW];
                }
                if constexpr (HasTileOf<P>::value && W == 1) {
                    In xin[NS];
                    Out yo[NS];
#pragma unroll
                    for (int c = 0; c < NS; c++) xin[c] = __builtin_bit_cast(In, cur[c]);
                    tile_of<P, 1, NS>(prm, &p, xin, yo);
#pragma unroll
                    for (int c = 0; c < NS; c++) col[(g * NS + c) * LW] = __builtin_bit_cast(Word, yo[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < NS; c++) col[(g * NS + c) * LW] = one(cur[c]);
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int c = 0; c < NS; c++) cur[c] = nxt[c];
            }
        }
    };
    auto store_as = [&](size_t v, auto full, int nf, auto nt) __attribute__((always_inline)) {
        char *dst = ybase + v * TF * yrowb;
#pragma unroll
        for (int j = 0; j < NI; j++) {
            const u32x4 v4 = *reinterpret_cast<const u32x4 *>(slot + j * 1024 + lid * 16);
            if (mine && (decltype(full)::value || j * RPI + mrow < nf)) {
                global_st<u32x4, decltype(nt)::value>(dst + size_t(j * RPI) * yrowb, yoff, v4);
            }
            if (j % 8 == 7) asm volatile("" ::: "memory");
        }
    };
    auto store = [&](size_t v, auto full, int nf) __attribute__((always_inline)) {
        if (plain)
            store_as(v, full, nf, std::false_type{});
        else
            store_as(v, full, nf, std::true_type{});
    };
    using Full = std::true_type;
    using Part = std::false_type;
    if (nfull > 0)
        fetch(0, Full{}, TF);
    else
        fetch(0, Part{}, ntail);
    for (size_t i = 0; i < nfull; i++) {
        hand_over();
        lds_wave_sync();
        if (i + 1 < nfull)
            fetch(i + 1, Full{}, TF);
        else if (ntail > 0)
            fetch(i + 1, Part{}, ntail);
        compute(Full{}, TF);
        lds_wave_sync();
        store(i, Full{}, TF);
        lds_wave_sync();
    }
    if (ntail > 0) {
        hand_over();
        lds_wave_sync();
        compute(Part{}, ntail);
        lds_wave_sync();
        store(nfull, Part{}, ntail);
    }
    if (active) p.store(prm, st, slanes, lane0 + lid);
}

// ------------------------------------------- FRAME_MAJOR, few lanes: one wave computes, one wave moves (round 6)
// Below ~24576 lanes a FrameMajor launch is a race against ONE wave's serial chain: 4096 frames x (39.5 cycles for the i32 DF1
// step at one wave per SIMD, tools/ubench_df1_step.hip) = 0.069 ms, where stream_frame_major_staged needs 0.139 ms at 16384 lanes
// and the f32 sections, whose chain is no longer, 0.157: the rest is its skeleton — the same in-order wave issues the tile's 32
// loads, hands them to LDS, walks its column, re-reads the tile and issues 32 stores — and a section that only copies takes 0.11 ms
// in it.  Here the two halves run side by side: a workgroup is TWO waves over LW = 32 lanes; wave 1 (the mover) brings tile k + 1 in
// by `global_load_lds_dwordx4` (8 row pieces of 128 bytes per request, straight into the tile: no staging registers) and takes
// tile k - 1 out (`ds_read_b128` + 16-byte stores) while wave 0 walks its column of tile k; two 16 KiB tiles (128 frames), ONE barrier
// per tile: at barrier k the mover has waited for tile k's requests and the compute wave has finished tile k - 1.  The compute wave's
// loop holds nothing but LDS reads (two frames per instruction), the processor's steps and LDS writes.
constexpr int kFmPairTile = 16384;  // bytes per tile: 128 frames x 32 lanes (32 KiB tiles: 16384 lanes 0.108 against 0.102 ms, and only two workgroups per CU)
template <class P, int TB = kFmPairTile>
__global__ __launch_bounds__(2 * kWave) void stream_frame_major_pair(
    const typename P::Params prm, uint32_t *st, const typename P::In *x, typename P::Out *y,
    const size_t lanes, const size_t frames, const size_t xl, const size_t yl, const size_t slanes)
{
    using In = typename P::In;
    using Out = typename P::Out;
    static_assert(FmStagedOf<P>::value && sizeof(In) == 4 && sizeof(Out) == 4, "one 4-byte input and output per lane and frame");
    constexpr int LW = 32;
    constexpr int RB = LW * 4;                      // bytes of a row piece (one frame of the workgroup's lanes)
    constexpr int TF = TB / RB;                     // frames per tile
    constexpr int PPR = RB / 16;                    // 16-byte pieces per row piece
    constexpr int RPI = kWave / PPR;                // rows one instruction covers
    constexpr int NI = TB / 1024;                   // instructions per tile
    constexpr int NS = 16;                          // frames per compute chunk
    static_assert(TF % NS == 0 && NI * RPI == TF, "tile shape");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    char *const slots = reinterpret_cast<char *>(smem);       // two tiles
    uint32_t *ptab = smem + 2 * TB / 4;                       // [P::LDS_WORDS]
    const int lid = threadIdx.x % kWave;
    const bool mover = __builtin_amdgcn_readfirstlane(int(threadIdx.x / kWave)) != 0;
    const size_t lane0 = size_t(blockIdx.x) * LW;
    const size_t nrows = lanes - lane0 < size_t(LW) ? lanes - lane0 : size_t(LW);  // lanes of this workgroup (a multiple of 4)
    const size_t ntiles = (frames + TF - 1) / TF;
    const int ntail = int(frames - (ntiles - 1) * TF);  // frames of the last tile (1 .. TF)

    if (mover) {
        // in instruction j: row j RPI + mrow of the tile, piece mpc of the row piece
        const size_t xrowb = xl * sizeof(In), yrowb = yl * sizeof(Out);
        const int mrow = lid / PPR, mpc = lid % PPR;
        const bool mine = size_t(mpc) * 16 < nrows * sizeof(In);  // this thread's piece lies inside the workgroup's lanes
        const char *const xbase = reinterpret_cast<const char *>(x) + lane0 * sizeof(In);
        char *const ybase = reinterpret_cast<char *>(y) + lane0 * sizeof(Out);
        const uint32_t xoff = uint32_t(mrow) * uint32_t(xrowb) + uint32_t(mpc) * 16, yoff = uint32_t(mrow) * uint32_t(yrowb) + uint32_t(mpc) * 16;
        const uint32_t slots_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)slots;
        auto load = [&](size_t k) __attribute__((always_inline)) {
            const char *src = xbase + k * TF * xrowb;
            const int nf = k + 1 == ntiles ? ntail : TF;
            const uint32_t dst = slots_lds + uint32_t(k & 1) * TB;
#pragma unroll
            for (int j = 0; j < NI; j++)
                if (mine && j * RPI + mrow < nf) glds16_s(uniform_ptr(src + size_t(j * RPI) * xrowb), xoff, dst + j * 1024);
        };
        auto store = [&](size_t k) __attribute__((always_inline)) {
            char *dst = ybase + k * TF * yrowb;
            const int nf = k + 1 == ntiles ? ntail : TF;
            const char *slot = slots + (k & 1) * TB;
#pragma unroll
            for (int j = 0; j < NI; j++) {
                const u32x4 v4 = *reinterpret_cast<const u32x4 *>(slot + j * 1024 + lid * 16);
                if (mine && j * RPI + mrow < nf) global_st<u32x4, true>(dst + size_t(j * RPI) * yrowb, yoff, v4);
                if (j % 8 == 7) asm volatile("" ::: "memory");
            }
        };
        load(0);
        for (size_t k = 0; k < ntiles; k++) {
            wait_vmcnt<0>();  // tile k has landed (and the stores of tile k - 2 are done)
            lds_barrier();    // barrier k: the compute wave has finished tile k - 1
            if (k >= 1) {
                store(k - 1);
                lds_wave_sync();  // its LDS reads are done: the tile may be overwritten
            }
            if (k + 1 < ntiles) load(k + 1);
        }
        lds_barrier();  // barrier ntiles: the last tile is computed
        store(ntiles - 1);
        return;
    }

    // ---- compute wave
    const bool active = size_t(lid) < nrows;
    P p;
    if constexpr (P::LDS_WORDS > 0) {
        P::fill_shared(ptab, lid, kWave);
        lds_wave_sync();
        p.set_shared(ptab);
    }
    if (active) p.load(prm, st, slanes, lane0 + lid);
    auto one = [&](uint32_t w) __attribute__((always_inline)) {
        uint32_t ww[1] = {w};
        to_words<Out>(p.step(prm, words_to<In>(ww)), ww);
        return ww[0];
    };
    for (size_t k = 0; k < ntiles; k++) {
        lds_barrier();  // barrier k: tile k has landed
        uint32_t *const col = reinterpret_cast<uint32_t *>(slots + (k & 1) * TB) + lid;  // frame f at col[f LW]
        if (!active) continue;
        if (k + 1 == ntiles && ntail != TF) {
            for (int f = 0; f < ntail; f++) col[f * LW] = one(col[f * LW]);
        } else if constexpr (MaxU<P>::value < 24) {
            for (int f = 0; f < TF; f++) col[f * LW] = one(col[f * LW]);  // a large body: keep the loop rolled
        } else {
            // chunks of NS frames: the next chunk's LDS reads are issued before the current chunk's arithmetic
            uint32_t cur[NS], nxt[NS];
#pragma unroll
            for (int c = 0; c < NS; c++) cur[c] = col[c * LW];
#pragma unroll 1
            for (int g = 0; g < TF / NS; g++) {
                if (g + 1 < TF / NS) {
#pragma unroll
                    for (int c = 0; c < NS; c++) nxt[c] = col[((g + 1) * NS + c) * LW];
                }
                if constexpr (HasTileOf<P>::value) {
                    In xin[NS];
                    Out yo[NS];
#pragma unroll
                    for (int c = 0; c < NS; c++) xin[c] = __builtin_bit_cast(In, cur[c]);
                    tile_of<P, 1, NS>(prm, &p, xin, yo);
#pragma unroll
                    for (int c = 0; c < NS; c++) col[(g * NS + c) * LW] = __builtin_bit_cast(uint32_t, yo[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < NS; c++) col[(g * NS + c) * LW] = one(cur[c]);
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int c = 0; c < NS; c++) cur[c] = nxt[c];
            }
        }
    }
    lds_barrier();  // barrier ntiles
    if (active) p.store(prm, st, slanes, lane0 + lid);
}

// ------------------------------------------------------------- FRAME_MAJOR, two waves per 64 lanes
// A serial chain of sections split over TWO waves (round 3).  A VALU-bound processor at one wave per SIMD — 65536 lanes are
// 1024 waves — runs at the issue rate of a single wave, and for the i32 sections that is one v_mad_i64_i32 per ~10 cycles
// where two waves on the SIMD get one per 5 (tools/ubench_valu.hip): a 4-section i32 chain sits at 0.58 of the HBM peak,
// the 8-section cascade at 0.47, both far from memory-bound.  Here a workgroup of two waves owns 64 lanes: wave 0 runs the
// first part of the chain (processor PA) on the input and leaves its output in an LDS tile of T frames, wave 1 runs the
// rest (PB) one tile behind and writes y; one workgroup barrier per tile, tiles double-buffered.  Twice the waves per SIMD
// and every lane still one thread per wave — and 8 sections are ONE pass over HBM instead of two.  Results are those of
// the single-wave kernels: both halves run the same per-sample functors on the same sample sequences.
// In place (y == x): wave 0 reads frame f at least a tile before wave 1 writes it.
template <class PA, class PB, int T = 32>
__global__ __launch_bounds__(2 * kWave) void stream_frame_major_duo(
    const typename PA::Params prmA, const typename PB::Params prmB, uint32_t *stA, uint32_t *stB, const typename PA::In *x,
    typename PB::Out *y, const size_t lanes, const size_t frames, const size_t xl, const size_t yl)
{
    using In = typename PA::In;
    using Mid = typename PA::Out;
    using Out = typename PB::Out;
    static_assert(std::is_same<Mid, typename PB::In>::value && PA::LDS_WORDS == 0 && PB::LDS_WORDS == 0 && PA::IN_DIV == 1 && PB::IN_DIV == 1,
                  "PB continues PA's sample stream; table-free processors");
    __shared__ Mid tile[2][T][kWave];
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) / kWave), lid = int(threadIdx.x) % kWave;
    const size_t lane = size_t(blockIdx.x) * kWave + lid;
    const bool active = lane < lanes;
    const size_t la = active ? lane : lanes - 1;  // idle threads of the last workgroup shadow a valid lane, stores masked
    const size_t nfull = frames / T;
    const int ntail = int(frames - nfull * T);  // frames of the last, partial tile
    if (wave == 0) {
        PA p;
        p.load(prmA, stA, lanes, la);
        const In *xp = x + la;
        In cur[T], nxt[T];
        // whole tiles: no per-frame predicates (wave-uniform runtime tests per frame fragment the loop into hundreds of blocks)
        auto fetch = [&](size_t k, In (&dst)[T], auto full) __attribute__((always_inline)) {
#pragma unroll
            for (int f = 0; f < T; f++)
                if (decltype(full)::value || f < ntail) dst[f] = nt_load<true>(xp + (k * T + f) * xl);
        };
        if (nfull)
            fetch(0, cur, std::true_type{});
        else
            fetch(0, cur, std::false_type{});
        for (size_t k = 0; k < nfull; k++) {
            if (k + 1 < nfull)
                fetch(k + 1, nxt, std::true_type{});
            else if (ntail)
                fetch(k + 1, nxt, std::false_type{});
#pragma unroll
            for (int f = 0; f < T; f++) tile[k & 1][f][lid] = p.step(prmA, cur[f]);
#pragma unroll
            for (int f = 0; f < T; f++) cur[f] = nxt[f];
            lds_barrier();  // tile k complete; wave 1 has finished tile k - 1 (it may be overwritten two tiles on)
        }
        if (ntail) {
            for (int f = 0; f < ntail; f++) tile[nfull & 1][f][lid] = p.step(prmA, cur[f]);
            lds_barrier();
        }
        lds_barrier();  // pairs with wave 1's last interval
        if (active) p.store(prmA, stA, lanes, lane);
    } else {
        PB p;
        p.load(prmB, stB, lanes, la);
        Out *yp = y + la;
        lds_barrier();  // tile 0
        for (size_t k = 0; k < nfull; k++) {
            Mid v[T];
#pragma unroll
            for (int f = 0; f < T; f++) v[f] = tile[k & 1][f][lid];
#pragma unroll
            for (int f = 0; f < T; f++) {
                // (no `if (active)`: an idle thread of the last workgroup shadows lane `lanes - 1` — same input, same state, the same
                // value to the same address — and a predicate per store is a branch per sample in the ISA)
                nt_store<true>(yp + (k * T + f) * yl, p.step(prmB, v[f]));
            }
            lds_barrier();
        }
        if (ntail) {
            for (int f = 0; f < ntail; f++) {
                const Out o = p.step(prmB, tile[nfull & 1][f][lid]);
                if (active) nt_store<true>(yp + (nfull * T + f) * yl, o);
            }
            lds_barrier();
        }
        if (active) p.store(prmB, stB, lanes, lane);
    }
}

// ------------------------------------------------------------- FRAME_MAJOR, one to three lanes
// The last lanes % 4 lanes of a call whose rows do not hold a whole number of 16-byte pieces (launch_stream runs them beside the
// rest on a second stream).  One thread per lane on the register-window kernel is a chain of frames at ~90 ns each — the load
// latency over the ring depth: 0.36-0.42 ms for 4096 frames, longer than the whole 65536-lane body beside it and three times
// a 16384-lane body.  Here ONE wave moves the samples of 256 frames per tile with all 64 threads (thread t: frames t, t + 64,
// t + 128, t + 192 of each lane), hands them over through LDS, threads 0 .. n - 1 walk their lane's 256 samples from there
// (the next tile's loads in flight), and all 64 threads store the results.
template <class P>
__global__ __launch_bounds__(kWave) void stream_frame_major_few(
    const typename P::Params prm, uint32_t *st, const typename P::In *x, typename P::Out *y,
    const int n, const size_t frames, const size_t xl, const size_t yl, const size_t slanes)
{
    using In = typename P::In;
    using Out = typename P::Out;
    static_assert(P::HAS_IN && P::IN_DIV == 1 && sizeof(In) == 4 && sizeof(Out) == 4, "one 4-byte sample in, one out");
    constexpr int K = 4, TF = K * kWave, ML = 3;
    __shared__ uint32_t tin[TF][4], tout[TF][4];
    __shared__ uint32_t ptab[P::LDS_WORDS ? P::LDS_WORDS : 1];
    const int t = threadIdx.x;
    if constexpr (P::LDS_WORDS > 0) {
        P::fill_shared(ptab, t, kWave);
        lds_wave_sync();
    }
    P p;
    if constexpr (P::LDS_WORDS > 0) p.set_shared(ptab);
    const bool own = t < n;
    if (own) p.load(prm, st, slanes, size_t(t));
    uint32_t cur[K][ML] = {};
    const size_t ntiles = (frames + TF - 1) / TF;
    auto fetch = [&](size_t tile) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const size_t f = tile * TF + size_t(k) * kWave + t;
            if (f < frames) {
#pragma unroll
                for (int j = 0; j < ML; j++)
                    if (j < n) cur[k][j] = __builtin_bit_cast(uint32_t, x[f * xl + j]);
            }
        }
    };
    fetch(0);
    for (size_t i = 0; i < ntiles; i++) {
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
            for (int j = 0; j < ML; j++) tin[k * kWave + t][j] = cur[k][j];
        lds_wave_sync();
        if (i + 1 < ntiles) fetch(i + 1);
        if (own) {
            const size_t left = frames - i * TF;
            const int nf = left < size_t(TF) ? int(left) : TF;
            int f = 0;
            for (; f + 8 <= nf; f += 8) {
                uint32_t v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = tin[f + u][t];
#pragma unroll
                for (int u = 0; u < 8; u++) tout[f + u][t] = __builtin_bit_cast(uint32_t, step1(p, prm, __builtin_bit_cast(In, v[u])));
            }
            for (; f < nf; f++) tout[f][t] = __builtin_bit_cast(uint32_t, step1(p, prm, __builtin_bit_cast(In, tin[f][t])));
        }
        lds_wave_sync();
#pragma unroll
        for (int k = 0; k < K; k++) {
            const size_t f = i * TF + size_t(k) * kWave + t;
            if (f < frames) {
#pragma unroll
                for (int j = 0; j < ML; j++)
                    if (j < n) y[f * yl + j] = __builtin_bit_cast(Out, tout[k * kWave + t][j]);
            }
        }
        lds_wave_sync();
    }
    if (own) p.store(prm, st, slanes, size_t(t));
}

// a size from the environment (diagnostic switches: honoured only with IDSP_DIAG=1)
inline size_t diag_size(const char *name, size_t dflt)
{
    const char *e = diag_env(name);
    return e ? size_t(strtoull(e, nullptr, 10)) : dflt;
}

}  // namespace idsp
#include "fm_sweep.h"  // the dense-sweep FrameMajor kernel (round 5): needs everything above, is needed by launch_stream below
namespace idsp {

// --------------------------------------------------------------------- launch
// Prefetch depth by occupancy: at <= 2 waves/SIMD nothing else hides HBM
// latency, so go deep; with many resident waves keep the register budget low.
// smallest launch (in waves) that takes the 256-thread LDS-DMA kernel (IDSP_DIAG=1 IDSP_LDS_MIN_WAVES overrides):
// measured equal to the single-wave register kernel at 16384 lanes, 15-20 % ahead at 32768-49152,
// slightly behind at 8192
inline size_t lds_min_waves()
{
    static const size_t v = diag_size("IDSP_LDS_MIN_WAVES", thr::kLdsMinWaves);
    return v;
}

// Row pitches of a call: elements between consecutive frames (FRAME_MAJOR) or between the starts of consecutive
// lanes (LANE_MAJOR) of x and y; 0 = dense.  The `_pitch` entry points of the ABI pass them through.
struct Pitch {
    size_t x = 0, y = 0;
};

// Parameters of a lane block that starts `first` lanes into the call: per-lane coefficient planes (ByLaneParams::coef) move
// with the lanes, everything else is shared
template <class T, class = void>
struct HasCoefPlanes : std::false_type {};
template <class T>
struct HasCoefPlanes<T, std::void_t<decltype(std::declval<T>().coef)>> : std::true_type {};
// (parameters that carry a per-lane side pointer — lockin_generic.hip's XWalk `xw`, computed from the lane index of the WHOLE call —
// cannot be shifted here: launch_stream's lane splits only take 4-byte-in / 4-byte-out processors, and this assert keeps it so)
template <class T, class = void>
struct HasSideWalk : std::false_type {};
template <class T>
struct HasSideWalk<T, std::void_t<decltype(std::declval<T>().xw)>> : std::true_type {};
template <class Params>
inline Params shift_lanes(Params p, size_t first, size_t elem)
{
    static_assert(!HasSideWalk<Params>::value, "lane split of a processor whose parameters hold a side pointer (XWalk): shift it here first");
    if constexpr (HasCoefPlanes<Params>::value) p.coef = static_cast<const char *>(p.coef) + first * elem;
    return p;
}

// Largest remainder (in lanes) that launch_stream runs beside the whole rounds on a second stream (tools/exp_split_streams.py:
// 73728 lanes 0.58 -> 0.69 of the HBM peak with an 8192-lane remainder, 81920 0.59 -> 0.63 with 16384, no gain from 24576 up)
constexpr size_t kSplitTailMax = thr::kSplitTailMax;

// set while launch_stream launches the remainder of a "whole rounds + remainder" split on the second stream: that launch keeps the staged kernel
// (beside the sweep kernel's whole rounds the pair kernel's remainder costs 6-10 %: 69632 lanes 0.413 -> 0.454 ms, profiles/r06_exp_fm_pair.txt)
inline bool &split_remainder_flag()
{
    static thread_local bool f = false;
    return f;
}

template <class P>
int launch_stream(const typename P::Params &prm, void *state, const typename P::In *x,
                  typename P::Out *y, size_t lanes, size_t frames, int layout, hipStream_t s, Pitch pitch = {}, size_t state_pitch = 0)
{
    if (lanes == 0) return IDSP_OK;
    uint32_t *st = static_cast<uint32_t *>(state);
    const size_t sp = state_pitch ? state_pitch : lanes;  // lanes between the state planes (a lane block of a larger call keeps the call's)
    if (layout == IDSP_LANE_MAJOR) {
        if (sp != lanes) return fail(IDSP_EINVAL, "internal: lane block with a state pitch on a LaneMajor kernel");
        const size_t xl = pitch.x ? pitch.x : frames, yl = pitch.y ? pitch.y : frames;
        const unsigned grid = unsigned((lanes + kWave - 1) / kWave);
        if constexpr (LmStagedOf<P>::value) {
            // 16-byte pieces need 16-byte aligned rows (of less than 64 MiB: 32-bit offsets inside a wave's 64 rows); below a
            // quarter tile of frames the 4-byte tile kernel has less to set up
            // (IDSP_DIAG=1 IDSP_NO_LM_STAGED=1: always the tile kernel)
            static const bool no_staged = diag_env("IDSP_NO_LM_STAGED") != nullptr;
            constexpr size_t isz = P::HAS_IN ? sizeof(typename P::In) : 0, osz = sizeof(typename P::Out), wide = isz > osz ? isz : osz;
            const bool x_ok = !P::HAS_IN || (reinterpret_cast<uintptr_t>(x) % 16 == 0 && (xl * isz) % 16 == 0 && xl * isz < (size_t(1) << 26));
            if (!no_staged && frames * wide >= size_t(kLmRun) / 4 && x_ok && reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
                (yl * osz) % 16 == 0 && yl * osz < (size_t(1) << 26)) {
                // lanes per wave: 64 when that already gives every SIMD a wave, else 32 or 16 (tools/tune_lm.hip;
                // IDSP_DIAG=1 IDSP_LM_LANES_PER_WAVE = 64 / 32 / 16 forces one)
                static const size_t forced_lw = diag_size("IDSP_LM_LANES_PER_WAVE", 0);
                // measured (profiles/r02_tune_lm_lanes_per_wave.jsonl, 4096 frames): i32 DF1 at 16384 lanes 0.168 / 0.157 / 0.136 ms
                // with 64 / 32 / 16 lanes per wave, at 32768 lanes 0.232 / 0.216 / 0.213, at 65536 0.382 / 0.393 / 0.446; the
                // 8-section cascade (VALU-bound: half-empty waves cost arithmetic) 0.57 / 0.45 / 0.52 at 16384 and
                // 0.61 / 0.91 / 1.56 at 65536
                constexpr bool heavy = P::COST > 120;
                const size_t lw = forced_lw ? forced_lw : lanes >= thr::kLmStaged64Lanes ? 64 : (lanes >= thr::kLmStaged32Lanes || heavy) ? 32 : 16;
                auto go = [&](auto lw_tag) {
                    constexpr int LW = decltype(lw_tag)::value;
                    // the 32- and 16-lane forms move 1 KiB per lane and tile (the same 32 / 16 KiB slot and staging registers
                    // as 512-byte runs would need at twice the lanes): 5-10 % faster at 16384 lanes and below, 10-20 % on rows
                    // that are not 128-byte aligned, equal at 65536 lanes (profiles/r02_tune_lm_run1k.jsonl)
                    constexpr int LB = LW == 64 ? kLmRun : 2 * kLmRun;
                    constexpr size_t bytes = lm_staged_lds_bytes<P, LW, LB>();
                    if (int rc = ensure_dyn_lds<&stream_lane_major_staged<P, LW, LB>>(bytes)) return rc;
                    note_kernel(LW == 64 ? "stream_lane_major_staged" : LW == 32 ? "stream_lane_major_staged[32 lanes/wave]" : "stream_lane_major_staged[16 lanes/wave]",
                                typeid(P).name());
                    static const unsigned lm_skew = unsigned(diag_size("IDSP_LM_SKEW", 0)),
                                          lm_skew_shift = (unsigned(diag_size("IDSP_LM_SKEW_SHIFT", 8)) & 31u) | ((unsigned(diag_size("IDSP_LM_SKEW_MOD", 2)) & 255u) << 8);
                    hipLaunchKernelGGL((stream_lane_major_staged<P, LW, LB>), dim3(unsigned((lanes + LW - 1) / LW)), dim3(kWave), bytes, s, prm, st, x, y,
                                       lanes, frames, xl, yl, lm_skew, lm_skew_shift);
                    return launch_status();
                };
                // forms instantiated per processor: all three for the cheap ones, 64 / 32 for the heavy ones (never 16 above),
                // 64 only for processors that ask for it (P::LM_ONE_FORM: rarely taken fall-back paths; build time)
                if constexpr (!LmOneFormOf<P>::value) {
                    if constexpr (!heavy) {
                        if (lw == 16) return go(std::integral_constant<int, 16>{});
                    }
                    if (lw <= 32) return go(std::integral_constant<int, 32>{});
                }
                return go(std::integral_constant<int, 64>{});
            }
        }
        note_kernel("stream_lane_major", typeid(P).name());
        hipLaunchKernelGGL((stream_lane_major<P>), dim3(grid), dim3(kWave), 0, s, prm, st, x, y, lanes, frames, xl, yl);
    } else {
        const size_t xl = pitch.x ? pitch.x : lanes / P::IN_DIV, yl = pitch.y ? pitch.y : lanes;
        const size_t waves = (lanes + kWave - 1) / kWave;
        // Whole rounds + remainder (round 3).  The LDS-DMA kernel runs one 256-lane block per CU and round; 65540 lanes are 257
        // blocks — one CU with two workgroups, both at half speed — and 73728 lanes 288: 0.58 of the HBM peak where 65536 run at
        // 0.78.  The lanes beyond the last whole round of 256 blocks therefore run BESIDE the whole rounds, on a second stream,
        // on the staged single-wave kernel (its waves spread over all CUs and fit next to the LDS-DMA workgroups): 69632 lanes
        // 0.57 -> 0.65, 73728 0.58 -> 0.69, 81920 0.59 -> 0.63, 147456 0.59 -> 0.65; no gain from a 24576-lane remainder up
        // (tools/exp_split_streams.py, profiles/r03_exp_split_streams.jsonl).  Both pieces are lane blocks of the caller's
        // tensors (row pitches xl / yl, state and coefficient planes at the call's pitch).
        // Rows that are only 4-byte aligned (round 3): a dense tensor of 65537 lanes, an odd pitch, a base pointer off the 16-byte grid.
        // `global_load_lds_dwordx4`, `global_load_dwordx4` and the 16-byte stores only need dword alignment, so the LDS-DMA kernel
        // and the staged kernel run on such rows as they are (tools/exp_fm_unaligned4.hip, profiles/r03_exp_fm_unaligned4.jsonl: bit for
        // bit the register-window kernel's output; 65536 lanes at pitch 65537 0.65 of the HBM peak with XCD-contiguous blocks against 0.41
        // for the register-window kernel's 4-byte accesses).  What they do need is whole 16-byte pieces per row, i.e. a lane count that
        // is a multiple of 4: the last lanes % 4 lanes run beside the rest on the second stream (stream_frame_major_few).
        // (IDSP_DIAG=1 IDSP_ALIGN16_ONLY=1: round 2's rule — 16-byte aligned rows or the register-window kernel)
        static const bool align16_only = diag_env("IDSP_ALIGN16_ONLY") != nullptr;
        const bool on_grid16 = reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0 && (xl * sizeof(typename P::In)) % 16 == 0 &&
                               (yl * sizeof(typename P::Out)) % 16 == 0;
        const bool rows_ok = on_grid16 || !align16_only;
        if constexpr (P::HAS_IN && P::IN_DIV == 1 && sizeof(typename P::In) == 4 && sizeof(typename P::Out) == 4) {
            static const bool no_few = diag_env("IDSP_NO_FM_FEW") != nullptr;
            if (lanes <= 3 && frames >= 64 && !no_few) {  // one to three lanes in all: the same kernel, on the caller's stream
                note_kernel("stream_frame_major_few", typeid(P).name());
                hipLaunchKernelGGL((stream_frame_major_few<P>), dim3(1), dim3(kWave), 0, s, prm, st, x, y, int(lanes), frames, xl, yl, sp);
                return launch_status();
            }
            const size_t odd = lanes % 4, body = lanes - odd;
            if (odd && !align16_only && body >= 8192 && frames >= 16 && (FmStagedOf<P>::value || LdsEligibleOf<P>::value)) {
                if (SideStream *ss = side_stream(s)) {
                    IDSP_HIP_TRY(hipEventRecord(ss->fork, s));
                    IDSP_HIP_TRY(hipStreamWaitEvent(ss->stream, ss->fork, 0));
                    hipLaunchKernelGGL((stream_frame_major_few<P>), dim3(1), dim3(kWave), 0, ss->stream, shift_lanes(prm, body, sizeof(typename P::In)), st + body,
                                       x + body, y + body, int(odd), frames, xl, yl, sp);
                    int rc = launch_status();
                    if (rc == IDSP_OK) rc = launch_stream<P>(prm, st, x, y, body, frames, layout, s, Pitch{xl, yl}, sp);
                    IDSP_HIP_TRY(hipEventRecord(ss->join, ss->stream));
                    IDSP_HIP_TRY(hipStreamWaitEvent(s, ss->join, 0));
                    if (rc == IDSP_OK) note_kernel_also(" + stream_frame_major_few (lanes % 4, second stream)");
                    return rc;
                }
            }
        }
        if constexpr (FmStagedOf<P>::value && P::HAS_IN && P::IN_DIV == 1 && sizeof(typename P::In) == 4 && sizeof(typename P::Out) == 4 && P::COST <= 120) {
            const size_t head = lanes / (size_t(256) * kFmBlock) * (size_t(256) * kFmBlock), tail = lanes - head;
            // (round 5: only when the whole rounds are 1, 2, 4, 8 or 16 times 65536 lanes — FULL 256-lane blocks on the sweep kernel; three
            // or five rounds would be narrow blocks themselves, and the call is one sweep over all its lanes instead)
            const size_t head_rounds = head / (size_t(256) * kFmBlock);
            // ... a restriction of the SWEEP kernel's geometry: a processor it does not take at `head` lanes (SWEEP_MAX_LPT 2 or 4: cascades, chains of
            // three sections and more) keeps round 3's split for any number of whole rounds (3 x 65536 + 8192 lanes of a 3-section chain: 0.65 against
            // 0.57 as one launch with a ragged last round, profiles/r03_exp_split_streams.jsonl)
            bool head_on_sweep = false;
            if constexpr (LdsEligibleOf<P>::value) head_on_sweep = head != 0 && sweep_takes<P>(head);
            const bool rounds_ok = !head_on_sweep || ((head_rounds & (head_rounds - 1)) == 0 && head_rounds <= 16);
            if (head && tail && tail <= kSplitTailMax && rounds_ok && lanes % 4 == 0 && frames >= 16 && LdsEligibleOf<P>::value && !diag_on() && rows_ok &&
                xl * 4 < (size_t(1) << 28) && yl * 4 < (size_t(1) << 28)) {
                if (SideStream *ss = side_stream(s)) {
                    IDSP_HIP_TRY(hipEventRecord(ss->fork, s));
                    IDSP_HIP_TRY(hipStreamWaitEvent(ss->stream, ss->fork, 0));
                    int rc = launch_stream<P>(prm, st, x, y, head, frames, layout, s, Pitch{xl, yl}, sp);
                    // which kernel the whole rounds went to: the name that launch recorded (the dense-sweep kernel of fm_sweep.h, or the LDS-DMA
                    // panel walk for the processors / shapes the sweep does not take)
                    const char *const head_name = noted_kernel();
                    const bool head_swept = head_name && std::strncmp(head_name, "stream_frame_major_sweep", 24) == 0;
                    if (rc == IDSP_OK) {
                        split_remainder_flag() = true;
                        rc = launch_stream<P>(shift_lanes(prm, head, sizeof(typename P::In)), st + head, x + head, y + head, tail, frames, layout, ss->stream,
                                              Pitch{xl, yl}, sp);
                        split_remainder_flag() = false;
                    }
                    // join even after a failed launch: the caller's stream must not run ahead of whatever the side stream holds
                    IDSP_HIP_TRY(hipEventRecord(ss->join, ss->stream));
                    IDSP_HIP_TRY(hipStreamWaitEvent(s, ss->join, 0));
                    if (rc == IDSP_OK)
                        note_kernel(head_swept ? "stream_frame_major_sweep + stream_frame_major_staged (remainder, second stream)"
                                               : "stream_frame_major_lds + stream_frame_major_staged (remainder, second stream)",
                                    typeid(P).name());
                    return rc;
                }
            }
        }
        if constexpr (FmStagedOf<P>::value && P::HAS_IN && P::IN_DIV == 1 && sizeof(typename P::In) == 4 && sizeof(typename P::Out) == 4 && P::COST <= thr::kPairMaxCost) {
            // Few lanes, long calls, cheap single sections: one wave computes, one wave moves (stream_frame_major_pair above).
            // (IDSP_DIAG=1 IDSP_NO_FM_PAIR=1: the staged single-wave kernel; IDSP_FM_PAIR_MAX_LANES=n: another upper lane count)
            static const bool no_pair = diag_env("IDSP_NO_FM_PAIR") != nullptr;
            static const size_t pair_max = diag_size("IDSP_FM_PAIR_MAX_LANES", thr::kPairMaxLanes);
            // rows on the 64-byte grid only: off it the staged kernel's plain accesses and XCD-contiguous order (round 4) win — 16384 lanes of a 16385-lane
            // tensor 0.315 ms here against 0.186 there
            const bool grid64 = reinterpret_cast<uintptr_t>(x) % 64 == 0 && reinterpret_cast<uintptr_t>(y) % 64 == 0 && (xl * 4) % 64 == 0 && (yl * 4) % 64 == 0;
            if (!no_pair && !split_remainder_flag() && lanes <= pair_max && frames >= thr::kPairMinFrames && lanes % 4 == 0 && grid64 && xl * 4 < (size_t(1) << 28) && yl * 4 < (size_t(1) << 28)) {
                constexpr size_t bytes = 2 * size_t(kFmPairTile) + size_t(P::LDS_WORDS) * 4;
                if (int rc = ensure_dyn_lds<&stream_frame_major_pair<P>>(bytes)) return rc;
                note_kernel("stream_frame_major_pair[compute + mover wave per 32 lanes]", typeid(P).name());
                hipLaunchKernelGGL((stream_frame_major_pair<P>), dim3(unsigned((lanes + 31) / 32)), dim3(2 * kWave), bytes, s, prm, st, x, y, lanes, frames, xl, yl, sp);
                return launch_status();
            }
        }
        if constexpr (P::HAS_IN && P::IN_DIV == 1 && sizeof(typename P::In) == 4) {
            // Round 5: ONE dense sweep for any lane count (fm_sweep.h).  Every lane count from kSweepMinLanes up on rows that sit on the
            // 64-byte grid goes here — whole multiples of 65536, ragged counts, 2^20 lanes alike — instead of panel walks of a persistent
            // grid or a ragged last block.  (Lane counts a little above a multiple of 65536 were split above: their whole rounds come back
            // here as full 256-lane blocks, the remainder runs beside them — 65552 lanes: 0.77 of the HBM peak against 0.54 as one sweep of
            // half-empty blocks.)  (IDSP_DIAG=1 IDSP_NO_SWEEP=1: the round-4
            // dispatch below.)  Rows off the 64-byte grid (dword alignment is all the requests and stores need): the same sweep with the
            // blocks dealt to the XCDs in contiguous eighths.
            static const bool no_sweep = diag_env("IDSP_NO_SWEEP") != nullptr;
            static const size_t sweep_min = diag_size("IDSP_SWEEP_MIN_LANES", sizeof(typename P::Out) == 4 ? kSweepMinLanesFps : kSweepMinLanes);
            static const bool cost_forced_ = diag_env("IDSP_LDS_COST") != nullptr, no_lds_ = diag_env("IDSP_NO_LDS_PATH") != nullptr;
            static const bool grid64_only = diag_env("IDSP_SWEEP_GRID64_ONLY") != nullptr;  // IDSP_DIAG=1: rows off the 64-byte grid stay on round 3's kernel
            const bool on_grid64 = reinterpret_cast<uintptr_t>(x) % 64 == 0 && reinterpret_cast<uintptr_t>(y) % 64 == 0 && (xl * sizeof(typename P::In)) % 64 == 0 &&
                                   (yl * sizeof(typename P::Out)) % 64 == 0;
            if constexpr (LdsEligibleOf<P>::value) {
                // rows off the grid: from the lane count up where round 3's XCD-contiguous kernel would walk panels on a persistent grid (131076 dense
                // lanes 0.56 -> 0.61, 100004 0.58 -> 0.65); below it that kernel is a single round itself and stays (65000 dense lanes 0.745)
                static const size_t off_grid_min = diag_size("IDSP_SWEEP_OFFGRID_MIN_LANES", kLdsGridCap * size_t(kFmBlock));  // IDSP_DIAG=1
                // ... and, 4-byte outputs, up to kSweepOffGridSmallMax lanes, where the alternative is the staged single-wave kernel (several frames per segment here: 40000 lanes at
                // pitch + 4 0.51 -> 0.72 of the peak, 32768 0.52 -> 0.61, 49152 0.59 -> 0.69; 57344: 0.69 either way)
                const bool off_grid_ok = !grid64_only && rows_ok && (lanes > off_grid_min || (sizeof(typename P::Out) == 4 && lanes <= thr::kSweepOffGridSmallMax));
                if (!no_sweep && !cost_forced_ && !no_lds_ && (on_grid64 || off_grid_ok) && lanes % 4 == 0 && lanes >= sweep_min && frames >= thr::kSweepMinFrames &&
                    sweep_takes<P>(lanes))
                    return launch_sweep<P>(prm, st, x, y, lanes, frames, xl, yl, sp, s);
            }
        }
        if constexpr (FmStagedOf<P>::value) {
            // Few lanes (the chip is not full and every FrameMajor kernel below runs at its per-lane latency): the staged
            // single-wave kernel.  Measured against the register-window and the LDS-DMA kernel (tools/tune_fm_small.hip,
            // profiles/r02_tune_fm_small.jsonl; i32 DF1 x 4096 frames): 16384 lanes 0.147 ms (32 lanes per wave) against
            // 0.211 / 0.335; 32768 lanes 0.211 (64) against 0.243 / 0.332; 8192 x 8192 0.247 against 0.384; 49152 lanes 0.32
            // against 0.33 / 0.33; 65536 lanes 0.43 against 0.41 / 0.335.  Processors with COST > 120 gain only around
            // 16384-32768 lanes (8-section cascade 0.44 against 0.48) and lose elsewhere.
            // (IDSP_DIAG=1 IDSP_NO_FM_STAGED=1: never; IDSP_FM_LANES_PER_WAVE = 64 / 32 / 16 forces the form at any lane count)
            static const bool no_fm_staged = diag_env("IDSP_NO_FM_STAGED") != nullptr;
            static const size_t forced_flw = diag_size("IDSP_FM_LANES_PER_WAVE", 0);
            constexpr size_t sz = sizeof(typename P::In);
            constexpr bool heavy = P::COST > 120;
            const bool in_range = heavy ? (lanes >= thr::kStagedHeavyMinLanes && lanes < thr::kStagedHeavyMaxLanes) : lanes < thr::kStagedMaxLanes;
            if (!no_fm_staged && (forced_flw || in_range) && frames >= 16 && (lanes * sz) % 16 == 0 && rows_ok &&
                xl * sz < (size_t(1) << 28) && yl * sz < (size_t(1) << 28)) {  // 32-bit offsets: up to 15 row pitches + 1 KiB inside a tile (16 lanes/wave)
                // rows off the 64-byte grid: a 32-lane wave's 128-byte row pieces each straddle two lines; whole 256-byte pieces from 12288
                // lanes (16384 lanes at pitch 16385: 0.19 ms with 64 lanes per wave, 0.27 with 32, 0.23 on the register-window kernel;
                // 8192 lanes 0.175 / 0.147 / 0.21 — tools/exp_fm_unaligned_small.py, profiles/r03_exp_fm_unaligned_small.jsonl)
                const bool off64 = (xl * sz) % 64 != 0 || (yl * sz) % 64 != 0 || reinterpret_cast<uintptr_t>(x) % 64 != 0 || reinterpret_cast<uintptr_t>(y) % 64 != 0;
                // (round 4, with plain accesses on such rows the 32-lane form wins again up to ~25000 lanes: 16385 lanes 0.217 -> 0.182 ms,
                // 12292 0.202 -> 0.163, 20484 0.240 -> 0.200, 24580 0.249 -> 0.223; 26628 0.255 with 64 against 0.268 with 32)
                const size_t lw = forced_flw ? forced_flw : heavy ? 32 : lanes >= (off64 ? thr::kStaged64LanesOffGrid : thr::kStaged64Lanes) ? 64 : lanes >= thr::kStaged32Lanes ? 32 : 16;
                auto go = [&](auto lw_tag) {
                    constexpr int LW = decltype(lw_tag)::value;
                    constexpr size_t bytes = size_t(kFmStagedTile) + size_t(P::LDS_WORDS) * 4;
                    if (int rc = ensure_dyn_lds<&stream_frame_major_staged<P, LW>>(bytes)) return rc;
                    note_kernel(LW == 64 ? "stream_frame_major_staged[64 lanes/wave]" : LW == 32 ? "stream_frame_major_staged[32 lanes/wave]" : "stream_frame_major_staged[16 lanes/wave]",
                                typeid(P).name());
                    // rows off the 64-byte grid: XCD-contiguous lane blocks (IDSP_DIAG=1 IDSP_STAGED_NO_XCDC=1: the plain order)
                    static const bool no_xcdc = diag_env("IDSP_STAGED_NO_XCDC") != nullptr;
                    const unsigned grid = unsigned((lanes + LW - 1) / LW);
                    // ... and plain instead of nontemporal accesses there (bit 1; IDSP_DIAG=1 IDSP_STAGED_NT=1: nontemporal everywhere)
                    static const bool all_nt = diag_env("IDSP_STAGED_NT") != nullptr;
                    const int xcdc = (!no_xcdc && grid >= 64 && off64 ? 1 : 0) | (!all_nt && off64 ? 2 : 0);
                    hipLaunchKernelGGL((stream_frame_major_staged<P, LW>), dim3(grid), dim3(kWave), bytes, s, prm, st, x, y, lanes, frames, xl, yl, sp, xcdc);
                    return launch_status();
                };
                if constexpr (!heavy) {
                    if (lw == 16) return go(std::integral_constant<int, 16>{});
                    if (lw == 64) return go(std::integral_constant<int, 64>{});
                }
                return go(std::integral_constant<int, 32>{});
            }
        }
        if constexpr (P::HAS_IN && P::IN_DIV == 1 && sizeof(typename P::In) == 4) {
            // cheap per-sample arithmetic (the extra LDS hop and the two barriers per tile cost issue
            // slots), whole 256-lane blocks, 16-byte aligned rows: LDS-DMA path
            // Which processors take the LDS-DMA path: P::LDS_ELIGIBLE where the processor says so (measured per section type
            // and section count against the register-window kernel, tools/tune_lds.hip, profiles/r02_tune_lds_heavy*.jsonl),
            // else the cost estimate; IDSP_DIAG=1 IDSP_LDS_COST=n replaces both by `COST <= n`.
            static const bool cost_forced = diag_env("IDSP_LDS_COST") != nullptr;
            static const int lds_cost_max = int(diag_size("IDSP_LDS_COST", 120));
            const bool eligible = cost_forced ? P::COST <= lds_cost_max : LdsEligibleOf<P>::value;
            static const size_t lds_max_waves = diag_size("IDSP_LDS_MAX_WAVES", size_t(1) << 40);
            static const bool no_lds = diag_env("IDSP_NO_LDS_PATH") != nullptr;
            constexpr size_t ow = sizeof(typename P::Out) / 4;
            // whole 256-lane blocks — or, for one-word outputs, any multiple of 4 lanes: the kernel's last block may be ragged
            // (reference: any N in `Lanes<C>`, dsp-process/src/compose.rs:468)
            if (!no_lds && eligible && waves >= lds_min_waves() && waves <= lds_max_waves && (lanes % kFmBlock == 0 || (ow == 1 && lanes % 4 == 0)) && rows_ok) {
                // Grid (profiles/r02_exp_c5_*.jsonl).  Up to 384 workgroups: one per 256-lane block.  Beyond: a persistent
                // grid of <= 256 workgroups (one per CU) that walks the lane blocks in column panels of equal rounds —
                // 2^20 lanes: 0.68 of peak (0.61-0.75 by output placement) against 0.66 with 4096 workgroups and 0.615
                // on the register-window kernel; 131072 lanes: 0.72 against 0.65 with 512 workgroups.  Two or four
                // lanes per thread (whole-row sweeps by 256 workgroups) are available to processors that declare
                // P::LDS_LPT_MAX; none does (see the kernel).  IDSP_DIAG=1 IDSP_LDS_LPT / IDSP_LDS_GRID override both.
                static const size_t forced_lpt = diag_size("IDSP_LDS_LPT", 0);
                static const size_t forced_grid = diag_size("IDSP_LDS_GRID", ~size_t(0));
                const size_t nblocks = (lanes + kFmBlock - 1) / kFmBlock;
                constexpr int kMaxLpt = LdsLptMaxOf<P>::value;
                int lpt = 1;
                if (lanes % kFmBlock != 0) {
                    // ragged last block: one lane per thread only
                } else if (forced_lpt) {
                    lpt = int(forced_lpt) > kMaxLpt ? kMaxLpt : int(forced_lpt);
                    while (lpt > 1 && nblocks % size_t(lpt) != 0) lpt >>= 1;
                } else {
                    for (int c = kMaxLpt; c > 1; c >>= 1)
                        if (nblocks % size_t(c) == 0 && nblocks / size_t(c) >= 224 && nblocks / size_t(c) <= 320) {
                            lpt = c;
                            break;
                        }
                }
                const size_t wgs = nblocks / size_t(lpt);
                size_t grid = wgs;
                if (forced_grid != ~size_t(0)) {
                    if (forced_grid && wgs > forced_grid) grid = forced_grid;
                } else if (wgs > kLdsGridCap) {
                    const size_t rounds = (wgs + 255) / 256;
                    grid = (wgs + rounds - 1) / rounds;
                }
                int rc = IDSP_OK;
                // Ring depth: the processor's tuned depth for single-pass launches; persistent launches (large lane
                // counts, row pitch >= 1.5 MiB) keep 7 tiles and the indexed form for every processor — depths 3-5 lose
                // 15-20 % there (profiles/r02_exp_c5_matrix7.jsonl: 0.56 vs 0.68 at 2^20 lanes).
                const bool persistent = grid < wgs;
                auto go = [&](auto lpt_tag, auto deep) {
                    constexpr int L = decltype(lpt_tag)::value;
                    constexpr bool DEEP = decltype(deep)::value;
                    constexpr int NB = DEEP ? 7 : LdsRingOf<P>::value;
                    constexpr bool RUN = DEEP ? false : LdsRunOf<P>::value;
                    constexpr size_t ts = L > kLdsT ? L : kLdsT;
                    constexpr size_t bytes = (size_t(NB) * ts * kFmBlock + 2 * ts * kFmBlock * ow + P::LDS_WORDS) * 4;
                    if ((rc = ensure_dyn_lds<&stream_frame_major_lds<P, NB, L, RUN>>(bytes))) return;
                    note_kernel(L == 1 ? "stream_frame_major_lds" : L == 2 ? "stream_frame_major_lds[2 lanes/thread]" : "stream_frame_major_lds[4 lanes/thread]",
                                typeid(P).name());
                    hipLaunchKernelGGL((stream_frame_major_lds<P, NB, L, RUN>), dim3(unsigned(grid)), dim3(kFmBlock), bytes, s,
                                       prm, st, x, y, lanes, frames, xl, yl, sp);
                };
                using Yes = std::true_type;
                using No = std::false_type;
                if constexpr (kMaxLpt >= 4) {
                    if (lpt == 4) go(std::integral_constant<int, 4>{}, No{});
                }
                if constexpr (kMaxLpt >= 2) {
                    if (lpt == 2) go(std::integral_constant<int, 2>{}, No{});
                }
                // rows that do not start on 64-byte boundaries (dense 65000-lane tensors, odd pitches): adjacent lane blocks on one XCD
                // (IDSP_DIAG=1 IDSP_LDS_NO_XCDC=1: the plain block order, 0.58-0.64 of the peak on such rows instead of 0.67-0.71)
                static const bool no_rot = diag_env("IDSP_LDS_NO_XCDC") != nullptr;
                const bool misaligned = (xl * 4) % 64 != 0 || (yl * ow * 4) % 64 != 0 || reinterpret_cast<uintptr_t>(x) % 64 != 0 || reinterpret_cast<uintptr_t>(y) % 64 != 0;
                {
                    if (lpt == 1 && misaligned && !no_rot) {
                        constexpr size_t bytes = (size_t(7) * kLdsT * kFmBlock + 2 * kLdsT * kFmBlock * ow + P::LDS_WORDS) * 4;
                        if (int e = ensure_dyn_lds<&stream_frame_major_lds<P, 7, 1, false, true>>(bytes)) return e;
                        note_kernel("stream_frame_major_lds[XCD-contiguous blocks]", typeid(P).name());
                        hipLaunchKernelGGL((stream_frame_major_lds<P, 7, 1, false, true>), dim3(unsigned(grid)), dim3(kFmBlock), bytes, s, prm, st, x, y,
                                           lanes, frames, xl, yl, sp);
                        return launch_status();
                    }
                }
                if (lpt == 1) {
                    if (persistent && LdsRingOf<P>::value != 7)
                        go(std::integral_constant<int, 1>{}, Yes{});
                    else
                        go(std::integral_constant<int, 1>{}, No{});
                }
                if (rc) return rc;
                return launch_status();
            }
        }
        // A CU's L1 moves ~10 B/cycle, so a launch must reach all 256 CUs: below 1024 waves (= 256
        // workgroups of 4) use one wave per workgroup (16384 lanes in 256-thread blocks would run on 64 CUs).
        const unsigned block = waves < 1024 ? unsigned(kWave) : unsigned(kFmBlock);
        const unsigned grid = unsigned((lanes + block - 1) / block);
        constexpr int kDeep = MaxU<P>::value, kShallow = kDeep < 8 ? kDeep : 8;
        // XCD-contiguous lane blocks for rows off the 64-byte grid: what helps the LDS-DMA kernel HURTS this one — 65537 dense lanes
        // 0.81 ms against 0.69 with the plain order (profiles/r03_perf_ragged_xcdc.jsonl) — so it is a diagnostic switch only
        // (IDSP_DIAG=1 IDSP_FM_XCDC=1)
        static const bool want_xcdc = diag_env("IDSP_FM_XCDC") != nullptr;
        const int xcdc = want_xcdc && grid >= 64;
        note_kernel(xcdc ? "stream_frame_major[XCD-contiguous blocks]" : "stream_frame_major", typeid(P).name());
        if (waves <= 2048)
            hipLaunchKernelGGL((stream_frame_major<P, kDeep>), dim3(grid), dim3(block), 0, s, prm, st, x, y, lanes, frames, xl, yl, xcdc, sp);
        else
            hipLaunchKernelGGL((stream_frame_major<P, kShallow>), dim3(grid), dim3(block), 0, s, prm, st, x, y, lanes, frames, xl, yl, xcdc, sp);
    }
    return launch_status();
}

// Two-wave launch of a chain split into PA (first sections) and PB (the rest): FRAME_MAJOR only, 4-byte samples
template <class PA, class PB>
int launch_duo(const typename PA::Params &pa, const typename PB::Params &pb, uint32_t *stA, uint32_t *stB, const typename PA::In *x,
               typename PB::Out *y, size_t lanes, size_t frames, hipStream_t s, Pitch pitch)
{
    const size_t xl = pitch.x ? pitch.x : lanes, yl = pitch.y ? pitch.y : lanes;
    note_kernel("stream_frame_major_duo", typeid(PB).name());
    hipLaunchKernelGGL((stream_frame_major_duo<PA, PB>), dim3(unsigned((lanes + kWave - 1) / kWave)), dim3(2 * kWave), 0, s, pa, pb, stA, stB, x, y, lanes,
                       frames, xl, yl);
    return launch_status();
}

// When a chain of m sections (per launch) goes to the two-wave kernel (IDSP_DIAG=1 IDSP_NO_DUO=1: never).  Measured at 4096 frames
// (tools/exp_duo.hip, profiles/r03_exp_duo.jsonl): i32 chain of 4 at 65536 lanes 0.483 -> 0.426 ms, at 131072 lanes 0.829 -> 0.909;
// i32 cascade of 8 at 32768 / 65536 / 131072 lanes 0.583 / 0.661 / 1.250 -> 0.482 / 0.578 / 1.066; and five to eight sections of a
// plain chain are ONE pass over HBM instead of two.
inline bool duo_wanted(size_t m, size_t lanes, int layout)
{
    static const bool off = diag_env("IDSP_NO_DUO") != nullptr;
    if (off || layout != IDSP_FRAME_MAJOR || lanes < thr::kDuoMinLanes) return false;
    return m >= 5 || (m == 4 && lanes <= thr::kDuo4MaxLanes);
}

}  // namespace idsp
