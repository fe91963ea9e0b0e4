// hbf_wave.h — statically specialised half-band cascade kernels for the
// reference's built-in cascades (HBF_DEC_CASCADE / HBF_INT_CASCADE over HBF_TAPS
// and HBF_TAPS_98, src/hbf.rs:258-349,385-421,476-512): the interpolators, and the
// FRAME_MAJOR decimators on shapes hbf_ring.h does not cover (cascades other than
// /16, lane counts that are not whole 16-lane groups).  LANE_MAJOR decimators and
// /16 FRAME_MAJOR ones run on the LDS-DMA ring kernels of hbf_ring.h since round 4.
//
// Mapping: ONE WAVE per lane, no workgroup barriers.  A wave walks its lane in
// chunks of kCH = 1024 high-rate samples; every stage of the cascade runs inside
// the chunk with the inter-stage streams in the wave's private LDS region (the
// reference's `Major` scratch, dsp-process/src/compose.rs:581-593).  Because LDS
// operations of one wave execute in order, stage hand-over needs only
// `s_waitcnt lgkmcnt(0)`; waves never wait for each other, and with ~10 KiB of
// LDS and < 128 VGPRs a CU holds 15-16 of them, which is what hides the HBM and
// LDS latencies.  Stage count, tap counts, tap values, LDS offsets and the
// outputs-per-thread blocking are all compile-time, so the tap multiplies use
// literal constants and there is no per-stage control flow.
//
// Within a stage every thread produces P consecutive outputs (P = 4, 2 or 1,
// chosen so that all 64 threads have work: n/64 clamped to 1..4) from one
// aligned window read; the sample history the reference keeps with
// `copy_within` (src/hbf.rs:182-183,224) sits in front of each stream buffer.
//
// Arithmetic: exactly `get()` (src/hbf.rs:46-68): Σ_k (new_k + old_k)·tap_k
// accumulated sequentially from -0.0 from the outermost tap inwards, then the
// delayed even sample (decimator) / the centre-tap identity (interpolator).
#pragma once

#include <type_traits>
#include <typeinfo>
#include <utility>

#include "common.h"
#include "hbf_taps.h"

namespace idsp {
namespace hbfw {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kW = 64;      // threads per workgroup = one wave
#ifndef IDSP_HBF_CH
#define IDSP_HBF_CH 1024
#endif
constexpr int kCH = IDSP_HBF_CH;   // high-rate samples per chunk
constexpr int kSlack = 8;   // words readable past the last valid sample of a stream

constexpr int pad4(int h) { return (4 - h % 4) % 4; }
constexpr int up4(int v) { return (v + 3) & ~3; }
constexpr int blocking(int n) { return n >= 4 * kW ? 4 : (n >= 2 * kW ? 2 : 1); }

// compile-time loop
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// K consecutive words (K = 1, 2, 4) from an LDS address aligned to K words
template <int K>
__device__ __forceinline__ void elem_load(const float *src, float *w)
{
    if constexpr (K == 4) {
        const v4f t = *reinterpret_cast<const v4f *>(src);
        w[0] = t.x, w[1] = t.y, w[2] = t.z, w[3] = t.w;
    } else if constexpr (K == 2) {
        const v2f t = *reinterpret_cast<const v2f *>(src);
        w[0] = t.x, w[1] = t.y;
    } else {
        w[0] = src[0];
    }
}
__device__ __forceinline__ void elem_store2(float *dst, float e0, float e1) { *reinterpret_cast<v2f *>(dst) = v2f{e0, e1}; }

template <int TS, int S, bool DEC>
struct Casc {
    static constexpr int stages = S;
    static constexpr int rate = 1 << S;
    static constexpr int M(int s) { return kHbfM[TS][hbf_tuple_index(DEC, S, s)]; }
    static constexpr float tap(int s, int k) { return kHbfTaps[TS][hbf_tuple_index(DEC, S, s)][k]; }
    // full-chunk element count of stage s: decimator outputs / interpolator inputs
    static constexpr int n(int s) { return DEC ? (kCH >> (s + 1)) : ((kCH >> S) << s); }
    // LDS words: stream A = decimator even stream / interpolator x stream, B = decimator odd stream
    static constexpr int sizeA(int s)
    {
        return DEC ? up4(up4(M(s) - 1) + n(s) + kSlack) : up4(up4(2 * M(s) - 1) + n(s) + kSlack);
    }
    static constexpr int sizeB(int s) { return DEC ? up4(up4(2 * M(s) - 1) + n(s) + kSlack) : 0; }
    static constexpr int offA(int s)
    {
        int o = 0;
        for (int t = 0; t < s; t++) o += sizeA(t) + sizeB(t);
        return o;
    }
    static constexpr int offB(int s) { return offA(s) + sizeA(s); }
    static constexpr int lds_words = offA(S);
    static constexpr int state_off(int s)
    {
        int o = 0;
        for (int t = 0; t < s; t++) o += DEC ? 3 * M(t) - 2 : 2 * M(t) - 1;
        return o;
    }
};

// History roll of a FULL chunk as one flat copy.  After a chunk, every stream's last H samples become the history in
// front of the next chunk (src/hbf.rs:182-183,224 `copy_within`).  Done stage by stage with `lid < H` predicates that
// was ~120 wave instructions per 1024-sample chunk (8 predicated reads, 8 predicated writes, their exec-mask
// bookkeeping) out of ~570 — on a kernel that is instruction-issue bound (profiles/NOTES.md section 3).  The histories of all
// streams are one list of `roll_total` words (118 for /16); thread t moves words t, t + 64, ... and knows their
// source / destination LDS offsets from before the chunk loop: 2 reads, 2 writes, one predicate.
template <class C>
struct Roll {
    static constexpr int total()
    {
        int t = 0;
        for (int s = 0; s < C::stages; s++) t += C::sizeB(0) ? 3 * C::M(s) - 2 : 2 * C::M(s) - 1;
        return t;
    }
    static constexpr int per_thread = (total() + kW - 1) / kW;
    int src[per_thread], dst[per_thread];

    __device__ __forceinline__ void plan(int lid)
    {
        constexpr bool DEC = C::sizeB(0) != 0;
#pragma unroll
        for (int k = 0; k < per_thread; k++) {
            const int j = lid + k * kW;
            int base = 0, d = 0, sft = 0;
            static_for<0, C::stages>([&](auto s) {
                constexpr int s_ = decltype(s)::value;
                constexpr int M = C::M(s_), He = DEC ? M - 1 : 2 * M - 1, Ho = DEC ? 2 * M - 1 : 0;
                if (j >= base && j < base + He) d = C::offA(s_) + pad4(He) + (j - base), sft = C::n(s_);
                base += He;
                if (Ho && j >= base && j < base + Ho) d = C::offB(s_) + pad4(Ho) + (j - base), sft = C::n(s_);
                base += Ho;
            });
            dst[k] = d;
            src[k] = d + sft;
        }
    }
    __device__ __forceinline__ void run(float *lds, int lid) const
    {
        float t[per_thread];
#pragma unroll
        for (int k = 0; k < per_thread; k++)
            if ((k + 1) * kW <= total() || lid + k * kW < total()) t[k] = lds[src[k]];
        lds_wave_sync();
#pragma unroll
        for (int k = 0; k < per_thread; k++)
            if ((k + 1) * kW <= total() || lid + k * kW < total()) lds[dst[k]] = t[k];
        lds_wave_sync();
    }
};

// Σ_k (w[lo + 2M-1-k] + w[lo + k]) * tap_k, sequential from -0.0 (src/hbf.rs:60-66)
template <class C, int s>
__device__ __forceinline__ float window_sum(const float *w, int lo)
{
    constexpr int M = C::M(s);
    float acc = -0.0f;
    static_for<0, M>([&](auto k) {
        constexpr int K = decltype(k)::value;
        acc = acc + (w[lo + 2 * M - 1 - K] + w[lo + K]) * C::tap(s, K);
    });
    return acc;
}

// load NV vectors of P elements from an aligned LDS address into w[]
template <int P, int NV>
__device__ __forceinline__ void lds_window(const float *src, float *w)
{
#pragma unroll
    for (int v = 0; v < NV; v++) elem_load<P>(src + P * v, w + P * v);
}

// where the last stage's outputs go: element i of the chunk
struct StridedOut {
    float *p;
    size_t stride;
    __device__ __forceinline__ void operator()(int i, float v) const { p[size_t(i) * stride] = v; }
};
// ------------------------------------------------------------- decimator
// stage s of `HbfDec` (src/hbf.rs:163-185): y[i] = get(odd)[i] + even[i]
template <class C, int s, bool FULL, class YS>
__device__ __forceinline__ void dec_stage(float *lds, int n_rt, const YS &yout, int lid)
{
    constexpr int M = C::M(s), N = C::n(s), P = blocking(N);
    constexpr int de = pad4(M - 1), dq = pad4(2 * M - 1);
    constexpr int oq = dq % P, NV = (oq + 2 * M + P - 1 + P - 1) / P;
    constexpr int oe = de % P, NVE = (oe + P + P - 1) / P;
    constexpr int ITER = (N / P + kW - 1) / kW;
    const float *Ev = lds + C::offA(s) + (de - oe);
    const float *O = lds + C::offB(s) + (dq - oq);
    const int n = FULL ? N : n_rt;
#pragma unroll
    for (int it = 0; it < ITER; it++) {
        const int g = lid + it * kW, i0 = g * P;
        if (i0 >= n) continue;
        float w[NV * P], e[NVE * P], out[P];
        lds_window<P, NV>(O + i0, w);
        lds_window<P, NVE>(Ev + i0, e);
#pragma unroll
        for (int p = 0; p < P; p++) out[p] = window_sum<C, s>(w, oq + p) + e[oe + p];
        if constexpr (s + 1 == C::stages) {
#pragma unroll
            for (int p = 0; p < P; p++)
                if (FULL || i0 + p < n) yout(i0 + p, out[p]);
        } else {
            // `ChunkIn<_, 2>`: consecutive outputs pair up as the next stage's [even, odd]
            constexpr int Mn = C::M(s + 1);
            float *En = lds + C::offA(s + 1) + up4(Mn - 1), *On = lds + C::offB(s + 1) + up4(2 * Mn - 1);
            if constexpr (P == 4) {
                elem_store2(En + (i0 >> 1), out[0], out[2]);
                elem_store2(On + (i0 >> 1), out[1], out[3]);
            } else if constexpr (P == 2) {
                En[g] = out[0];
                On[g] = out[1];
            } else {
                (g & 1 ? On : En)[g >> 1] = out[0];
            }
        }
    }
}

template <class C, bool FULL, class YS>
__device__ __forceinline__ void dec_chunk(float *lds, int nin, const YS &yout, int lid, const Roll<C> &roll)
{
    static_for<0, C::stages>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        dec_stage<C, s_, FULL>(lds, nin >> (s_ + 1), yout, lid);
        lds_wave_sync();
    });
    if constexpr (FULL) {
        roll.run(lds, lid);
        return;
    }
    // ragged last chunk: roll the histories stage by stage, word j <- word n_s + j (src/hbf.rs:182-183)
    float ke[C::stages], ko[C::stages];
    static_for<0, C::stages>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        constexpr int M = C::M(s_), He = M - 1, Ho = 2 * M - 1;
        const int ns = nin >> (s_ + 1);
        ke[s_] = lid < He ? lds[C::offA(s_) + pad4(He) + ns + lid] : 0.f;
        ko[s_] = lid < Ho ? lds[C::offB(s_) + pad4(Ho) + ns + lid] : 0.f;
    });
    lds_wave_sync();
    static_for<0, C::stages>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        constexpr int M = C::M(s_), He = M - 1, Ho = 2 * M - 1;
        if (lid < He) lds[C::offA(s_) + pad4(He) + lid] = ke[s_];
        if (lid < Ho) lds[C::offB(s_) + pad4(Ho) + lid] = ko[s_];
    });
    lds_wave_sync();
}

// LM: x[(lane*frames + f)*R + k], y[lane*frames + f];  FM: x[(f*lanes + lane)*R + k], y[f*lanes + lane]
// FRAME_MAJOR: lanes l and l+1 share 128-byte lines (a lane owns R*4 <= 64 bytes per frame).  Workgroups
// are dealt to the 8 XCDs round-robin, and every XCD has its own L2, so with lane = blockIdx the two
// halves of a line are fetched by two different L2s (PMC: 2.0x FETCH_SIZE, 10x WRITE_SIZE at C3).
// Give each XCD one contiguous eighth of the lanes instead: neighbours then meet in the same L2, a few
// dispatch rounds apart.  grid = 8 * ceil(lanes / 8); workgroups past the end exit.
__device__ __forceinline__ size_t xcd_lane(size_t lanes)
{
    const size_t per = (lanes + 7) / 8;
    return (blockIdx.x % 8) * per + blockIdx.x / 8;
}

// FRAME_MAJOR decimator, one wave per lane: the fallback for lane counts that are not whole workgroups of the kernels below
// (and of hbf_ring.h).  x[(f*lanes + lane)*R + k], y[f*lanes + lane]; its loads are R*4-byte fragments that rely on L2
// to merge neighbouring lanes, hence one chunk in flight only (two ran 1.72 instead of 1.4 ms at C3).
template <class C>
__global__ __launch_bounds__(kW) void hbf_dec_wave(uint32_t *st, const float *x, float *y, const size_t lanes, const size_t frames)
{
    __shared__ __attribute__((aligned(16))) float lds[C::lds_words];
    constexpr int S = C::stages, R = C::rate;
    static_assert(R >= 4, "FRAME_MAJOR wave kernel needs 16-byte frame pieces");
    const int lid = threadIdx.x;
    const size_t lane = xcd_lane(lanes);
    if (lane >= lanes) return;

    // history <- state words (per stage: even[M-1] then odd[2M-1], oldest first)
    static_for<0, S>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        constexpr int M = C::M(s_), He = M - 1, Ho = 2 * M - 1, so = C::state_off(s_);
        if (lid < He) lds[C::offA(s_) + pad4(He) + lid] = __uint_as_float(st[size_t(so + lid) * lanes + lane]);
        if (lid < Ho) lds[C::offB(s_) + pad4(Ho) + lid] = __uint_as_float(st[size_t(so + He + lid) * lanes + lane]);
    });

    Roll<C> roll;
    roll.plan(lid);
    constexpr int CHF = kCH / R;        // output frames per chunk
    constexpr int kPre = kCH / 4 / kW;  // 16-byte pieces per thread and chunk
    constexpr int M0 = C::M(0), PPF = R / 4;
    float *E0n = lds + C::offA(0) + up4(M0 - 1), *O0n = lds + C::offB(0) + up4(2 * M0 - 1);
    v4f pre[kPre];
    auto piece = [&](size_t f0, int q) -> const v4f * {
        return reinterpret_cast<const v4f *>(x + ((f0 + size_t(q / PPF)) * lanes + lane) * size_t(R)) + (q % PPF);
    };
    auto fetch = [&](size_t f0) {
        if (f0 >= frames) return;
        const int nq = frames - f0 >= size_t(CHF) ? kCH / 4 : int(frames - f0) * R / 4;
#pragma unroll
        for (int i = 0; i < kPre; i++)
            if (lid + i * kW < nq) pre[i] = *piece(f0, lid + i * kW);
    };
    fetch(0);
    for (size_t f0 = 0; f0 < frames; f0 += size_t(CHF)) {
        const int nin = frames - f0 >= size_t(CHF) ? kCH : int(frames - f0) * R;
        // stage-0 input: pairs [even, odd] split into the two streams
#pragma unroll
        for (int i = 0; i < kPre; i++) {
            const int q = lid + i * kW;
            if (q < nin / 4) {
                elem_store2(E0n + 2 * q, pre[i].x, pre[i].z);
                elem_store2(O0n + 2 * q, pre[i].y, pre[i].w);
            }
        }
        fetch(f0 + size_t(CHF));
        lds_wave_sync();
        const StridedOut yout{y + f0 * lanes + lane, lanes};
        if (nin == kCH)
            dec_chunk<C, true>(lds, nin, yout, lid, roll);
        else
            dec_chunk<C, false>(lds, nin, yout, lid, roll);
    }

    static_for<0, S>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        constexpr int M = C::M(s_), He = M - 1, Ho = 2 * M - 1, so = C::state_off(s_);
        if (lid < He) st[size_t(so + lid) * lanes + lane] = __float_as_uint(lds[C::offA(s_) + pad4(He) + lid]);
        if (lid < Ho) st[size_t(so + He + lid) * lanes + lane] = __float_as_uint(lds[C::offB(s_) + pad4(Ho) + lid]);
    });
}

// FRAME_MAJOR decimator, kBlkLanes lanes per workgroup (one wave each).  The wave-per-lane kernel reads 64-byte
// pieces and writes single floats y[f*lanes + lane]; the 32 lanes of an output line are different
// workgroups, and L2 (refilled every ~10 us by the input stream) evicts the line between their writes
// (PMC: WRITE_SIZE 3.4x the output bytes).  Here the waves of a workgroup load whole contiguous runs
// (kBlkLanes * R * 4 bytes per frame) cooperatively into each other's stage-0 streams and stage a chunk's
// outputs in an LDS tile [frame][lane] that is written as kBlkLanes * 4-byte pieces: FETCH 1.00x, WRITE 1.06x
// at 16 lanes.  Two workgroup barriers per chunk; with 16 lanes (one workgroup per CU in lockstep) the
// barriers cost what the traffic saved (1.45 ms), 4 lanes per workgroup (4 independent workgroups per CU)
// measured best: 1.23-1.29 ms vs 1.37-1.46 ms for the wave-per-lane kernel at C3.
// Round 3 (tools/exp_hbf.sh with VARIANTS=BLK8/BLK2/..., profiles/r03_exp_hbf_blk.jsonl): 8 lanes per workgroup 1.85 ms, 2 lanes
// 1.11, 4 lanes 1.08 on the same box; padding the per-lane LDS regions so that the cooperative stage-0 writes of the four
// lanes fall 16 banks apart (they are 4 apart: 2116 words per region) changed nothing (1.0818 ms both ways) — the
// bank-conflict cycles of profiles/r03_c3_pmc.csv are not what bounds this kernel.
#ifndef IDSP_HBF_BLK_LANES
#define IDSP_HBF_BLK_LANES 4
#endif
constexpr int kBlkLanes = IDSP_HBF_BLK_LANES;

template <class C>
__global__ __launch_bounds__(kBlkLanes *kW) void hbf_dec_block_fm(uint32_t *st, const float *x, float *y, const size_t lanes,
                                                                   const size_t frames)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = C::stages, R = C::rate;
    constexpr int CHF = kCH / R;  // output frames per chunk
    const int lid = threadIdx.x % kW, wave = threadIdx.x / kW;
    float *lds = smem + wave * up4(C::lds_words);
    float *otile = smem + kBlkLanes * up4(C::lds_words);  // [2][CHF][kBlkLanes]
    const size_t ngroups = lanes / kBlkLanes, per = (ngroups + 7) / 8;
    const size_t group = (blockIdx.x % 8) * per + blockIdx.x / 8;  // XCD-aware, see xcd_lane()
    if (group >= ngroups) return;
    const size_t lane = group * kBlkLanes + wave;

    static_for<0, S>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        constexpr int M = C::M(s_), He = M - 1, Ho = 2 * M - 1, so = C::state_off(s_);
        if (lid < He) lds[C::offA(s_) + pad4(He) + lid] = __uint_as_float(st[size_t(so + lid) * lanes + lane]);
        if (lid < Ho) lds[C::offB(s_) + pad4(Ho) + lid] = __uint_as_float(st[size_t(so + He + lid) * lanes + lane]);
    });

    Roll<C> roll;
    roll.plan(lid);
    // Cooperative input: a frame of the workgroup's 16 lanes is 16 * R * 4 contiguous bytes; thread t of the
    // workgroup moves 16-byte vector v = t % VPF of frame t / VPF (+ FPI per instruction), so every wave
    // instruction reads whole contiguous runs, and drops it into the owning lane's stage-0 streams.
    constexpr int M0 = C::M(0), PPF = R / 4;          // 16-byte pieces per lane and frame
    constexpr int VPF = kBlkLanes * PPF;              // vectors per frame of the workgroup
    constexpr int NT = kBlkLanes * kW;                // threads
    static_assert(NT % VPF == 0 && (CHF * VPF) % NT == 0, "chunk is a whole number of workgroup loads");
    constexpr int FPI = NT / VPF;                     // frames per load instruction
    constexpr int kPre = CHF * VPF / NT;              // loads per thread and chunk
    const int tv = threadIdx.x % VPF, tf = threadIdx.x / VPF;
    const int olane = tv / PPF, opiece = tv % PPF;    // owner lane (wave) and piece inside its frame
    float *olds = smem + olane * up4(C::lds_words);
    float *E0o = olds + C::offA(0) + up4(M0 - 1), *O0o = olds + C::offB(0) + up4(2 * M0 - 1);
    const v4f *xg = reinterpret_cast<const v4f *>(x + group * kBlkLanes * size_t(R)) + tv;
    const size_t fpitch = lanes * size_t(R) / 4;      // vectors per frame row
    // two chunks in flight (register ring, statically indexed: the chunk loop is unrolled by two)
    v4f pre[2][kPre];
    auto fetch = [&](int slot, size_t f0) {
        if (f0 >= frames) return;
        const int nf = frames - f0 < size_t(CHF) ? int(frames - f0) : CHF;
#pragma unroll
        for (int i = 0; i < kPre; i++) {
            const int fr = tf + i * FPI;
            if (fr < nf) pre[slot][i] = xg[(f0 + size_t(fr)) * fpitch];
        }
    };
    auto chunk = [&](int slot, size_t f0) {
        const int nf = frames - f0 < size_t(CHF) ? int(frames - f0) : CHF;
        const int nin = nf * R;
        // stage-0 input of the owner lane: pairs [even, odd] split into the two streams (piece q of its chunk)
#pragma unroll
        for (int i = 0; i < kPre; i++) {
            const int fr = tf + i * FPI;
            if (fr < nf) {
                const int q = fr * PPF + opiece;
                *reinterpret_cast<v2f *>(E0o + 2 * q) = v2f{pre[slot][i].x, pre[slot][i].z};
                *reinterpret_cast<v2f *>(O0o + 2 * q) = v2f{pre[slot][i].y, pre[slot][i].w};
            }
        }
        fetch(slot, f0 + 2 * size_t(CHF));
        lds_barrier();  // every lane's chunk is in place
        float *ot = otile + slot * CHF * kBlkLanes;
        if (nf == CHF)
            dec_chunk<C, true>(lds, nin, StridedOut{ot + wave, size_t(kBlkLanes)}, lid, roll);
        else
            dec_chunk<C, false>(lds, nin, StridedOut{ot + wave, size_t(kBlkLanes)}, lid, roll);
        lds_barrier();  // outputs staged, and nobody still reads the stage-0 streams
        // 16 threads write one frame's 64 bytes; the other tile is free until the next chunk's barrier
        for (int t = threadIdx.x; t < nf * kBlkLanes; t += NT)
            y[(f0 + size_t(t / kBlkLanes)) * lanes + group * kBlkLanes + (t % kBlkLanes)] = ot[t];
    };
    fetch(0, 0);
    fetch(1, size_t(CHF));
    for (size_t f0 = 0; f0 < frames; f0 += 2 * size_t(CHF)) {
        chunk(0, f0);
        if (f0 + CHF < frames) chunk(1, f0 + CHF);
    }

    static_for<0, S>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        constexpr int M = C::M(s_), He = M - 1, Ho = 2 * M - 1, so = C::state_off(s_);
        if (lid < He) st[size_t(so + lid) * lanes + lane] = __float_as_uint(lds[C::offA(s_) + pad4(He) + lid]);
        if (lid < Ho) st[size_t(so + He + lid) * lanes + lane] = __float_as_uint(lds[C::offB(s_) + pad4(Ho) + lid]);
    });
}

// ------------------------------------------------------------ interpolator
// stage s of `HbfInt` (src/hbf.rs:207-227): pair i = [get(x)[i], x[M + i]]
template <class C, int s, bool FULL, bool LM>
__device__ __forceinline__ void int_stage(float *lds, int n_rt, float *y, size_t lanes, size_t frames, size_t lane,
                                          size_t f0, int lid)
{
    // the last stage stores to global memory: with P = 2 a thread owns 4 consecutive outputs = one 16-byte
    // store, so a wave instruction writes 1 KiB of whole lines (P = 4 puts 32 bytes per thread behind two
    // instructions that each cover half of every line)
    constexpr int M = C::M(s), N = C::n(s), P = (s + 1 == C::stages && blocking(N) > 2) ? 2 : blocking(N), R = C::rate;
    constexpr int dx = pad4(2 * M - 1), ox = dx % P, NV = (ox + 2 * M + P - 1 + P - 1) / P;
    constexpr int ITER = (N / P + kW - 1) / kW;
    const float *X = lds + C::offA(s) + (dx - ox);
    const int n = FULL ? N : n_rt;
#pragma unroll
    for (int it = 0; it < ITER; it++) {
        const int g = lid + it * kW, i0 = g * P;
        if (i0 >= n) continue;
        float w[NV * P], out[2 * P];
        lds_window<P, NV>(X + i0, w);
#pragma unroll
        for (int p = 0; p < P; p++) {
            out[2 * p] = window_sum<C, s>(w, ox + p);  // interpolated
            out[2 * p + 1] = w[ox + M + p];            // centre tap: identity
        }
        if constexpr (s + 1 == C::stages) {
            // 2P consecutive outputs of this lane starting at chunk-local sample 2*i0
            const int o0 = 2 * i0;
            if constexpr (P >= 2) {
#pragma unroll
                for (int h = 0; h < 2 * P; h += 4) {
                    const int o = o0 + h;  // 4 consecutive samples inside one frame (R >= 4)
                    float *dst = LM ? y + (lane * frames + f0) * size_t(R) + o
                                    : y + ((f0 + size_t(o / R)) * lanes + lane) * size_t(R) + (o % R);
                    if (FULL || o + 3 < 2 * n)
                        *reinterpret_cast<v4f *>(dst) = v4f{out[h], out[h + 1], out[h + 2], out[h + 3]};
                    else
                        for (int j = 0; j < 4; j++)
                            if (o + j < 2 * n) dst[j] = out[h + j];
                }
            } else {
                float *dst = LM ? y + (lane * frames + f0) * size_t(R) + o0
                                : y + ((f0 + size_t(o0 / R)) * lanes + lane) * size_t(R) + (o0 % R);
                *reinterpret_cast<v2f *>(dst) = v2f{out[0], out[1]};
            }
        } else {
            // `ChunkOut<_, 2>`: the pairs flatten into the next stage's input stream
            float *Xn = lds + C::offA(s + 1) + up4(2 * C::M(s + 1) - 1) + 2 * i0;
            if constexpr (P == 4) {
                *reinterpret_cast<v4f *>(Xn) = v4f{out[0], out[1], out[2], out[3]};
                *reinterpret_cast<v4f *>(Xn + 4) = v4f{out[4], out[5], out[6], out[7]};
            } else if constexpr (P == 2) {
                *reinterpret_cast<v4f *>(Xn) = v4f{out[0], out[1], out[2], out[3]};
            } else {
                *reinterpret_cast<v2f *>(Xn) = v2f{out[0], out[1]};
            }
        }
    }
}

template <class C, bool FULL, bool LM>
__device__ __forceinline__ void int_chunk(float *lds, int nf, float *y, size_t lanes, size_t frames, size_t lane,
                                          size_t f0, int lid, const Roll<C> &roll)
{
    static_for<0, C::stages>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        int_stage<C, s_, FULL, LM>(lds, nf << s_, y, lanes, frames, lane, f0, lid);
        lds_wave_sync();
    });
    if constexpr (FULL) {
        roll.run(lds, lid);
        return;
    }
    float kx[C::stages];
    static_for<0, C::stages>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        constexpr int H = 2 * C::M(s_) - 1;
        kx[s_] = lid < H ? lds[C::offA(s_) + pad4(H) + (nf << s_) + lid] : 0.f;
    });
    lds_wave_sync();
    static_for<0, C::stages>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        constexpr int H = 2 * C::M(s_) - 1;
        if (lid < H) lds[C::offA(s_) + pad4(H) + lid] = kx[s_];
    });
    lds_wave_sync();
}

// LM: x[lane*frames + f], y[(lane*frames + f)*R + k];  FM: x[f*lanes + lane], y[(f*lanes + lane)*R + k]
template <class C, bool LM>
__global__ __launch_bounds__(kW) void hbf_int_wave(uint32_t *st, const float *x, float *y, const size_t lanes,
                                                   const size_t frames)
{
    __shared__ __attribute__((aligned(16))) float lds[C::lds_words];
    constexpr int S = C::stages, R = C::rate;
    static_assert(R >= 4 || LM, "FRAME_MAJOR wave kernel needs 16-byte frame pieces");
    const int lid = threadIdx.x;
    const size_t lane = LM ? size_t(blockIdx.x) : xcd_lane(lanes);
    if (lane >= lanes) return;

    static_for<0, S>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        constexpr int H = 2 * C::M(s_) - 1;
        if (lid < H) lds[C::offA(s_) + pad4(H) + lid] = __uint_as_float(st[size_t(C::state_off(s_) + lid) * lanes + lane]);
    });

    Roll<C> roll;
    roll.plan(lid);
    constexpr int CHF = kCH / R;  // input frames per chunk (<= 512)
    constexpr int kPre = (CHF + kW - 1) / kW;
    float *X0n = lds + C::offA(0) + up4(2 * C::M(0) - 1);
    float pre[kPre];
    auto fetch = [&](size_t f0) {
        const int nf = frames - f0 < size_t(CHF) ? int(frames - f0) : CHF;
#pragma unroll
        for (int i = 0; i < kPre; i++) {
            const int j = lid + i * kW;
            // LANE_MAJOR: contiguous, streamed once -> nontemporal.  FRAME_MAJOR: a line holds one frame of 32 lanes,
            // i.e. of 32 workgroups; it must stay cached for them (nontemporal re-fetched it for every lane)
            if (j < nf) pre[i] = LM ? __builtin_nontemporal_load(x + lane * frames + f0 + size_t(j)) : x[(f0 + size_t(j)) * lanes + lane];
        }
    };
    fetch(0);
    for (size_t f0 = 0; f0 < frames; f0 += CHF) {
        const int nf = frames - f0 < size_t(CHF) ? int(frames - f0) : CHF;
#pragma unroll
        for (int i = 0; i < kPre; i++) {
            const int j = lid + i * kW;
            if (j < nf) X0n[j] = pre[i];
        }
        if (f0 + CHF < frames) fetch(f0 + CHF);
        lds_wave_sync();
        if (nf == CHF)
            int_chunk<C, true, LM>(lds, nf, y, lanes, frames, lane, f0, lid, roll);
        else
            int_chunk<C, false, LM>(lds, nf, y, lanes, frames, lane, f0, lid, roll);
    }

    static_for<0, S>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        constexpr int H = 2 * C::M(s_) - 1;
        if (lid < H) st[size_t(C::state_off(s_) + lid) * lanes + lane] = __float_as_uint(lds[C::offA(s_) + pad4(H) + lid]);
    });
}

// FRAME_MAJOR interpolator, kBlkLanes lanes per workgroup: each wave interpolates its lane's chunk into an
// LDS staging row (as if LANE_MAJOR), then the workgroup writes the chunk frame by frame as contiguous
// kBlkLanes * R * 4-byte runs (whole lines) instead of 64-byte pieces at a lanes*R*4 pitch per wave.
template <class C>
__global__ __launch_bounds__(kBlkLanes *kW) void hbf_int_block_fm(uint32_t *st, const float *x, float *y, const size_t lanes,
                                                                   const size_t frames)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = C::stages, R = C::rate;
    constexpr int CHF = kCH / R;  // input frames per chunk
    constexpr int NT = kBlkLanes * kW;
    const int lid = threadIdx.x % kW, wave = threadIdx.x / kW;
    float *lds = smem + wave * up4(C::lds_words);
    float *stage = smem + kBlkLanes * up4(C::lds_words);  // [kBlkLanes][kCH]
    const size_t ngroups = lanes / kBlkLanes, per = (ngroups + 7) / 8;
    const size_t group = (blockIdx.x % 8) * per + blockIdx.x / 8;  // XCD-aware, see xcd_lane()
    if (group >= ngroups) return;
    const size_t lane = group * kBlkLanes + wave;

    static_for<0, S>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        constexpr int H = 2 * C::M(s_) - 1;
        if (lid < H) lds[C::offA(s_) + pad4(H) + lid] = __uint_as_float(st[size_t(C::state_off(s_) + lid) * lanes + lane]);
    });

    Roll<C> roll;
    roll.plan(lid);
    constexpr int kPre = (CHF + kW - 1) / kW;
    float *X0n = lds + C::offA(0) + up4(2 * C::M(0) - 1);
    float pre[kPre];
    auto fetch = [&](size_t f0) {
        const int nf = frames - f0 < size_t(CHF) ? int(frames - f0) : CHF;
#pragma unroll
        for (int i = 0; i < kPre; i++) {
            const int j = lid + i * kW;
            if (j < nf) pre[i] = x[(f0 + size_t(j)) * lanes + lane];
        }
    };
    // write-out: vector v of the workgroup's frame = 16-byte piece v % (R/4) of lane v / (R/4)
    constexpr int PPF = R / 4, VPF = kBlkLanes * PPF;
    static_assert(NT % VPF == 0, "threads cover whole frames");
    constexpr int FPI = NT / VPF;
    const int tv = threadIdx.x % VPF, tf = threadIdx.x / VPF;
    const float *src = stage + (tv / PPF) * kCH + (tv % PPF) * 4;
    v4f *dst = reinterpret_cast<v4f *>(y + group * kBlkLanes * size_t(R)) + tv;
    const size_t fpitch = lanes * size_t(R) / 4;

    fetch(0);
    for (size_t f0 = 0; f0 < frames; f0 += CHF) {
        const int nf = frames - f0 < size_t(CHF) ? int(frames - f0) : CHF;
#pragma unroll
        for (int i = 0; i < kPre; i++) {
            const int j = lid + i * kW;
            if (j < nf) X0n[j] = pre[i];
        }
        if (f0 + CHF < frames) fetch(f0 + CHF);
        lds_wave_sync();
        // LANE_MAJOR addressing onto the staging row: y' + (0 * frames + 0) * R + o
        if (nf == CHF)
            int_chunk<C, true, true>(lds, nf, stage + wave * kCH, 1, 0, 0, 0, lid, roll);
        else
            int_chunk<C, false, true>(lds, nf, stage + wave * kCH, 1, 0, 0, 0, lid, roll);
        lds_barrier();  // every lane's chunk is staged
        for (int fr = tf; fr < nf; fr += FPI) dst[(f0 + size_t(fr)) * fpitch] = *reinterpret_cast<const v4f *>(src + fr * R);
        lds_barrier();  // staging rows free again
    }

    static_for<0, S>([&](auto s) {
        constexpr int s_ = decltype(s)::value;
        constexpr int H = 2 * C::M(s_) - 1;
        if (lid < H) st[size_t(C::state_off(s_) + lid) * lanes + lane] = __float_as_uint(lds[C::offA(s_) + pad4(H) + lid]);
    });
}

// -------------------------------------------------------------------- host
template <int TS, int S, bool DEC>
int launch_wave(uint32_t *st, const float *x, float *y, size_t lanes, size_t frames, bool lm, hipStream_t stream)
{
    using C = Casc<TS, S, DEC>;
    const dim3 grid{unsigned(lm ? lanes : 8 * ((lanes + 7) / 8))}, block{unsigned(kW)};
    if (lm) {
        if constexpr (DEC) {
            return 1;  // LANE_MAJOR decimators: hbf_ring.h
        } else {
            note_kernel("hbf_int_wave[LaneMajor]", typeid(C).name());
            hipLaunchKernelGGL((hbf_int_wave<C, true>), grid, block, 0, stream, st, x, y, lanes, frames);
        }
    } else {
        if constexpr (S >= 2) {
            if constexpr (DEC) {
                constexpr size_t bytes = (size_t(kBlkLanes) * up4(C::lds_words) + 2 * (kCH / C::rate) * kBlkLanes) * sizeof(float);
                static const bool use_block = !diag_env("IDSP_HBF_NO_BLOCK_FM");
                if (use_block && bytes <= 160 * 1024 && lanes % kBlkLanes == 0) {
                    if (ensure_dyn_lds<&hbf_dec_block_fm<C>>(bytes)) return 1;  // once per device (common.h)
                    const size_t ngroups = lanes / kBlkLanes;
                    note_kernel("hbf_dec_block_fm", typeid(C).name());
                    hipLaunchKernelGGL((hbf_dec_block_fm<C>), dim3(unsigned(8 * ((ngroups + 7) / 8))), dim3(kBlkLanes * kW), bytes, stream,
                                       st, x, y, lanes, frames);
                    return 0;
                }
                note_kernel("hbf_dec_wave[FrameMajor]", typeid(C).name());
                hipLaunchKernelGGL((hbf_dec_wave<C>), grid, block, 0, stream, st, x, y, lanes, frames);
            }
            else {
                constexpr size_t bytes = (size_t(kBlkLanes) * up4(C::lds_words) + size_t(kBlkLanes) * kCH) * sizeof(float);
                static const bool use_block = !diag_env("IDSP_HBF_NO_BLOCK_FM");
                if (use_block && bytes <= 160 * 1024 && lanes % kBlkLanes == 0) {
                    if (ensure_dyn_lds<&hbf_int_block_fm<C>>(bytes)) return 1;  // once per device (common.h)
                    const size_t ngroups = lanes / kBlkLanes;
                    note_kernel("hbf_int_block_fm", typeid(C).name());
                    hipLaunchKernelGGL((hbf_int_block_fm<C>), dim3(unsigned(8 * ((ngroups + 7) / 8))), dim3(kBlkLanes * kW), bytes, stream,
                                       st, x, y, lanes, frames);
                    return 0;
                }
                note_kernel("hbf_int_wave[FrameMajor]", typeid(C).name());
                hipLaunchKernelGGL((hbf_int_wave<C, false>), grid, block, 0, stream, st, x, y, lanes, frames);
            }
        } else {
            return 1;  // not handled here
        }
    }
    return 0;
}

template <int TS, bool DEC>
int launch_wave_s(int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames, bool lm,
                  hipStream_t stream)
{
    switch (stages) {
        case 1: return launch_wave<TS, 1, DEC>(st, x, y, lanes, frames, lm, stream);
        case 2: return launch_wave<TS, 2, DEC>(st, x, y, lanes, frames, lm, stream);
        case 3: return launch_wave<TS, 3, DEC>(st, x, y, lanes, frames, lm, stream);
        case 4: return launch_wave<TS, 4, DEC>(st, x, y, lanes, frames, lm, stream);
        case 5: return launch_wave<TS, 5, DEC>(st, x, y, lanes, frames, lm, stream);
        default: return 1;
    }
}

}  // namespace hbfw

// Returns 0 when a specialised wave kernel was launched, 1 when the request is
// not covered (caller falls back to the generic kernels of hbf.hip).
int hbf_wave_dec(int tap_set, int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames,
                 bool lane_major, hipStream_t stream);
int hbf_wave_int(int tap_set, int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames,
                 bool lane_major, hipStream_t stream);

}  // namespace idsp
