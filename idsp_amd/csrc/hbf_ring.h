// hbf_ring.h — half-band decimator cascades (HBF_DEC_CASCADE over HBF_TAPS / HBF_TAPS_98, src/hbf.rs:142-192,385-421)
// on an LDS-DMA input ring with packed-f32 arithmetic.  Round 4 replacement of hbf_wave.h's decimator kernels for the
// shapes it covers (they stay as the fallback, and the interpolators stay there).  Round 6: the LANE_MAJOR kernel of this
// file gave way to the register-blocked one of hbf_blk.h (which reuses the stage-0 body from here); the FRAME_MAJOR /16
// kernel stays (the blocked FrameMajor form measured equal at best, hbf_blk.h).
//
// Why: the round-3 kernel was co-limited three ways at C3 (/16, 16384 lanes x 65536 samples) — per 1024 input samples a
// wave issued ~285 essential f32 VALU operations at the ~4.2 cycles this chip takes per non-packed f32 operation and
// SIMD (0.6 ms of VALU), ~290 LDS cycles on a CU whose 16 waves share one LDS (73 % busy: the P = 1 last stage read its
// 46-word window with 46 `ds_read_b32`), and its input went through 32 VGPRs of double buffer.  Here:
//
//  * input: `global_load_lds_dwordx4` into a ring of four 1 KiB slots per lane (one request per wave and slot, three
//    to four in flight), no VGPR staging.  FRAME_MAJOR (/16): 16 lanes per workgroup; wave w's request moves frame 16 q + w of ALL 16 lanes = 1 KiB of contiguous global memory
//    into row 16 q + w of a [64 frames][16 lanes x 64 B + 64 B pad] tile; one LDS barrier per slot hands the rows over.
//  * stage 0 reads the raw interleaved [even, odd] samples straight from the ring (no split into two streams): thread
//    t of slot q owns 16-byte piece t (2 outputs) and reads pieces t - M .. t; consecutive threads read consecutive
//    pieces (conflict-free `ds_read_b128`), the pieces before the slot are the tail of the previous slot (the ring is
//    the reference's `odd` / `even` history, src/hbf.rs:166-183).
//  * arithmetic: the symmetric sums `new + old` (src/hbf.rs:60-63) stay scalar `v_add_f32` (their operands sit an odd
//    distance apart, so no two of them form aligned register pairs), their results are placed as pairs of neighbouring
//    outputs, and the tap multiplies and the sequential accumulation run as `v_pk_mul_f32` / `v_pk_add_f32` on those
//    pairs: 2M instead of 3M VALU operations per output, every IEEE operation and its order per output unchanged
//    (nothing fused, nothing re-associated: -ffp-contract=off).  The lowest-rate stages have one output per thread;
//    there the pairs run along the taps (k, k + 1) with the reversed `new` pair selected by `op_sel`, and the stage's odd
//    stream is kept twice, one word apart, so that odd and even threads both read aligned 8-byte pairs.
//  * one round = 1024 input samples per lane = 4 slots; stages >= 1 run once per round with 4 / 2 / 1 outputs per thread.
//
// Rounds past the end of the data, and the pieces of the last round past it, are REQUESTED like any other (from a
// clamped, valid address) and computed on whatever they hold; only the stores, the history roll and the state
// write-back know the true count.  That keeps the request / wait bookkeeping static.
#pragma once

#include <type_traits>
#include <typeinfo>
#include <utility>

#include "common.h"
#include "hbf_taps.h"
#include "lds_dma.h"

namespace idsp {
namespace hbfr {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
// One `ds_read_b64` that stays one: the volatile access is not merged with its neighbours into `ds_read2_b64` (half
// the LDS rate, 8-bit offsets).  Volatile accesses do not take part in address-space inference, hence the explicit
// LDS pointer (a generic one would make this a `flat_load`).
__device__ __forceinline__ v2f lds_read_b64(const float *p)
{
    return *(const volatile __attribute__((address_space(3))) v2f *)p;
}

constexpr int kW = 64;               // threads per wave
constexpr int kSC = 1024;            // raw samples per lane and round
constexpr int kSlots = 4;            // ring slots = requests per lane and round
constexpr int kSlotW = kSC / kSlots; // 256 words = 1 KiB = one request
constexpr int up4(int v) { return (v + 3) & ~3; }

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Cascade geometry.  Stage s halves the rate; stage 0 reads the ring, stages s >= 1 read their [even, odd] streams
// from the wave's LDS region: E_s (M - 1 words of history, then N(s) samples) and O_s (2M - 1 words of history, then
// N(s) samples), O_s twice when the stage runs one output per thread.
template <int TS, int S>
struct Lay {
    static constexpr int stages = S;
    static constexpr int rate = 1 << S;
    static constexpr int M(int s) { return kHbfM[TS][hbf_tuple_index(true, S, s)]; }
    static constexpr float tap(int s, int k) { return kHbfTaps[TS][hbf_tuple_index(true, S, s)][k]; }
    static constexpr int He(int s) { return M(s) - 1; }
    static constexpr int Ho(int s) { return 2 * M(s) - 1; }
    static constexpr int N(int s) { return kSC >> (s + 1); }  // outputs of stage s per round
    static constexpr int P(int s) { return N(s) >= 4 * kW ? 4 : (N(s) >= 2 * kW ? 2 : 1); }
    static constexpr bool dual(int s) { return s >= 1 && s < S && P(s) == 1; }
    static constexpr int sizeE(int s) { return up4(He(s) + N(s) + 4); }
    static constexpr int sizeO(int s) { return up4(Ho(s) + N(s) + 6); }
    static constexpr int offE(int s)
    {
        int o = 0;
        for (int t = 1; t < s; t++) o += sizeS(t);
        return o;
    }
    static constexpr int offO(int s) { return offE(s) + sizeE(s); }
    // Second copy: element j at offOB + Ho + j + 1.  In a 32-lane `ds_read_b64` group the 16 even threads read 32
    // consecutive words of the first copy from offO + t and the 16 odd threads 32 consecutive words of the second from
    // offOB + t + 1: the copies sit == 30 (mod 64) words apart, so that the two runs cover all 64 banks once.
    static constexpr int gapB(int s)
    {
        int d = sizeO(s);
        while (d % 64 != 30) d += 2;
        return d;
    }
    static constexpr int offOB(int s) { return offO(s) + gapB(s); }
    static constexpr int sizeS(int s) { return sizeE(s) + (dual(s) ? up4(gapB(s) + sizeO(s)) : sizeO(s)); }
    static constexpr int words = offE(S);
    static constexpr int state_off(int s)
    {
        int o = 0;
        for (int t = 0; t < s; t++) o += 3 * M(t) - 2;
        return o;
    }
};

// ------------------------------------------------------------------------------------------------ stage bodies
// raw sample n (relative to the thread's own piece, n in [-4M, 3]) out of the pieces t - M .. t
template <int M, int n>
__device__ __forceinline__ float raw_at(const v4f *pc)
{
    constexpr int m = n + 4 * M;
    static_assert(m >= 0 && m < 4 * (M + 1), "raw sample outside the loaded pieces");
    return pc[m >> 2][m & 3];
}

// Stage 0 on one thread's pieces: the two outputs 2t, 2t + 1 of its slot.  e_0[d] = raw[2d], o_0[d] = raw[2d + 1]
// relative to output 2t; y_i = sum_k (o[i - k] + o[i - (2M-1) + k]) * c_k + e[i - (M-1)]  (src/hbf.rs:46-68,163-185).
// The sum starts from -0.0 (f32::sum), and -0.0 + p == p for every p, so the first product seeds the accumulator.
template <class L>
__device__ __forceinline__ v2f stage0_pair(const v4f *pc)
{
    constexpr int M = L::M(0);
    v2f acc{0.f, 0.f};
    static_for<0, M>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        const float t0 = raw_at<M, 2 * (0 - k) + 1>(pc) + raw_at<M, 2 * (0 - (2 * M - 1) + k) + 1>(pc);
        const float t1 = raw_at<M, 2 * (1 - k) + 1>(pc) + raw_at<M, 2 * (1 - (2 * M - 1) + k) + 1>(pc);
        const v2f p = v2f{t0, t1} * L::tap(0, k);
        if constexpr (k == 0)
            acc = p;
        else
            acc = acc + p;
    });
    acc.x = acc.x + raw_at<M, 2 * (0 - (M - 1))>(pc);
    acc.y = acc.y + raw_at<M, 2 * (1 - (M - 1))>(pc);
    return acc;
}

// Stage s >= 1 with P = 4 or 2 outputs per thread: window words w[j] = O_s element P t - Ho + j, pairs of neighbouring
// outputs.  sink(q, y) receives outputs P t + 2q and P t + 2q + 1.
template <class L, int s, class Sink>
__device__ __forceinline__ void stage_pairs(const float *str, int lid, Sink &&sink)
{
    constexpr int M = L::M(s), P = L::P(s);
    static_assert(P == 4 || P == 2, "pairs of neighbouring outputs");
    constexpr int NWIN = 2 * M + P - 1, NV = (NWIN + P - 1) / P;
    float w[NV * P], e[P];
    const float *O = str + L::offO(s) + P * lid;
    const float *E = str + L::offE(s) + P * lid;
    if constexpr (P == 4) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const v4f t = reinterpret_cast<const v4f *>(O)[v];
            w[4 * v] = t.x, w[4 * v + 1] = t.y, w[4 * v + 2] = t.z, w[4 * v + 3] = t.w;
        }
        const v4f t = *reinterpret_cast<const v4f *>(E);
        e[0] = t.x, e[1] = t.y, e[2] = t.z, e[3] = t.w;
    } else {
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const v2f t = lds_read_b64(O + 2 * v);
            w[2 * v] = t.x, w[2 * v + 1] = t.y;
        }
        const v2f t = *reinterpret_cast<const v2f *>(E);
        e[0] = t.x, e[1] = t.y;
    }
    static_for<0, P / 2>([&](auto q_) {
        constexpr int q = decltype(q_)::value;
        v2f acc{0.f, 0.f};
        static_for<0, M>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            const float t0 = w[2 * q + 2 * M - 1 - k] + w[2 * q + k];
            const float t1 = w[2 * q + 1 + 2 * M - 1 - k] + w[2 * q + 1 + k];
            const v2f p = v2f{t0, t1} * L::tap(s, k);
            if constexpr (k == 0)
                acc = p;
            else
                acc = acc + p;
        });
        acc = acc + v2f{e[2 * q], e[2 * q + 1]};
        sink(q_, acc);
    });
}

// Stage s >= 1 with one output per thread (N(s) <= 64): pairs run along the taps.  Window words w[j] = O_s element
// t - Ho + j, read as aligned pairs from the copy whose alignment matches the thread's parity.  With wp[m] =
// (w[2m], w[2m+1]): old pair (k, k+1) = wp[k/2], new pair = wp[M-1-k/2] reversed (op_sel).  Returns y[t].
template <class L, int s>
__device__ __forceinline__ float stage_single(const float *str, int lid)
{
    constexpr int M = L::M(s);
    const float *O = str + ((lid & 1) ? L::offOB(s) + 1 + lid : L::offO(s) + lid);
    v2f wp[M];
#pragma unroll
    for (int m = 0; m < M; m++) wp[m] = lds_read_b64(O + 2 * m);
    const float ev = str[L::offE(s) + lid];
    float acc = 0.f;
    static_for<0, M / 2>([&](auto h_) {
        constexpr int h = decltype(h_)::value, k = 2 * h;
        const v2f nw = wp[M - 1 - h], od = wp[h];
        const v2f t = v2f{nw.y, nw.x} + od;
        const v2f p = t * v2f{L::tap(s, k), L::tap(s, k + 1)};
        if constexpr (k == 0)
            acc = p.x;
        else
            acc = acc + p.x;
        acc = acc + p.y;
    });
    if constexpr (M % 2 == 1) {
        const v2f mid = wp[(M - 1) / 2];  // (w[M-1], w[M]) = (old, new) of the centre-most tap
        const float p = (mid.y + mid.x) * L::tap(s, M - 1);
        if constexpr (M == 1)
            acc = p;
        else
            acc = acc + p;
    }
    return acc + ev;
}

// ---------------------------------------------------------------------------------------- one wave's cascade
// Everything a wave does between taking its pieces out of the ring and handing a round's outputs over.
//
// Stage schedule ("skew").  Stage 0 runs slot by slot; if stages 1 .. S-1 all followed the fourth slot, the wave would
// consume its four slots in a burst and then compute for two thirds of the round — the request issued after the first
// slot would be needed three short steps later (measured: it has not landed, 0.92 ms where the arithmetic alone takes
// 0.57 and the requests alone 0.70).  So the later stages are spread over the slot steps: stage 1 follows slot 3 of its
// own round, stage s >= 2 follows slot s - 2 of the NEXT round (working on the previous round's streams, which nothing
// has touched since).  Every request then has about two thirds of a round to land.  The streams need no extra space;
// the last stage's outputs of round c leave during round c + 1, and a short epilogue finishes the last round.
template <class L>
struct WaveCascade {
    static constexpr int S = L::stages, M0 = L::M(0);
    static constexpr int roll_words(int s) { return L::He(s) + L::Ho(s) * (L::dual(s) ? 2 : 1); }
    static constexpr int roll_pt()
    {
        int m = 1;
        for (int s = 1; s < S; s++) m = (roll_words(s) + kW - 1) / kW > m ? (roll_words(s) + kW - 1) / kW : m;
        return m;
    }
    static constexpr int PT = roll_pt();
    // slot step after which stage s runs (s >= 1), and whether it then works on the previous round
    static constexpr int step_of(int s) { return s == 1 ? kSlots - 1 : s - 2; }
    static constexpr bool lagging(int s) { return s >= 2; }

    float *str;  // this wave's streams
    uint32_t str_lds;  // ... as an LDS byte address
    int lid;
    int rdst[S > 1 ? S : 2][PT];  // history roll: destination word of this thread's k-th word of stage s

    __device__ __forceinline__ void init(float *streams, int lane_id)
    {
        str = streams;
        str_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)streams;
        lid = lane_id;
        static_for<1, S>([&](auto s) {
            constexpr int s_ = decltype(s)::value;
#pragma unroll
            for (int k = 0; k < PT; k++) {
                const int j = lid + k * kW;
                int d = L::offE(s_) + j;
                if (j >= L::He(s_)) d = L::offO(s_) + (j - L::He(s_));
                if constexpr (L::dual(s_)) {
                    if (j >= L::He(s_) + L::Ho(s_)) d = L::offOB(s_) + 1 + (j - L::He(s_) - L::Ho(s_));
                }
                rdst[s_][k] = d;
            }
        });
    }

    // After stage s has consumed n_s = n >> (s + 1) input pairs, stream words [n_s, n_s + H) become the history [0, H)
    // (src/hbf.rs:182-183 `copy_within`).  n = raw samples of the round; FULL: n == kSC.
    template <int s, bool FULL>
    __device__ __forceinline__ void roll(int n) const
    {
        constexpr int total = roll_words(s), pt = (total + kW - 1) / kW;
        const int ns = FULL ? L::N(s) : n >> (s + 1);
        float t[pt];
#pragma unroll
        for (int k = 0; k < pt; k++)
            if ((k + 1) * kW <= total || lid + k * kW < total) t[k] = str[rdst[s][k] + ns];
        lds_wave_sync();
#pragma unroll
        for (int k = 0; k < pt; k++)
            if ((k + 1) * kW <= total || lid + k * kW < total) str[rdst[s][k]] = t[k];
    }

    // history of stages >= 1 <- state words (per stage: even[M-1] then odd[2M-1], oldest first; SoA across lanes)
    __device__ __forceinline__ void load_state(const uint32_t *st, size_t lanes, size_t lane)
    {
        static_for<1, S>([&](auto s) {
            constexpr int s_ = decltype(s)::value;
            constexpr int He = L::He(s_), Ho = L::Ho(s_), so = L::state_off(s_);
            if (lid < He) str[L::offE(s_) + lid] = __uint_as_float(st[size_t(so + lid) * lanes + lane]);
            if (lid < Ho) {
                const float v = __uint_as_float(st[size_t(so + He + lid) * lanes + lane]);
                str[L::offO(s_) + lid] = v;
                if constexpr (L::dual(s_)) str[L::offOB(s_) + 1 + lid] = v;
            }
        });
    }
    __device__ __forceinline__ void store_state(uint32_t *st, size_t lanes, size_t lane) const
    {
        static_for<1, S>([&](auto s) {
            constexpr int s_ = decltype(s)::value;
            constexpr int He = L::He(s_), Ho = L::Ho(s_), so = L::state_off(s_);
            if (lid < He) st[size_t(so + lid) * lanes + lane] = __float_as_uint(str[L::offE(s_) + lid]);
            if (lid < Ho) st[size_t(so + He + lid) * lanes + lane] = __float_as_uint(str[L::offO(s_) + lid]);
        });
    }

    // stage 0 of slot q: y = outputs 2t, 2t+1 of the slot -> stage 1's streams (`ChunkIn<_, 2>`: consecutive outputs
    // pair up as the next stage's [even, odd])
    // The same two stores as `ds_write_addtid_b32` (LDS address = M0 + offset + 4 * lane: no address VGPR, 2 LDS cycles where `ds_write_b32` takes 4 —
    // MI355X_MICROARCH.md, LDS): consecutive threads write consecutive words of stage 1's streams, which is exactly that form.  Sixteen waves share the
    // CU's LDS pipe and it is the busiest unit of this kernel (NOTES round 5: 8 of these stores per wave and round = 32 of ~ 350 cycles).
    template <int Q>
    __device__ __forceinline__ void put_stage0_tid(v2f yv) const
    {
        static_assert(S >= 2, "a one-stage cascade stores stage 0 directly");
        // M0 carries the wave's LDS base here, and in the FrameMajor kernel that base lies above 64 KiB (the streams start behind a
        // 69632-byte ring): the form relies on the DS add-TID address being M0 + offset + 4 * lane with ALL of M0's bits, as gfx950 does it
        // (the C3 FrameMajor parity tests compare every output through this path); GFX9-era documents describe M0[15:0] for these
        // instructions, so any other target takes the plain stores.
#if !defined(__gfx950__)
        put_stage0(Q, yv);
#else
        if constexpr (L::dual(1)) {
            put_stage0(Q, yv);
        } else {
            constexpr int offe = (L::offE(1) + L::He(1) + Q * (kSlotW / 4)) * 4, offo = (L::offO(1) + L::Ho(1) + Q * (kSlotW / 4)) * 4;
            static_assert(offe >= 0 && offo < 65536, "16-bit instruction offsets");
            unsigned keep;
            // (trailing s_nop: a compiler-placed M0 reader right behind the statement would sit in the s_mov-to-M0 hazard window)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tds_write_addtid_b32 %1 offset:%4\n\tds_write_addtid_b32 %2 offset:%5\n\ts_mov_b32 m0, %0\n\ts_nop 0"
                         : "=&s"(keep)
                         : "v"(yv.x), "v"(yv.y), "s"(__builtin_amdgcn_readfirstlane(str_lds)), "i"(offe), "i"(offo)
                         : "memory");
        }
#endif
    }
    __device__ __forceinline__ void put_stage0(int q, v2f yv) const
    {
        static_assert(S >= 2, "a one-stage cascade stores stage 0 directly");
        const int j = q * (kSlotW / 4) + lid;
        str[L::offE(1) + L::He(1) + j] = yv.x;
        str[L::offO(1) + L::Ho(1) + j] = yv.y;
        if constexpr (L::dual(1)) str[L::offOB(1) + L::Ho(1) + 1 + j] = yv.y;
    }

    // stage s (1 .. S-1) of one round; the last stage calls out(p, v) with the thread's p-th output = round output P t + p
    template <int s, class Out>
    __device__ __forceinline__ void stage(Out &&out) const
    {
        constexpr int P = L::P(s);
        lds_wave_sync();  // the producer's stream writes are issued (LDS operations of one wave execute in order)
        if constexpr (P >= 2) {
            stage_pairs<L, s>(str, lid, [&](auto q_, v2f yv) {
                constexpr int q = decltype(q_)::value;
                if constexpr (s + 1 == S) {
                    out(std::integral_constant<int, 2 * q>{}, yv.x);
                    out(std::integral_constant<int, 2 * q + 1>{}, yv.y);
                } else {
                    const int j = (P / 2) * lid + q;
                    str[L::offE(s + 1) + L::He(s + 1) + j] = yv.x;
                    str[L::offO(s + 1) + L::Ho(s + 1) + j] = yv.y;
                    if constexpr (L::dual(s + 1)) str[L::offOB(s + 1) + L::Ho(s + 1) + 1 + j] = yv.y;
                }
            });
        } else {
            if (L::N(s) >= kW || lid < L::N(s)) {
                const float yv = stage_single<L, s>(str, lid);
                if constexpr (s + 1 == S) {
                    out(std::integral_constant<int, 0>{}, yv);
                } else {
                    const int j = lid >> 1;
                    if (lid & 1) {
                        str[L::offO(s + 1) + L::Ho(s + 1) + j] = yv;
                        if constexpr (L::dual(s + 1)) str[L::offOB(s + 1) + L::Ho(s + 1) + 1 + j] = yv;
                    } else {
                        str[L::offE(s + 1) + L::He(s + 1) + j] = yv;
                    }
                }
            }
        }
    }
};

// Request / wait bookkeeping.  Every wave issues exactly ONE request per slot and the output stores of a round at a
// fixed place, so "my request for slot q has landed" is `s_waitcnt vmcnt(N)` with N = the vector-memory operations the
// wave has issued since (requests and stores retire in issue order).  Two forms of the round body:
//   FAST  whole rounds in the middle of the stream: static N, unclamped requests off an SGPR base
//   SAFE  the first two rounds (the store of "the round before" is missing from the queue) and the last two (their
//         requests reach past the data, the last one is ragged): vmcnt(0), requests clamped to a valid address,
//         runtime roll, state write-back.
// Stores: S = 1 after every slot step; S = 2 after step 3 (stage 1 is the last stage); S >= 3 after step S - 3 of the
// following round (the last stage lags, see WaveCascade).
template <int S>
constexpr int store_step() { return S == 2 ? kSlots - 1 : S - 3; }

// ============================================================================================== FRAME_MAJOR
// x[(f*lanes + lane)*16 + k], y[f*lanes + lane]; /16 cascades (64-byte frames), 16 lanes = 16 waves per workgroup.
// LDS: [ring 64 rows x kFmPitch][16 x streams][tile 64 x 17].  Ring row 16 q + r = frame r of slot q, all 16 lanes (wave r
// requests it: 1 KiB of contiguous global memory); a lane's piece g of the round sits in row g >> 2 at
// lane * 64 + (g & 3) * 16.  Rows are padded by 64 bytes so that the 16 threads of a `ds_read_b128` group — pieces of
// four different frames — fall into four different bank quarters.
// Step I = 4 c + q: wait for the own request of slot q, LDS barrier (everybody's rows of slot q have landed, everybody
// has finished stage 0 of slot q - 1), request number I + 3 (the refill of slot q - 1) — except the wave that owns row
// 15, whose last frame the threads 0 .. 3 of the NEXT slot still read: it stays one slot behind (number I + 2).  The
// last stage (step 1 of the following round) puts the round's 64 output frames into the tile; they are stored behind
// the barrier of step 2, one 64-byte piece per thread of the first 16 of every wave.  Younger than request I when its
// wait comes: the requests I + 1, I + 2 (late wave: I + 1) and the store if step 2 fell in between.
constexpr int kFmLanes = 16;
constexpr int kFmPitch = kFmLanes * 64 + 64;  // bytes per ring row
constexpr int kFmRows = kSlots * 16;
constexpr int kFmTilePitch = kFmLanes + 1;    // words per tile row: conflict-free column writes

template <class L>
__global__ __launch_bounds__(kFmLanes *kW) void hbf_dec_ring_fm(uint32_t *st, const float *x, float *y, const size_t lanes,
                                                                 const size_t frames)
{
    extern __shared__ __attribute__((aligned(16))) float smem_fm[];
    using WC = WaveCascade<L>;
    constexpr int S = L::stages, R = L::rate, M0 = L::M(0);
    static_assert(R == 16 && M0 <= 4 && S == 4, "64-byte frames whose stage-0 history fits the previous frame");
    constexpr int NOUT = kSC / R;  // 64 output frames per round
    constexpr int kStoreStep = WC::step_of(S - 1) + 1;  // the step whose barrier follows the last stage
    const int lid = threadIdx.x % kW, w = __builtin_amdgcn_readfirstlane(threadIdx.x / kW);
    const size_t ngroups = lanes / kFmLanes, per = (ngroups + 7) / 8;
    const size_t group = (blockIdx.x % 8) * per + blockIdx.x / 8;  // every XCD a contiguous eighth of the lane groups
    if (group >= ngroups) return;
    const size_t lane0 = group * kFmLanes, lane = lane0 + w;
#ifdef IDSP_EXP_HBF_CLK
    const long long clk_s0 = clock64(), clk_r0 = wall_clock64();
#endif
    char *ringb = reinterpret_cast<char *>(smem_fm);
    float *str = smem_fm + kFmRows * kFmPitch / 4 + w * up4(L::words);
    float *tile = smem_fm + kFmRows * kFmPitch / 4 + kFmLanes * up4(L::words);
    WC wc;
    wc.init(str, lid);

    // byte address of raw sample n of this lane (n relative to slot 0 of the round, negative ones wrap)
    auto raw_byte = [&](int n) -> int {
        const int m = (n + kSC) % kSC, g = m >> 2;
        return (g >> 2) * kFmPitch + w * 64 + (g & 3) * 16 + (m & 3) * 4;
    };
    {
        constexpr int He = L::He(0), Ho = L::Ho(0);
        if (lid < He) *reinterpret_cast<float *>(ringb + raw_byte(2 * (lid - He))) = __uint_as_float(st[size_t(lid) * lanes + lane]);
        if (lid < Ho) *reinterpret_cast<float *>(ringb + raw_byte(2 * (lid - Ho) + 1)) = __uint_as_float(st[size_t(He + lid) * lanes + lane]);
    }
    wc.load_state(st, lanes, lane);

    int pa[M0 + 1], pa0[M0 + 1];  // bytes
#pragma unroll
    for (int i = 0; i <= M0; i++) {
        const int g = lid - M0 + i;
        pa[i] = (g >> 2) * kFmPitch + w * 64 + (g & 3) * 16;  // g >> 2 is -1 for the four pieces before the slot
        pa0[i] = (((g + kSC / 4) % (kSC / 4)) >> 2) * kFmPitch + w * 64 + (g & 3) * 16;
    }

    const bool late = w == kFmLanes - 1;  // owns row 15 of every slot
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)ringb;
    const uint32_t voff = uint32_t(lid) * 16;
    const size_t fpitch = lanes * size_t(R);  // floats per frame row
    // request number m = (round m / 4, slot m % 4): frame 16 m + w, all 16 lanes; SAFE: past the end -> frame 0
    auto request = [&](auto safe_, size_t m) {
        const size_t f = m * 16 + size_t(w);
        const float *src = uniform_ptr(x + (!decltype(safe_)::value || f < frames ? f : 0) * fpitch + lane0 * R);
#ifndef IDSP_EXP_HBF_NOLOAD
        glds16_si<0>(src, voff, ring_lds + uint32_t((int(m % kSlots) * 16 + w) * kFmPitch));
#endif
    };
    // a round's 64 output frames: wave w stores frames 4w .. 4w+3 as 64-byte pieces (16 threads)
    auto store_tile = [&](size_t c, int nout) {
        if (lid < 16) {
            const int r = 4 * w + (lid >> 2), j = lid & 3;
            if (r < nout) {
                const float *t = tile + r * kFmTilePitch + 4 * j;
                *reinterpret_cast<v4f *>(y + (c * NOUT + size_t(r)) * lanes + lane0 + 4 * j) = v4f{t[0], t[1], t[2], t[3]};
            }
        }
    };
    auto to_tile = [&](auto, float v) { tile[lid * kFmTilePitch + w] = v; };  // one output per thread: frame t of the round
    auto stages_after = [&](auto q_, auto safe_, size_t c, int n) {
        constexpr int q = decltype(q_)::value;
        constexpr bool SAFE = decltype(safe_)::value;
        static_for<1, S>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            if constexpr (WC::step_of(s) == q) {
                if constexpr (!WC::lagging(s)) {
                    wc.template stage<s>(to_tile);
                    wc.template roll<s, !SAFE>(n);
                } else {
                    if (c != 0) {
                        wc.template stage<s>(to_tile);
                        wc.template roll<s, true>(kSC);
                    }
                }
            }
        });
    };

    const size_t total = frames * size_t(R);
    const size_t rounds = (total + kSC - 1) / kSC;
    auto round = [&](auto safe_, size_t c) {
        constexpr bool SAFE = decltype(safe_)::value;
        const bool last = SAFE && c + 1 == rounds;
        const int n = last ? int(total - c * kSC) : kSC;
        static_for<0, kSlots>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            if constexpr (SAFE) {
                wait_vmcnt<0>();
            } else {
                if (late)
                    wait_vmcnt<(q == (kStoreStep + 1) % kSlots || q == (kStoreStep + 2) % kSlots ? 2 : 1)>();
                else
                    wait_vmcnt<(q == kStoreStep ? 2 : 3)>();
            }
            lds_barrier();
            request(safe_, c * kSlots + q + (late ? 2 : 3));
            if constexpr (q == kStoreStep) {
                if (c != 0) store_tile(c - 1, NOUT);
            }
            v4f pc[M0 + 1];
#pragma unroll
            for (int i = 0; i <= M0; i++) pc[i] = *reinterpret_cast<const v4f *>(ringb + (q == 0 ? pa0[i] : pa[i] + q * 16 * kFmPitch));
            lds_wave_sync();
            if constexpr (SAFE) {
                if (last && q == (n - 1) / kSlotW) {
                    constexpr int He = L::He(0), Ho = L::Ho(0);
                    if (lid < He) st[size_t(lid) * lanes + lane] = __float_as_uint(*reinterpret_cast<const float *>(ringb + raw_byte(n + 2 * (lid - He))));
                    if (lid < Ho) st[size_t(He + lid) * lanes + lane] = __float_as_uint(*reinterpret_cast<const float *>(ringb + raw_byte(n + 2 * (lid - Ho) + 1)));
                    lds_wave_sync();
                }
            }
#ifdef IDSP_EXP_HBF_NOSTAGES
            if (pc[M0].x == 12345.678f) y[0] = pc[0].y;
#else
            wc.template put_stage0_tid<q>(stage0_pair<L>(pc));
            stages_after(q_, safe_, c, n);
#endif
        });
    };
    request(std::true_type{}, 0);
    request(std::true_type{}, 1);
    if (!late) request(std::true_type{}, 2);
    for (size_t c = 0; c < rounds; c++) {
        if (c < 2 || c + 2 >= rounds)
            round(std::true_type{}, c);
        else
            round(std::false_type{}, c);
    }
    const int n_last = int(total - (rounds - 1) * kSC);
#ifndef IDSP_EXP_HBF_NOSTAGES
    static_for<2, S>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        wc.template stage<s>(to_tile);
        wc.template roll<s, false>(n_last);
    });
#endif
    wait_vmcnt<0>();  // the requests past the end of the data are still landing: they must not outlive the workgroup's LDS
    lds_barrier();    // every wave's column of the last tile
    store_tile(rounds - 1, n_last / R);
    wc.store_state(st, lanes, lane);
#ifdef IDSP_EXP_HBF_CLK
    if (threadIdx.x == 0 && (group == 0 || group == ngroups / 2 || group + 1 == ngroups))
        printf("clk fm group %d: %lld shader cycles in %lld x 10 ns\n", int(group), clock64() - clk_s0, wall_clock64() - clk_r0);
#endif
}

// -------------------------------------------------------------------------------------------------------- host
template <int TS, int S>
int launch_ring(uint32_t *st, const float *x, float *y, size_t lanes, size_t frames, bool lm, hipStream_t stream)
{
    using L = Lay<TS, S>;
    if (lm) return 1;  // LaneMajor: the register-blocked kernel of hbf_blk.h (round 6)
    if constexpr (L::rate == 16 && L::M(0) <= 4) {
        if (lanes % kFmLanes != 0) return 1;
        constexpr size_t bytes = size_t(kFmRows) * kFmPitch + (size_t(kFmLanes) * up4(L::words) + size_t(kSC / L::rate) * kFmTilePitch) * sizeof(float);
        static_assert(bytes <= 160 * 1024, "one workgroup per CU");
        if (ensure_dyn_lds<&hbf_dec_ring_fm<L>>(bytes)) return 2;
        const size_t ngroups = lanes / kFmLanes;
        note_kernel("hbf_dec_ring[FrameMajor]", typeid(L).name());
        hipLaunchKernelGGL((hbf_dec_ring_fm<L>), dim3(unsigned(8 * ((ngroups + 7) / 8))), dim3(kFmLanes * kW), bytes, stream, st, x, y, lanes,
                           frames);
        return 0;
    }
    return 1;
}

}  // namespace hbfr

// Returns 0 when a ring kernel was launched, 1 when the request is not covered (the caller falls back to
// hbf_wave.h), 2 on a HIP error (idsp_last_error() holds the text).
int hbf_ring_dec(int tap_set, int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames,
                 bool lane_major, hipStream_t stream);

}  // namespace idsp
