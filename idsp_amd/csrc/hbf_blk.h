// hbf_blk.h — half-band decimator cascades (HBF_DEC_CASCADE over HBF_TAPS / HBF_TAPS_98, src/hbf.rs:142-192,385-421),
// register-blocked: round 6 successor of the slot-wise ring kernels of hbf_ring.h for the shapes it covers.
//
// Why.  tools/ubench_lds_issue.hip (profiles/r06_ubench_lds_issue.txt) settled what rounds 4 and 5 argued about: LDS and
// VALU instructions of a CU's waves DO issue side by side (16 packed multiplies + 16 `ds_read_b128` per iteration take
// as long as the reads alone), an LDS read costs the CU's one LDS pipe time in proportion to its BYTES (b64 1.5, b128
// 2.8 ticks per wave instruction), and a 4- or 8-byte-misaligned b64 / b128 costs 43 ticks.  Priced that way the ring
// kernel's round (1024 raw samples of one lane) is 172 LDS ticks x 18 waves per CU against 200 VALU instructions x 4.5
// waves per SIMD at 2.9 ticks: the LDS pipe, not issue, was the busiest unit, and most of its bytes were re-reads — every
// thread fetched 20 raw words to produce two stage-0 outputs, and the lowest-rate stages read a 2M-word window per
// single output.  Here every thread reads a window ONCE for four or eight neighbouring outputs:
//
//  * stage 0: thread t owns 16 consecutive raw samples (four 16-byte pieces) of the round and reads them plus the M
//    pieces before them: M + 4 `ds_read_b128` for 8 outputs (ring kernel: 4 (M + 1) for 8).  The whole round (4 KiB) is
//    requested at once, right after the previous round's pieces are in registers, and has a full round of arithmetic to
//    land.  A thread's own pieces are 64 bytes apart in memory — a four-way bank conflict for `ds_read_b128` — so the
//    request permutes on the global side (as cic_ring.h does): lane j of the request for KiB k fetches piece
//    4 (j % 16) + j / 16 of that KiB, which puts piece i of thread 16 k + u at 64 k + 16 i + u: consecutive threads read
//    consecutive pieces, every request still covers one whole KiB.
//  * stages s >= 1 run on RUNS of 64 P input pairs, P = 4 (or 2) outputs per thread: a stage that receives fewer pairs
//    per round waits B = 64 P / pairs-per-round rounds and then produces P outputs per thread from one window of
//    2M - 1 + P words read as aligned vectors (stage 3 of /16: 13 `ds_read_b128` per 4 outputs where the ring kernel read
//    23 `ds_read_b64` per single output).  Streams keep their history in front, padded to a multiple of P words so that
//    both the producer's vector writes and the window reads are aligned (misaligned ones are 15 x slower, see above).
//  * arithmetic as in hbf_ring.h: symmetric sums scalar, tap multiplies and the sequential accumulation packed over pairs
//    of neighbouring outputs; every IEEE operation and its order per output unchanged (0 ULP against the oracle).
//  * the last stage's outputs leave as 16 bytes per thread, one contiguous KiB per wave and run.
//
// One code path with run-time counts: rounds past the regular ones (the last two: requests reach the end of the row, the
// last round may be short, partially filled runs are flushed) differ only in the request form (clamped addresses,
// `vmcnt(0)`), the predicates of the output stores and the roll distances — all wave-uniform scalars.
#pragma once

#include "hbf_ring.h"

#ifndef IDSP_HBF_BLK_NTSTORE  // 1: the LaneMajor outputs leave as nontemporal stores
#define IDSP_HBF_BLK_NTSTORE 0
#endif
#ifndef IDSP_HBF_BLK_SPREAD  // 1: the four requests of the next round one by one between the stage-0 output pairs; 0: all four before stage 0
#define IDSP_HBF_BLK_SPREAD 0
#endif

namespace idsp {
namespace hbfb {

using hbfr::kSC;
using hbfr::kW;
using hbfr::lds_read_b64;
using hbfr::static_for;
using hbfr::up4;
using hbfr::v2f;
using hbfr::v4f;

constexpr int upn(int v, int n) { return (v + n - 1) / n * n; }
constexpr int kOwn = kSC / 4 / kW;  // 16-byte pieces per thread and round
static_assert(kOwn == 4, "a thread owns one 64-byte run of the round");

// compiler-only fence between the LDS stores of one stage and the LDS loads of the next: LDS operations of a wave
// execute in order, so no wait is needed, only that hipcc does not move accesses across
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

// Cascade geometry.  PM: bit s set = stage s (>= 1) runs at four outputs per thread (runs of 256 pairs), clear = at two (runs
// of 128 pairs: half the stream space).  LaneMajor: all four; FrameMajor (16 lanes' streams in one workgroup's LDS): stage 1 only.
constexpr int kPmAll = 0x3e, kPmFirst = 0x02;
template <int TS, int S, int PM>
struct BLay {
    static constexpr int stages = S;
    static constexpr int rate = 1 << S;
    static constexpr int M(int s) { return kHbfM[TS][hbf_tuple_index(true, S, s)]; }
    static constexpr float tap(int s, int k) { return kHbfTaps[TS][hbf_tuple_index(true, S, s)][k]; }
    static constexpr int He(int s) { return M(s) - 1; }
    static constexpr int Ho(int s) { return 2 * M(s) - 1; }
    static constexpr int P(int s) { return s == 0 ? 8 : ((PM >> s) & 1 ? 4 : 2); }  // outputs per thread and run
    static constexpr int L(int s) { return kW * P(s); }           // s >= 1: input pairs = outputs per run
    static constexpr int Nr(int s) { return kSC >> (s + 1); }     // pairs arriving per round
    static constexpr int B(int s) { return L(s) / Nr(s); }        // rounds per run
    // history in front of the samples, right-aligned in a field of a multiple of P words: sample i of the run at word HxP + i
    static constexpr int HeP(int s) { return upn(He(s), P(s)); }
    static constexpr int HoP(int s) { return upn(Ho(s), P(s)); }
    static constexpr int sizeE(int s) { return up4(HeP(s) + L(s) + P(s)); }
    static constexpr int sizeO(int s) { return up4(HoP(s) + L(s)); }
    static constexpr int offE(int s)
    {
        int o = 0;
        for (int t = 1; t < s; t++) o += sizeE(t) + sizeO(t);
        return o;
    }
    static constexpr int offO(int s) { return offE(s) + sizeE(s); }
    static constexpr int words = offE(S);
    static constexpr int state_off(int s)
    {
        int o = 0;
        for (int t = 0; t < s; t++) o += 3 * M(t) - 2;
        return o;
    }
    static constexpr int roll_words(int s) { return He(s) + Ho(s); }
    static constexpr int roll_pt()
    {
        int m = 1;
        for (int s = 1; s < S; s++) m = (roll_words(s) + kW - 1) / kW > m ? (roll_words(s) + kW - 1) / kW : m;
        return m;
    }
};

// Stage s >= 1 on one run: thread t's P outputs P t .. P t + P - 1 of the run.  Window words w[j] = O_s word P t + j =
// sample P t - HoP + j; with dO = HoP - Ho, output P t + i needs the samples (P t + i) - k -> w[dO + i + 2M - 1 - k] and
// (P t + i) - (2M - 1) + k -> w[dO + i + k]  (src/hbf.rs:46-68,163-185); the even sample (P t + i) - (M - 1) is E_s word
// P t + dE + i.  sink(q, y) receives the outputs P t + 2 q and P t + 2 q + 1.
template <class L, int s, class Sink>
__device__ __forceinline__ void run_stage(const float *str, int lid, Sink &&sink)
{
    constexpr int M = L::M(s), P = L::P(s);
    static_assert(P == 4 || P == 2, "pairs of neighbouring outputs");
    constexpr int dO = L::HoP(s) - L::Ho(s), dE = L::HeP(s) - L::He(s);
    constexpr int NV = (L::HoP(s) + P) / P, NE = dE == 0 ? 1 : 2;
    float w[NV * P], e[NE * P];
    const float *O = str + L::offO(s) + P * lid;
    const float *E = str + L::offE(s) + P * lid;
    if constexpr (P == 4) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const v4f t = reinterpret_cast<const v4f *>(O)[v];
            w[4 * v] = t.x, w[4 * v + 1] = t.y, w[4 * v + 2] = t.z, w[4 * v + 3] = t.w;
        }
#pragma unroll
        for (int v = 0; v < NE; v++) {
            const v4f t = reinterpret_cast<const v4f *>(E)[v];
            e[4 * v] = t.x, e[4 * v + 1] = t.y, e[4 * v + 2] = t.z, e[4 * v + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const v2f t = lds_read_b64(O + 2 * v);
            w[2 * v] = t.x, w[2 * v + 1] = t.y;
        }
#pragma unroll
        for (int v = 0; v < NE; v++) {
            const v2f t = lds_read_b64(E + 2 * v);
            e[2 * v] = t.x, e[2 * v + 1] = t.y;
        }
    }
    static_for<0, P / 2>([&](auto q_) {
        constexpr int q = decltype(q_)::value;
        v2f acc{0.f, 0.f};
        static_for<0, M>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            const float t0 = w[dO + 2 * q + 2 * M - 1 - k] + w[dO + 2 * q + k];
            const float t1 = w[dO + 2 * q + 1 + 2 * M - 1 - k] + w[dO + 2 * q + 1 + k];
            const v2f p = v2f{t0, t1} * L::tap(s, k);
            if constexpr (k == 0)
                acc = p;  // the sum starts from -0.0 (f32::sum) and -0.0 + p == p
            else
                acc = acc + p;
        });
        acc = acc + v2f{e[dE + 2 * q], e[dE + 2 * q + 1]};
        sink(q_, acc);
    });
}

// ---------------------------------------------------------------------------------------- one wave's cascade
// Stages 1 .. S-1 of one lane: streams, run bookkeeping, history rolls, state.  `pos[s]` = pairs waiting in stage s's
// stream (wave-uniform); a stage runs when its run is complete, or on whatever it holds when the call ends.
template <class L>
struct Cascade {
    static constexpr int S = L::stages, PT = L::roll_pt(), SA = S > 1 ? S : 2;
    float *str;
    int lid;
    int rdst[SA][PT];  // history roll: destination word of this thread's r-th word of stage s
    int pos[SA];

    // histories of the stages >= 1 <- state words (per stage: even[M-1] then odd[2M-1], oldest first; SoA across lanes)
    __device__ __forceinline__ void init(float *streams, int lane_id, const uint32_t *st, size_t lanes, size_t lane)
    {
        str = streams, lid = lane_id;
#pragma unroll
        for (int s = 0; s < SA; s++) pos[s] = 0;
        static_for<1, S>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            constexpr int He = L::He(s), Ho = L::Ho(s), so = L::state_off(s);
            constexpr int eb = L::offE(s) + L::HeP(s) - He, ob = L::offO(s) + L::HoP(s) - Ho;
            if (lid < He) str[eb + lid] = __uint_as_float(st[size_t(so + lid) * lanes + lane]);
            if (lid < Ho) str[ob + lid] = __uint_as_float(st[size_t(so + He + lid) * lanes + lane]);
#pragma unroll
            for (int k = 0; k < PT; k++) {
                const int j = lid + k * kW;
                rdst[s][k] = j < He ? eb + j : ob + (j - He);
            }
        });
    }
    __device__ __forceinline__ void store_state(uint32_t *st, size_t lanes, size_t lane) const
    {
        static_for<1, S>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            constexpr int He = L::He(s), Ho = L::Ho(s), so = L::state_off(s);
            constexpr int eb = L::offE(s) + L::HeP(s) - He, ob = L::offO(s) + L::HoP(s) - Ho;
            if (lid < He) st[size_t(so + lid) * lanes + lane] = __float_as_uint(str[eb + lid]);
            if (lid < Ho) st[size_t(so + He + lid) * lanes + lane] = __float_as_uint(str[ob + lid]);
        });
    }

    // One round's stage-0 outputs (thread t: 8 t + 2 q, 8 t + 2 q + 1 in y0[q]; n raw samples of the round were valid) go to
    // stage 1's streams — consecutive outputs pair up as the next stage's [even, odd] (`ChunkIn<_, 2>`) — then every stage
    // whose run is complete runs (all that hold anything when `last`).  out(ov, nv): the last stage's run, thread t's P
    // outputs P t .. P t + P - 1 of which the run's first nv are valid.
    template <class Out>
    __device__ __forceinline__ void round(const v2f (&y0)[kOwn], int n, bool last, Out &&out)
    {
        static_assert(S >= 2, "a one-stage cascade stores stage 0 directly");
        *reinterpret_cast<v4f *>(str + L::offE(1) + L::HeP(1) + 4 * lid) = v4f{y0[0].x, y0[1].x, y0[2].x, y0[3].x};
        *reinterpret_cast<v4f *>(str + L::offO(1) + L::HoP(1) + 4 * lid) = v4f{y0[0].y, y0[1].y, y0[2].y, y0[3].y};
        int k = n >> 2;  // pairs handed to the next stage in this round
        static_for<1, S>([&](auto s_) {
            constexpr int s = decltype(s_)::value, P = L::P(s);
            pos[s] += k;
            k = 0;
            if (pos[s] == L::L(s) || (last && pos[s] > 0)) {
                const int nv = pos[s];
                lds_order();
                if constexpr (s + 1 == S) {
                    float ov[P];
                    run_stage<L, s>(str, lid, [&](auto q_, v2f yv) {
                        constexpr int q = decltype(q_)::value;
                        ov[2 * q] = yv.x, ov[2 * q + 1] = yv.y;
                    });
                    out(ov, nv);
                } else {
                    // the thread's outputs P t .. P t + P - 1 = the pairs (P / 2) t .. of the next stage's run, behind the pairs already there
                    float *En = str + L::offE(s + 1) + L::HeP(s + 1) + pos[s + 1] + (P / 2) * lid;
                    float *On = str + L::offO(s + 1) + L::HoP(s + 1) + pos[s + 1] + (P / 2) * lid;
                    if constexpr (P == 4) {
                        v2f ev, od;
                        run_stage<L, s>(str, lid, [&](auto q_, v2f yv) {
                            if constexpr (decltype(q_)::value == 0)
                                ev.x = yv.x, od.x = yv.y;
                            else
                                ev.y = yv.x, od.y = yv.y;
                        });
                        *reinterpret_cast<v2f *>(En) = ev;
                        *reinterpret_cast<v2f *>(On) = od;
                    } else {
                        run_stage<L, s>(str, lid, [&](auto, v2f yv) { *En = yv.x, *On = yv.y; });
                    }
                }
                // after nv pairs, stream words [nv, nv + H) become the history [0, H)  (src/hbf.rs:182-183 `copy_within`)
                {
                    constexpr int tot = L::roll_words(s), pt = (tot + kW - 1) / kW;
                    float t[pt];
                    lds_order();
#pragma unroll
                    for (int r = 0; r < pt; r++)
                        if ((r + 1) * kW <= tot || lid + r * kW < tot) t[r] = str[rdst[s][r] + nv];
                    lds_wave_sync();
#pragma unroll
                    for (int r = 0; r < pt; r++)
                        if ((r + 1) * kW <= tot || lid + r * kW < tot) str[rdst[s][r]] = t[r];
                }
                pos[s] = 0;
                k = nv >> 1;
            }
        });
    }
};

// stage 0 of a thread's four own pieces: y0[q] = outputs 8 t + 2 q, 8 t + 2 q + 1 from the pieces q .. q + M0 (hbf_ring.h stage0_pair)
// before(q) runs ahead of pair q: the kernels put one of the next round's four requests there, so that a wave's requests are
// spread over stage 0 instead of queueing behind one another (and behind the other waves' bursts) at the texture addresser
template <class L, class Before>
__device__ __forceinline__ void stage0_own(const v4f *pc, v2f (&y0)[kOwn], Before &&before)
{
    static_for<0, kOwn>([&](auto q_) {
        constexpr int q = decltype(q_)::value;
        before(q_);
        y0[q] = hbfr::stage0_pair<L>(&pc[q]);
    });
}

// Position (in 16-byte pieces) of piece g of the round (0 .. 255) in the LaneMajor ring, see the header: the requests permute
// on the global side so that the four pieces of a thread sit 16 pieces apart and those of neighbouring threads side by side.
__device__ __forceinline__ constexpr int ring_pos(int g) { return (g >> 6) * 64 + (g & 3) * 16 + ((g & 63) >> 2); }

// =============================================================================================== LANE_MAJOR
// x[(lane*frames + f)*R + k], y[lane*frames + f].  One wave per lane, no barriers.
// LDS: [history M0 pieces][ring 4 KiB][streams].  Round c: wait for its four requests, read M0 + 4 pieces, copy the
// round's last M0 pieces to the history slots (they are the next round's pieces -M0 .. -1), request round c + 1 into the
// ring, stage 0, then every stage whose run is complete.
// RR: rounds the ring holds (1: the next round is requested when this one is in registers; 2: the round after next, two rounds of lead)
template <class L, int RR>
__global__ __launch_bounds__(kW) void hbf_dec_blk_lm(uint32_t *st, const float *x, float *y, const size_t lanes, const size_t frames)
{
    extern __shared__ __attribute__((aligned(16))) float smem_blk[];
    constexpr int S = L::stages, R = L::rate, M0 = L::M(0);
    constexpr int HIST = 4 * M0;  // words
    constexpr int NP = M0 + kOwn;
    const int lid = threadIdx.x;
    const size_t lane = blockIdx.x;
    static_assert(RR == 1 || RR == 2, "ring of one or two rounds");
    float *const hist = smem_blk, *const ring = smem_blk + HIST, *const str = ring + RR * kSC;

    // stage-0 history (even[M0-1] then odd[2 M0-1], oldest first) -> the raw positions -1, -2, ... of the history slots
    {
        constexpr int He = L::He(0), Ho = L::Ho(0);
        if (lid < He) hist[HIST + 2 * (lid - He)] = __uint_as_float(st[size_t(lid) * lanes + lane]);
        if (lid < Ho) hist[HIST + 2 * (lid - Ho) + 1] = __uint_as_float(st[size_t(He + lid) * lanes + lane]);
    }
    Cascade<L> cs;
    cs.init(str, lid, st, lanes, lane);

    // word offsets (from `hist`) of the pieces 4 t - M0 .. 4 t + 3 of a round
    int pa[RR][NP];  // [ring half the round sits in]
#pragma unroll
    for (int h = 0; h < NP; h++) {
        const int g = 4 * lid - M0 + h;
#pragma unroll
        for (int r = 0; r < RR; r++) pa[r][h] = g >= 0 ? HIST + r * kSC + 4 * ring_pos(g) : 4 * (g + M0);
    }

    const size_t total = frames * size_t(R);  // raw samples of the lane
    const size_t npieces = total / 4;          // whole 16-byte pieces (the dispatcher guarantees total % 4 == 0)
    const float *xl = x + lane * total;
    float *yl = y + lane * frames;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)ring;
    const int perm = 4 * (lid & 15) + (lid >> 4);  // the piece of a KiB this lane requests
    const uint32_t voff = uint32_t(perm) * 16;
    const size_t rounds = (total + kSC - 1) / kSC;

    // request of KiB k of round c; `fast`: the whole round lies inside the row
    auto request_k = [&](auto k_, size_t c, bool fast) {
#ifndef IDSP_EXP_HBF_NOLOAD
        constexpr int k = decltype(k_)::value;
        const uint32_t dst = ring_lds + uint32_t(c % RR) * (kSC * 4);
        if (fast) {
            const float *xc = uniform_ptr(xl + c * kSC);
            // the instruction offset moves the global AND the LDS address (lds_dma.h)
            glds16_si<k * 1024>(xc, voff, dst);
        } else {
            const size_t pc = c * (kSC / 4) + size_t(64 * k + perm);
            glds16(xl + (pc < npieces ? pc * 4 : 0), dst + uint32_t(k) * 1024);
        }
#endif
    };
    auto request = [&](size_t c, bool fast) { static_for<0, 4>([&](auto k_) { request_k(k_, c, fast); }); };

    size_t out_done = 0;  // outputs of the lane stored so far
    // vector-memory operations issued after the requests of the round about to be consumed (q0) and of the one after it (q1):
    // requests and stores retire in issue order, so "round c has landed" is vmcnt <= q0
    int q0 = 0, q1 = 0;

#ifdef IDSP_EXP_HBF_PHASES  // where one wave's time goes: wait for data / piece reads / request issue / arithmetic (shader-clock ticks)
    long long ph[4] = {0, 0, 0, 0}, tp = clock64();
    const long long ph_wall0 = wall_clock64(), ph_clk0 = tp;
#define IDSP_PH(i) { const long long t_ = clock64(); ph[i] += t_ - tp; tp = t_; }
#else
#define IDSP_PH(i)
#endif
    request(0, kSC <= total);
    if (RR == 2 && rounds > 1) {
        request(1, 2 * kSC <= total);
        q0 = 4;
    }
    for (size_t c = 0; c < rounds; c++) {
        const bool last = c + 1 == rounds;
        const int n = last ? int(total - c * kSC) : kSC;  // raw samples of this round
        IDSP_PH(3)
        // "my four requests have landed": they retire in issue order with the stores behind them
        switch (q0 < 6 ? q0 : 6) {  // a smaller count than needed only waits longer
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<1>(); break;
            case 2: wait_vmcnt<2>(); break;
            case 3: wait_vmcnt<3>(); break;
            case 4: wait_vmcnt<4>(); break;
            case 5: wait_vmcnt<5>(); break;
            default: wait_vmcnt<6>(); break;
        }
        int q2 = 0;  // ... after the requests issued in this round
        IDSP_PH(0)
        v4f pc[NP];
        if (RR == 1 || (c & 1) == 0) {
#pragma unroll
            for (int h = 0; h < NP; h++) pc[h] = *reinterpret_cast<const v4f *>(hist + pa[0][h]);
        } else {
#pragma unroll
            for (int h = 0; h < NP; h++) pc[h] = *reinterpret_cast<const v4f *>(hist + pa[RR - 1][h]);
        }
        lds_wave_sync();
        IDSP_PH(1)
        if (!last) {
            // the round's last M0 pieces are the next round's history
#pragma unroll
            for (int i = 0; i < kOwn; i++) {
                const int g = 4 * lid + i - (kSC / 4 - M0);
                if (g >= 0) *reinterpret_cast<v4f *>(hist + 4 * g) = pc[M0 + i];
            }
        }
        // the requests of round c + RR go into the ring half this round's pieces have just left
        const bool req = c + RR < rounds, req_fast = (c + RR + 1) * kSC <= total;
        if (req) q1 += 4;
#if !IDSP_HBF_BLK_SPREAD
        if (req) request(c + RR, req_fast);
#endif
        IDSP_PH(2)
#ifdef IDSP_EXP_HBF_NOSTAGES
        if (pc[M0].x == 12345.678f) yl[0] = pc[0].y;
#else
        v2f y0[kOwn];
        stage0_own<L>(pc, y0, [&](auto k_) {
#if IDSP_HBF_BLK_SPREAD
            if (req) request_k(k_, c + RR, req_fast);
#endif
        });
        if constexpr (S == 1) {
            const int nv = n / 2, i0 = 8 * lid;
            float *dst = yl + out_done + i0;
            if (i0 + 7 < nv) {
                *reinterpret_cast<v4f *>(dst) = v4f{y0[0].x, y0[0].y, y0[1].x, y0[1].y};
                *reinterpret_cast<v4f *>(dst + 4) = v4f{y0[2].x, y0[2].y, y0[3].x, y0[3].y};
            } else {
#pragma unroll
                for (int q = 0; q < kOwn; q++) {
                    if (i0 + 2 * q < nv) dst[2 * q] = y0[q].x;
                    if (i0 + 2 * q + 1 < nv) dst[2 * q + 1] = y0[q].y;
                }
            }
            out_done += size_t(nv);
            q1 += 2, q2 += 2;
        } else {
            cs.round(y0, n, last, [&](auto &ov, int nv) {
                constexpr int P = L::P(S - 1);
                const int nv_all = nv;
                const int i0 = P * lid;
                float *dst = yl + out_done + i0;
#ifdef IDSP_EXP_HBF_NOSTORE  // timing only: the outputs stay in registers
                if (ov[0] != 12345.678f) nv = 0;
#endif
                if (i0 + P - 1 < nv) {
#if IDSP_HBF_BLK_NTSTORE
                    if constexpr (P == 4)
                        __builtin_nontemporal_store(v4f{ov[0], ov[1], ov[2], ov[3]}, reinterpret_cast<v4f *>(dst));
                    else
                        __builtin_nontemporal_store(v2f{ov[0], ov[1]}, reinterpret_cast<v2f *>(dst));
#else
                    if constexpr (P == 4)
                        *reinterpret_cast<v4f *>(dst) = v4f{ov[0], ov[1], ov[2], ov[3]};
                    else
                        *reinterpret_cast<v2f *>(dst) = v2f{ov[0], ov[1]};
#endif
                } else {
#pragma unroll
                    for (int p = 0; p < P; p++)
                        if (i0 + p < nv) dst[p] = ov[p];
                }
                out_done += size_t(nv_all);
                q1 += 1, q2 += 1;
            });
        }
#endif
        if (RR == 1)
            q0 = q2;
        else
            q0 = q1, q1 = q2;
    }
    lds_wave_sync();
    // state: the last raw samples of the row (stage 0) and the stream histories (stages >= 1)
    {
        constexpr int He = L::He(0), Ho = L::Ho(0);
        const int n = int(total - (rounds - 1) * kSC);
        const int half = int((rounds - 1) % RR) * kSC;  // the ring half the last round sits in
        auto raw_word = [&](int r) { return r >= 0 ? HIST + half + 4 * ring_pos(r >> 2) + (r & 3) : HIST + r; };
        if (lid < He) st[size_t(lid) * lanes + lane] = __float_as_uint(hist[raw_word(n + 2 * (lid - He))]);
        if (lid < Ho) st[size_t(He + lid) * lanes + lane] = __float_as_uint(hist[raw_word(n + 2 * (lid - Ho) + 1)]);
    }
    cs.store_state(st, lanes, lane);
#ifdef IDSP_EXP_HBF_PHASES
    IDSP_PH(3)
    if (lid == 0 && (lane == 0 || lane == lanes / 2 || lane == lanes / 2 + 1 || lane + 1 == lanes))
        printf("phases lm lane %d: wait %lld  pieces %lld  requests %lld  stages %lld ticks; %lld ticks in %lld x 10 ns\n", int(lane), ph[0], ph[1], ph[2], ph[3],
               clock64() - ph_clk0, wall_clock64() - ph_wall0);
#endif
}

// ============================================================================================== FRAME_MAJOR
// x[(f*lanes + lane)*16 + k], y[f*lanes + lane]; /16 cascades (64-byte frames), 16 lanes = 16 waves per workgroup.
// LDS: [65 rows x kFmPitch][16 x streams][tile 128 x 17].  Row r (0 .. 63) = frame r of the round, all 16 lanes: ONE request
// (wave r % 16 issues it) moves that KiB of contiguous global memory; row -1 = the previous round's last frame.  Thread t of
// wave w owns frame t of lane w: the four pieces at row t, column w, with the pieces of row t - 1 as history.  Rows are
// padded by 16 bytes so that the 16 threads of a `ds_read_b128` group — one piece of 16 different rows — fall into 16
// different bank quartets.
// Round c: wait for the own requests, barrier A (every row has landed), store the output tile if the last stage filled it
// in the round before, read the pieces, thread 63 copies its frame to row -1, barrier B (everybody has read: the rows may be
// overwritten), request round c + 1, then the arithmetic of the whole round with no further barrier.  (The ring kernel had
// four barriers per round and one slot of lead; NOTES round 4: the barriers alone were 14 %.)
// NL = 16 or 8 lanes per workgroup.  8: a request moves a 512-byte row (the low 32 lanes of the instruction), a workgroup is 8 waves
// and 76 KiB of LDS, so two of them share a CU and one computes while the other is in its barriers / piece reads / waits; y leaves as
// 32-byte pieces.
constexpr int kFmRows = kSC / 16;             // frames per round

template <class L, int NL>
__global__ __launch_bounds__(NL *kW) void hbf_dec_blk_fm(uint32_t *st, const float *x, float *y, const size_t lanes, const size_t frames)
{
    extern __shared__ __attribute__((aligned(16))) float smem_blk_fm[];
    constexpr int S = L::stages, R = L::rate, M0 = L::M(0);
    static_assert(R == 16 && M0 <= 4 && S == 4, "64-byte frames whose stage-0 history fits the previous frame");
    static_assert(NL == 16 || NL == 8, "lanes per workgroup");
    constexpr int kFmLanes = NL, kFmPitch = NL * 16 + 4 /* words per row */, kFmTilePitch = NL + 1 /* conflict-free column writes */;
    constexpr int RQ = kFmRows / NL;  // requests per wave and round
    constexpr int NP = M0 + kOwn, PL = L::P(S - 1), TROWS = L::L(S - 1);
    const int lid = threadIdx.x % kW, w = __builtin_amdgcn_readfirstlane(threadIdx.x / kW);
    const size_t ngroups = lanes / kFmLanes, per = (ngroups + 7) / 8;
    const size_t group = (blockIdx.x % 8) * per + blockIdx.x / 8;  // every XCD a contiguous eighth of the lane groups
    if (group >= ngroups) return;
    const size_t lane0 = group * kFmLanes, lane = lane0 + w;
    float *const rows = smem_blk_fm + kFmPitch;  // row 0
    float *const str = smem_blk_fm + (kFmRows + 1) * kFmPitch + w * up4(L::words);
    float *const tile = smem_blk_fm + (kFmRows + 1) * kFmPitch + kFmLanes * up4(L::words);

    // word (from `rows`) of raw sample r of this lane, r relative to the round's first sample; negative: the frame before
    auto raw_word = [&](int r) { return (r >> 4) * kFmPitch + w * 16 + (r & 15); };
    {
        constexpr int He = L::He(0), Ho = L::Ho(0);
        if (lid < He) rows[raw_word(2 * (lid - He))] = __uint_as_float(st[size_t(lid) * lanes + lane]);
        if (lid < Ho) rows[raw_word(2 * (lid - Ho) + 1)] = __uint_as_float(st[size_t(He + lid) * lanes + lane]);
    }
    Cascade<L> cs;
    cs.init(str, lid, st, lanes, lane);

    int pa[NP];  // words from `rows`: the last M0 pieces of row t - 1, then the four of row t
#pragma unroll
    for (int h = 0; h < NP; h++) pa[h] = (h < M0 ? (lid - 1) * kFmPitch + 4 * (4 - M0 + h) : lid * kFmPitch + 4 * (h - M0)) + w * 16;

    const uint32_t rows_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)rows;
    const uint32_t voff = uint32_t(lid) * 16;
    const size_t fpitch = lanes * size_t(R);  // floats per frame row
    const size_t rounds = (frames + kFmRows - 1) / kFmRows;
    // wave w requests the rows w, w + NL, w + 2 NL, ... of round c (k = 0 .. RQ - 1); frames past the end -> frame 0
    auto request_k = [&](int k, size_t c, bool fast) {
#ifndef IDSP_EXP_HBF_NOLOAD
        const size_t f = c * kFmRows + size_t(NL * k + w);
        const float *src = uniform_ptr(x + (fast || f < frames ? f : 0) * fpitch + lane0 * R);
        if (NL == 16 || lid < NL * 4) glds16_s(src, voff, rows_lds + uint32_t((NL * k + w) * kFmPitch * 4));
#endif
    };
    auto request = [&](size_t c, bool fast) {
#pragma unroll
        for (int k = 0; k < RQ; k++) request_k(k, c, fast);
    };
    // the last stage's run in the tile: wave w stores the frames RPW w .. RPW w + RPW - 1 as 4 NL-byte pieces
    int tile_nv = 0;
    size_t tile_base = 0, out_done = 0;
    auto store_tile = [&]() {
        constexpr int RPW = TROWS / kFmLanes, PPR = NL / 4;  // rows per wave, 16-byte pieces per row
        static_assert(RPW * PPR <= kW, "one piece per thread");
        if (lid < RPW * PPR) {
            const int r = RPW * w + lid / PPR, j = lid % PPR;
            if (r < tile_nv) {
                const float *t = tile + r * kFmTilePitch + 4 * j;
                *reinterpret_cast<v4f *>(y + (tile_base + size_t(r)) * lanes + lane0 + 4 * j) = v4f{t[0], t[1], t[2], t[3]};
            }
        }
        tile_nv = 0;
    };

#ifdef IDSP_EXP_HBF_PHASES
    long long ph[6] = {0, 0, 0, 0, 0, 0}, tp = clock64();
    const long long ph_wall0 = wall_clock64(), ph_clk0 = tp;
#endif
    request(0, kFmRows <= frames);
    for (size_t c = 0; c < rounds; c++) {
        const bool last = c + 1 == rounds;
        const int n = last ? int((frames - c * kFmRows) * R) : kSC;  // raw samples of this round
        IDSP_PH(5)
        wait_vmcnt<0>();
        IDSP_PH(0)
        lds_barrier();  // A
        IDSP_PH(1)
        if (tile_nv > 0) store_tile();
        v4f pc[NP];
#pragma unroll
        for (int h = 0; h < NP; h++) pc[h] = *reinterpret_cast<const v4f *>(rows + pa[h]);
        lds_wave_sync();
        if (!last && lid == kW - 1) {
#pragma unroll
            for (int h = 0; h < M0; h++) *reinterpret_cast<v4f *>(rows - kFmPitch + w * 16 + 4 * (4 - M0 + h)) = pc[kOwn + h];
        }
        IDSP_PH(2)
        lds_barrier();  // B
        IDSP_PH(3)
        const bool req_fast = (c + 2) * kFmRows <= frames;
#if !IDSP_HBF_BLK_SPREAD
        if (!last) request(c + 1, req_fast);
#endif
        IDSP_PH(4)
#ifdef IDSP_EXP_HBF_NOSTAGES
        if (pc[M0].x == 12345.678f) y[0] = pc[0].y;
#else
        v2f y0[kOwn];
        stage0_own<L>(pc, y0, [&](auto k_) {
#if IDSP_HBF_BLK_SPREAD
            if (!last) {
#pragma unroll
                for (int k = 0; k < RQ / 4; k++) request_k(decltype(k_)::value * (RQ / 4) + k, c + 1, req_fast);
            }
#endif
        });
        cs.round(y0, n, last, [&](auto &ov, int nv) {
#pragma unroll
            for (int p = 0; p < PL; p++) tile[(PL * lid + p) * kFmTilePitch + w] = ov[p];
            tile_nv = nv, tile_base = out_done;
            out_done += size_t(nv);
        });
#endif
    }
    lds_barrier();  // every wave's column of the last tile
    if (tile_nv > 0) store_tile();
    {
        constexpr int He = L::He(0), Ho = L::Ho(0);
        const int n = int((frames - (rounds - 1) * kFmRows) * R);
        if (lid < He) st[size_t(lid) * lanes + lane] = __float_as_uint(rows[raw_word(n + 2 * (lid - He))]);
        if (lid < Ho) st[size_t(He + lid) * lanes + lane] = __float_as_uint(rows[raw_word(n + 2 * (lid - Ho) + 1)]);
    }
    cs.store_state(st, lanes, lane);
#ifdef IDSP_EXP_HBF_PHASES
    IDSP_PH(5)
    if (lid == 0 && (w == 0 || w == NL / 2 - 1 || w == NL - 1) && (group == 0 || group + 1 == ngroups))
        printf("phases fm group %d wave %d: wait %lld  barrier A %lld  tile + pieces %lld  barrier B %lld  requests %lld  stages %lld ticks; %lld ticks in %lld x 10 ns\n",
               int(group), w, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], clock64() - ph_clk0, wall_clock64() - ph_wall0);
#endif
}

// -------------------------------------------------------------------------------------------------------- host
#ifndef IDSP_HBF_BLK_PM
#define IDSP_HBF_BLK_PM kPmAll
#endif
#ifndef IDSP_HBF_BLK_RR
#define IDSP_HBF_BLK_RR 1
#endif
#ifndef IDSP_HBF_BLK_PAD  // more bytes of LDS per wave than the kernel uses = fewer waves per CU (LaneMajor)
#define IDSP_HBF_BLK_PAD 0
#endif
#ifndef IDSP_HBF_BLK_FM_LANES
#define IDSP_HBF_BLK_FM_LANES 16
#endif
#ifndef IDSP_HBF_BLK_PM_FM
#define IDSP_HBF_BLK_PM_FM kPmFirst
#endif
template <int TS, int S>
int launch_blk(uint32_t *st, const float *x, float *y, size_t lanes, size_t frames, bool lm, hipStream_t stream)
{
    if (lanes > 0x7fffffffu) return 1;
#ifdef IDSP_HBF_BLK_OFF  // A/B harness: the ring kernels of hbf_ring.h instead
    return 1;
#endif
    if (lm) {
        using L = BLay<TS, S, IDSP_HBF_BLK_PM>;
        // (IDSP_DIAG=1 IDSP_HBF_LDS_PAD=n: n more bytes of LDS per wave — fewer waves per CU, to read the occupancy slope)
        static const size_t pad = [] {
            const char *e = diag_env("IDSP_HBF_LDS_PAD");
            return e ? size_t(strtoul(e, nullptr, 10)) & ~size_t(15) : size_t(0);
        }();
        constexpr int RR = IDSP_HBF_BLK_RR;
        const size_t bytes = (size_t(4 * L::M(0)) + RR * kSC + up4(L::words)) * sizeof(float) + pad + IDSP_HBF_BLK_PAD;
        if (ensure_dyn_lds<&hbf_dec_blk_lm<L, RR>>(bytes)) return 2;
        note_kernel("hbf_dec_blk[LaneMajor]", typeid(L).name());
        hipLaunchKernelGGL((hbf_dec_blk_lm<L, RR>), dim3(unsigned(lanes)), dim3(kW), bytes, stream, st, x, y, lanes, frames);
        return 0;
    }
    if constexpr (S == 4) {
        using L = BLay<TS, S, IDSP_HBF_BLK_PM_FM>;
        if constexpr (L::M(0) <= 4) {
            constexpr int NL = IDSP_HBF_BLK_FM_LANES;
            if (lanes % NL != 0) return 1;
            constexpr size_t bytes = (size_t(kFmRows + 1) * (NL * 16 + 4) + size_t(NL) * up4(L::words) + size_t(L::L(S - 1)) * (NL + 1)) * sizeof(float);
            static_assert(bytes <= 160 * 1024, "one workgroup per CU");
            if (ensure_dyn_lds<&hbf_dec_blk_fm<L, NL>>(bytes)) return 2;
            const size_t ngroups = lanes / NL;
            note_kernel(NL == 16 ? "hbf_dec_blk[FrameMajor]" : "hbf_dec_blk[FrameMajor, 8 lanes per workgroup]", typeid(L).name());
            hipLaunchKernelGGL((hbf_dec_blk_fm<L, NL>), dim3(unsigned(8 * ((ngroups + 7) / 8))), dim3(NL * kW), bytes, stream, st, x, y, lanes, frames);
            return 0;
        }
    }
    return 1;
}

}  // namespace hbfb

// Returns 0 when a blocked kernel was launched, 1 when the request is not covered (the caller goes on to hbf_ring.h /
// hbf_wave.h), 2 on a HIP error (idsp_last_error() holds the text).
int hbf_blk_dec(int tap_set, int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames,
                bool lane_major, hipStream_t stream);

}  // namespace idsp
