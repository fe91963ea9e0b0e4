// hbf_blk.h — half-band decimator cascades (HBF_DEC_CASCADE over HBF_TAPS / HBF_TAPS_98, src/hbf.rs:142-192,385-421),
// register-blocked: round 6 successor of the slot-wise LaneMajor ring kernel of round 4 (hbf_ring.h keeps the FrameMajor one).
//
// Why.  tools/ubench_lds_issue.hip (profiles/r06_ubench_lds_issue.txt): LDS and VALU instructions of a CU's waves issue side
// by side, an LDS read costs the CU's one LDS pipe time in proportion to its BYTES (b64 1.5, b128 2.8 cycles per wave
// instruction), and a 4- or 8-byte-misaligned b64 / b128 costs 43.  The ring kernel's round (1024 raw samples of one lane) moved
// about 50 KiB through LDS, most of it re-reads: every thread fetched 20 raw words to produce two stage-0 outputs, and the
// lowest-rate stages read a 2M-word window per single output.  And the launch is POWER-bound (NOTES round 6: the L2 counters give
// the same 1.3-1.4 M cycles per launch for the product, for its requests alone and for the ring kernel, at 1.56-1.66 GHz with
// the arithmetic and 2.18 GHz without; all-zero input makes both kernels equally fast): what buys time is less energy per
// sample.  Here every thread reads a window ONCE for four or eight neighbouring outputs — 29 KiB of LDS traffic per round:
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
//    both the producer's vector writes and the window reads are aligned.
//  * arithmetic as in hbf_ring.h: symmetric sums scalar, tap multiplies and the sequential accumulation packed over pairs
//    of neighbouring outputs; every IEEE operation and its order per output unchanged (0 ULP against the oracle).
//  * the last stage's outputs leave as 16 bytes per thread, one contiguous KiB per wave and run.
//
// One code path with run-time counts: rounds past the regular ones (the last two: requests reach the end of the row, the
// last round may be short, partially filled runs are flushed) differ only in the request form (clamped addresses), the
// predicates of the output stores and the roll distances — all wave-uniform scalars.
//
// Measured and dropped (profiles/r06_exp_hbf_ab_*.jsonl, all bit-exact, A/B in one process on the same buffers): a ring of two
// rounds (two rounds of lead: 0.886 against 0.873 ms), 8 .. 15 waves per CU (flat within 1.5 %), the four requests spread over
// stage 0 (LaneMajor: no change), nontemporal output stores (no change), stages 2 / 3 at two outputs per thread (no change);
// and a FrameMajor form (16 or 2 x 8 lanes per workgroup, two barriers per round instead of four): equal to the ring kernel
// at best (0.94-0.99 against 0.95-0.96 ms) — its sixteen waves run the stages in lock step at 5.5 cycles per instruction where
// the LaneMajor waves, free of barriers, reach 2.9 — and removed again (git: 4125bd5).
#pragma once

#include "hbf_ring.h"

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
// of 128 pairs: half the stream space).  The product runs all stages at four.
constexpr int kPmAll = 0x3e;
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
template <class L>
__device__ __forceinline__ void stage0_own(const v4f *pc, v2f (&y0)[kOwn])
{
    static_for<0, kOwn>([&](auto q_) {
        constexpr int q = decltype(q_)::value;
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
template <class L>
__global__ __launch_bounds__(kW) void hbf_dec_blk_lm(uint32_t *st, const float *x, float *y, const size_t lanes, const size_t frames)
{
    extern __shared__ __attribute__((aligned(16))) float smem_blk[];
    constexpr int S = L::stages, R = L::rate, M0 = L::M(0);
    constexpr int HIST = 4 * M0;  // words
    constexpr int NP = M0 + kOwn;
    const int lid = threadIdx.x;
    const size_t lane = blockIdx.x;
    float *const hist = smem_blk, *const ring = smem_blk + HIST, *const str = ring + kSC;

    // stage-0 history (even[M0-1] then odd[2 M0-1], oldest first) -> the raw positions -1, -2, ... of the history slots
    {
        constexpr int He = L::He(0), Ho = L::Ho(0);
        if (lid < He) hist[HIST + 2 * (lid - He)] = __uint_as_float(st[size_t(lid) * lanes + lane]);
        if (lid < Ho) hist[HIST + 2 * (lid - Ho) + 1] = __uint_as_float(st[size_t(He + lid) * lanes + lane]);
    }
    Cascade<L> cs;
    cs.init(str, lid, st, lanes, lane);

    // word offsets (from `hist`) of the pieces 4 t - M0 .. 4 t + 3 of a round
    int pa[NP];
#pragma unroll
    for (int h = 0; h < NP; h++) {
        const int g = 4 * lid - M0 + h;
        pa[h] = g >= 0 ? HIST + 4 * ring_pos(g) : 4 * (g + M0);
    }

    const size_t total = frames * size_t(R);  // raw samples of the lane
    const size_t npieces = total / 4;          // whole 16-byte pieces (the dispatcher guarantees total % 4 == 0)
    const float *xl = x + lane * total;
    float *yl = y + lane * frames;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)ring;
    const int perm = 4 * (lid & 15) + (lid >> 4);  // the piece of a KiB this lane requests
    const uint32_t voff = uint32_t(perm) * 16;
    const size_t rounds = (total + kSC - 1) / kSC;

    // requests of round c; `fast`: the whole round lies inside the row
    auto request = [&](size_t c, bool fast) {
#ifndef IDSP_EXP_HBF_NOLOAD
        if (fast) {
            const float *xc = uniform_ptr(xl + c * kSC);
            static_for<0, 4>([&](auto k_) {
                constexpr int k = decltype(k_)::value;
                // the instruction offset moves the global AND the LDS address (lds_dma.h)
                glds16_si<k * 1024>(xc, voff, ring_lds);
            });
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const size_t pc = c * (kSC / 4) + size_t(64 * k + perm);
                glds16(xl + (pc < npieces ? pc * 4 : 0), ring_lds + uint32_t(k) * 1024);
            }
        }
#endif
    };

    size_t out_done = 0;  // outputs of the lane stored so far
    int pending = 0;      // vector-memory operations issued after the latest requests

#ifdef IDSP_EXP_HBF_PHASES  // where one wave's time goes: wait for data / piece reads / request issue / arithmetic (shader cycles)
    long long ph[4] = {0, 0, 0, 0}, tp = clock64();
    const long long ph_wall0 = wall_clock64(), ph_clk0 = tp;
#define IDSP_PH(i) { const long long t_ = clock64(); ph[i] += t_ - tp; tp = t_; }
#else
#define IDSP_PH(i)
#endif
    request(0, kSC <= total);
    for (size_t c = 0; c < rounds; c++) {
        const bool last = c + 1 == rounds;
        const int n = last ? int(total - c * kSC) : kSC;  // raw samples of this round
        IDSP_PH(3)
        // "my four requests have landed": they retire in issue order with the stores behind them
        if (pending == 0)
            wait_vmcnt<0>();
        else if (pending == 1)
            wait_vmcnt<1>();
        else
            wait_vmcnt<2>();
        IDSP_PH(0)
        v4f pc[NP];
#pragma unroll
        for (int h = 0; h < NP; h++) pc[h] = *reinterpret_cast<const v4f *>(hist + pa[h]);
        lds_wave_sync();
        IDSP_PH(1)
        if (!last) {
            // the round's last M0 pieces are the next round's history
#pragma unroll
            for (int i = 0; i < kOwn; i++) {
                const int g = 4 * lid + i - (kSC / 4 - M0);
                if (g >= 0) *reinterpret_cast<v4f *>(hist + 4 * g) = pc[M0 + i];
            }
            request(c + 1, (c + 2) * kSC <= total);  // into the ring this round's pieces have just left
            pending = 0;
        }
        IDSP_PH(2)
#ifdef IDSP_EXP_HBF_NOSTAGES
        if (pc[M0].x == 12345.678f) yl[0] = pc[0].y;
#else
        v2f y0[kOwn];
        stage0_own<L>(pc, y0);
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
            pending += 2;
        } else {
            cs.round(y0, n, last, [&](auto &ov, int nv) {
                constexpr int P = L::P(S - 1);
                const int i0 = P * lid;
                float *dst = yl + out_done + i0;
                out_done += size_t(nv);
                pending += 1;
#ifdef IDSP_EXP_HBF_NOSTORE  // timing only: the outputs stay in registers
                if (ov[0] != 12345.678f) nv = 0;
#endif
                if (i0 + P - 1 < nv) {
                    if constexpr (P == 4)
                        *reinterpret_cast<v4f *>(dst) = v4f{ov[0], ov[1], ov[2], ov[3]};
                    else
                        *reinterpret_cast<v2f *>(dst) = v2f{ov[0], ov[1]};
                } else {
#pragma unroll
                    for (int p = 0; p < P; p++)
                        if (i0 + p < nv) dst[p] = ov[p];
                }
            });
        }
#endif
    }
    lds_wave_sync();
    // state: the last raw samples of the row (stage 0) and the stream histories (stages >= 1)
    {
        constexpr int He = L::He(0), Ho = L::Ho(0);
        const int n = int(total - (rounds - 1) * kSC);
        auto raw_word = [&](int r) { return r >= 0 ? HIST + 4 * ring_pos(r >> 2) + (r & 3) : HIST + r; };
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

// -------------------------------------------------------------------------------------------------------- host
#ifndef IDSP_HBF_BLK_PM  // experiment knob: which stages run at four outputs per thread (tools/exp_hbf_blk.sh)
#define IDSP_HBF_BLK_PM kPmAll
#endif
template <int TS, int S>
int launch_blk(uint32_t *st, const float *x, float *y, size_t lanes, size_t frames, bool lm, hipStream_t stream)
{
    if (!lm) return 1;                  // FrameMajor: hbf_ring.h
    if (lanes > 0x7fffffffu) return 1;  // one workgroup per lane in a 31-bit grid: not covered beyond, as in cic_ring_host.h
#ifdef IDSP_HBF_BLK_OFF  // A/B harness (tools/exp_hbf_ab.py): the wave kernels of hbf_wave.h instead
    return 1;
#endif
    using L = BLay<TS, S, IDSP_HBF_BLK_PM>;
    // (IDSP_DIAG=1 IDSP_HBF_LDS_PAD=n: n more bytes of LDS per wave — fewer waves per CU, to read the occupancy slope)
    static const size_t pad = [] {
        const char *e = diag_env("IDSP_HBF_LDS_PAD");
        return e ? size_t(strtoul(e, nullptr, 10)) & ~size_t(15) : size_t(0);
    }();
    const size_t bytes = (size_t(4 * L::M(0)) + kSC + up4(L::words)) * sizeof(float) + pad;
    if (ensure_dyn_lds<&hbf_dec_blk_lm<L>>(bytes)) return 2;
    note_kernel("hbf_dec_blk[LaneMajor]", typeid(L).name());
    hipLaunchKernelGGL((hbf_dec_blk_lm<L>), dim3(unsigned(lanes)), dim3(kW), bytes, stream, st, x, y, lanes, frames);
    return 0;
}

}  // namespace hbfb

// Returns 0 when the blocked kernel was launched, 1 when the request is not covered (the caller goes on to hbf_ring.h /
// hbf_wave.h), 2 on a HIP error (idsp_last_error() holds the text).
int hbf_blk_dec(int tap_set, int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames,
                bool lane_major, hipStream_t stream);

}  // namespace idsp
