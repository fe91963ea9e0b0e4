// fm_sweep.h — FrameMajor per-lane recurrences as ONE dense sweep over memory, for any lane count (round 5).
//
// Replaces the same triple loop as lane_stream.h (dsp-process/src/process.rs:122-141 driven by `Lanes`,
// dsp-process/src/compose.rs:468-494: any N in `Lanes<C>`).
//
// What was measured (profiles/NOTES.md, round 5).  The LDS-DMA kernel of lane_stream.h runs at 0.78 of the HBM peak when
// its 256 workgroups together cover every row of the tensor as one dense piece (C2: 65536 lanes, 256 KiB rows) — at a
// 2 GiB and at a 32 GiB footprint alike — and at 0.60-0.70 whenever a round of workgroups covers only a PANEL of every
// row (more than 65536 lanes on a persistent grid: 256 KiB pieces at a 4 MiB stride for C5), whatever the kernel's inner
// structure is (role waves with decoupled loads and stores ran the same rates).  The memory system wants the launch as
// a whole to move front to back through the tensor.  So here the lane count never changes the shape of the walk:
//
//   * always ONE round of G <= 256 co-resident workgroups (one per CU); a workgroup owns LPT sub-blocks of `bw` lanes
//     each, INTERLEAVED over the row: sub-block s of workgroup w = lanes [(s G + w) bw, (s G + w + 1) bw).  Consecutive
//     blocks of a row belong to consecutive workgroups (the 8 XCDs in turn), every row is requested as one dense piece,
//     and the launch sweeps x and y front to back exactly as the single-round C2 launch does.  Thread t runs the LPT
//     independent recurrences of lanes t of its sub-blocks (LPT = 1, 2, 4, 8, 16 — state registers x LPT);
//   * `bw` <= 256 lanes (a multiple of 16: blocks start on 64-byte boundaries of a 64-byte aligned row) is chosen by the
//     host so that G LPT bw covers the lanes with less than one workgroup's worth of waste at G close to 256
//     (sweep_geometry below): 65536 lanes -> 256 x 1 x 256; 100000 -> 241 x 2 x 208; 2^20 -> 256 x 16 x 256.  Pieces
//     (16 bytes = 4 lanes) beyond a block's `bw` lanes are masked out of the requests and stores (partial EXEC — no
//     duplicate traffic); a sub-block that lies wholly beyond the last lane (only the last sub-block of the last few
//     workgroups can) runs as a CLONE of the workgroup's sub-block 0 — same requests, same state, same results stored to
//     the same addresses — so that every wave issues the same number of memory operations per tile and the hand-counted
//     `s_waitcnt vmcnt(N)` stays static;
//   * tiles are always 8 one-KiB segments (ring of NB = 7: 56 KiB of requests in flight per CU, the depth C2 was tuned
//     to, whatever LPT is): LPT <= 8 -> 8 / LPT frames x LPT sub-blocks, LPT = 16 -> half a frame (the main loop is
//     unrolled over the tiles of a frame so that the state registers are indexed statically);
//   * more lanes than 256 x 16 x 256: several such sweeps one after the other inside the launch (`rounds`), each over a
//     contiguous range of lanes.
//
// Everything else (LDS-DMA ring, output staging in LDS, nontemporal 16-byte stores, hand-counted vmcnt, in-place safe
// because a workgroup's requests run ahead of its own stores) is stream_frame_major_lds's.
#pragma once

#include "lane_stream.h"

namespace idsp {

constexpr int kSweepNB = 7;    // ring depth in 8 KiB tiles
constexpr int kSweepT = 8;     // segments per tile
// Schedules of the shipped instantiations (FORM bits and SLP: see the kernel; tools/exp_fm_roles.hip measures all of them,
// profiles/r05_exp_fm_sweep_*.jsonl).  FULL 256-lane blocks are bound by memory, and at that ceiling what matters is how a wave's two requests
// and two stores per tile are spread over the tile period: all 1024 waves of the launch are in step, and without pacing they all
// issue right behind the barrier, then nothing for a thousand cycles.  A short `s_sleep` before each request (the cheap processors
// have the slack) makes the rate the same for every placement of y and for y == x.  NARROW blocks (lane counts that are no multiple of
// 65536 LPT) leave the memory system slack, the serial skeleton shows, and the form that reads the whole tile into registers first, with
// one barrier per tile, is 15-40 % faster there than the read-step-write forms (100000 lanes 0.70-0.72 against 0.66, 147456 0.67 against
// 0.48, 49152 0.73 against 0.67).
// One schedule for everything — registers first, one barrier per tile (FORM 3) — and for the cheap processors' full blocks `s_sleep 4` before each
// request (SLP 4, a compile-time constant: the same sleep behind a run-time switch measured 0.754 where this runs 0.800 — these kernels sit on a
// timing edge).  Over five boxes: 65536 lanes 0.78-0.80, flat over the placement of y and in place, where the unpaced forms range 0.71-0.79 by box
// and 0.60-0.79 by placement; 2^20 lanes 0.735-0.74 flat (two-barrier unpaced: 0.72-0.75).  CHEAP = COST <= 50: i32 DF1 (with its tile form) and the
// f32 sections; the dither / wide / multi-section bodies have no slack to sleep in.
constexpr int kSweepForm = 3, kSweepPace = 4;
template <class P, class = void>
struct SweepUnpacedOf : std::false_type {};
template <class P>
struct SweepUnpacedOf<P, std::void_t<decltype(P::SWEEP_UNPACED)>> : std::integral_constant<bool, P::SWEEP_UNPACED> {};
// processors whose full blocks are bound by memory alone and take the paced schedule (and, on rows off the grid, the split requests): little
// arithmetic per sample and not declared SWEEP_UNPACED (the clamped sections, `Normal`: biquad_sections.h)
template <class P>
constexpr bool sweep_cheap() { return P::COST <= 50 && !SweepUnpacedOf<P>::value; }
// processors whose full blocks run 8 / 16 blocks per workgroup on the two-barrier schedule (launch_sweep_lpt below)
template <class P, class = void>
struct SweepBigTwoBarrierOf : std::false_type {};
template <class P>
struct SweepBigTwoBarrierOf<P, std::void_t<decltype(P::SWEEP_BIG_TWO_BARRIER)>> : std::integral_constant<bool, P::SWEEP_BIG_TWO_BARRIER> {};

struct SweepGeom {
    int lpt = 1;             // sub-blocks per workgroup
    unsigned grid = 0;       // workgroups
    unsigned bw = 256;       // lanes per sub-block (multiple of 16, <= 256)
    unsigned rounds = 1;     // sweeps inside the launch
    unsigned fps = 1;        // frames per one-KiB segment (LPT = 1 and bw <= 128: launch_sweep sets it for 4-byte outputs)
    size_t round_lanes = 0;  // lanes per sweep (the last one may hold fewer)
};

// Geometry for `lanes` lanes (a multiple of 4) with at most `max_lpt` sub-blocks per workgroup; false: not coverable
// (fewer than 16 lanes).  Waste (lanes of capacity beyond the last lane) costs duplicate traffic of at most one
// workgroup; a grid below 256 leaves CUs idle — scored 4 : 1 (tools/exp_fm_sweep.hip sweeps the alternatives).
inline bool sweep_geometry(size_t lanes, int max_lpt, SweepGeom &out, unsigned max_grid = 256)
{
    if (lanes < 16 || max_lpt < 1) return false;
    const size_t cap = size_t(max_grid) * size_t(max_lpt) * 256;
    const size_t rounds = (lanes + cap - 1) / cap;
    size_t lr = (lanes + rounds - 1) / rounds;
    lr = (lr + 15) / 16 * 16;
    int lpt = 1;
    while (lpt < max_lpt && size_t(max_grid) * size_t(lpt) * 256 < lr) lpt *= 2;
    double best = 1e30;
    for (unsigned bw = 256; bw >= 16; bw -= 16) {
        const size_t per = size_t(lpt) * bw, g = (lr + per - 1) / per;
        if (g > max_grid) continue;
        const double waste = double(g * per - lr) / double(lr), idle = double(max_grid - g) / double(max_grid);
        const double score = waste + 0.25 * idle + (256 - bw) * 1e-5;
        if (score < best) best = score, out.bw = bw, out.grid = unsigned(g);
    }
    if (best >= 1e30) return false;
    out.lpt = lpt, out.rounds = unsigned(rounds), out.round_lanes = lr;
    return true;
}

template <class P>
constexpr size_t sweep_lds_bytes(int nb = kSweepNB, int ts = kSweepT)
{
    return (size_t(nb) * ts * kFmBlock + 2 * size_t(ts) * kFmBlock * (sizeof(typename P::Out) / 4) + P::LDS_WORDS) * 4;
}

// FORM bit 0: the tile's samples go to registers before the first step (else read - step - write per sample);
// FORM bit 1: ONE workgroup barrier per tile (below) instead of two;
// FORM bit 2 (with bit 1): the wave's second request and second store go out half way through the tile's arithmetic instead of
// right behind the first ones — the chip's requests then come in two waves per tile period instead of one burst;
// SLP: `s_sleep SLP` (64 cycles each) before every request (pacing experiment).
// TSEG: one-KiB segments per tile (8 or 16: twice the steps per barrier pair, for the shapes where the serial skeleton shows).
// FPSM: several frames per segment (`fps_` of them; LPT = 1 and 4-byte outputs only).  A template flag, not just the run-time count: with the slot
// loop in it the one-frame-per-segment kernel lost 5-30 % at 65536 lanes (0.34 -> 0.36-0.49 ms by schedule).
// XC (full 256-lane blocks on rows off the 64-byte grid): every request goes out as TWO instructions, lanes 0-59 and lanes 60-63.  Same lines, same
// bytes — but a 1 KiB request that starts off the grid ran 0.59-0.65 of the peak in the copy skeleton where 60 + 4 lanes run 0.68-0.72 and 960-byte
// segments 0.74-0.76 (tools/ubench_fm_rows_off_grid.hip, NOTES round 5).  The second instruction always has its four lanes on (a lane beyond a partial
// block's pieces re-requests the block's first piece into its own, unused, part of the segment): the request count stays static.
template <class P, int LPT, int NB = kSweepNB, int FORM = 3, int SLP = 0, int TSEG = kSweepT, bool FPSM = false, bool XC = false>
__global__ __launch_bounds__(kFmBlock) void stream_frame_major_sweep(
    const typename P::Params prm, uint32_t *st, const typename P::In *x, typename P::Out *y,
    const size_t lanes, const size_t frames, const size_t xl, const size_t yl, const size_t slanes,
    const unsigned bw, const unsigned rounds, const size_t round_lanes, const unsigned xcdc, const unsigned fps_)
{
    using In = typename P::In;
    using Out = typename P::Out;
    static_assert(P::HAS_IN && P::IN_DIV == 1 && sizeof(In) == 4, "sweep kernel: one 4-byte input per lane and frame");
    static_assert(LPT >= 1 && (LPT & (LPT - 1)) == 0, "LPT is a power of two");
    constexpr int OW = sizeof(Out) / 4, B = BatchOf<P>::value, TS = TSEG;
    constexpr int SB = LPT < TS ? LPT : TS;  // sub-blocks per tile
    constexpr int R = TS / SB;               // frames per tile
    constexpr int PH = LPT / SB;             // tiles per frame group (LPT = 16: 2)
    constexpr int RPW = TS / 4;              // segments per wave and tile
    constexpr bool PRE = (FORM & 1) != 0, ONEBAR = (FORM & 2) != 0, SPLIT = (FORM & 4) != 0;
    static_assert(!SPLIT || (ONEBAR && RPW == 2), "split requests: one-barrier schedule");
    static_assert(TS == 8 || TS == 16, "tile of 8 or 16 segments");
    constexpr int RQ = XC ? 2 : 1;           // request instructions per segment
    static_assert(!XC || (!FPSM && !SPLIT), "split requests: one frame per segment, unsplit schedule");
    constexpr int kYoungS = OW + (NB - 2) * (RPW + RPW * OW);  // split schedule: request, stores, request, stores per iteration
    constexpr int kYoung = RPW * OW + (NB - 1) * (RPW * RQ + RPW * OW);
    constexpr int kYoung1 = RPW * OW + (NB - 2) * (RPW * RQ + RPW * OW);  // one-barrier schedule
    static_assert(kYoung <= 63, "vmcnt range");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *tin = smem;                             // [NB][TS][256]
    uint32_t *tout = smem + NB * TS * kFmBlock;       // [2][TS][256 * OW]
    uint32_t *ptab = tout + 2 * TS * kFmBlock * OW;   // [P::LDS_WORDS]
    const int tid = threadIdx.x, lid = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)smem;
    // xcdc (rows off the 64-byte grid — dense rows of a lane count that is no multiple of 16, odd pitches, a base pointer off the
    // grid): neighbouring blocks then share the 64-byte pieces at their boundaries, and with consecutive blocks dealt to the 8 XCDs
    // in turn both halves of every shared piece go through two different L2s (round 3: 0.58-0.64 against 0.77).  Workgroup
    // blockIdx runs on XCD blockIdx % 8: give XCD j the j-th contiguous eighth of the block positions, so that neighbours meet in
    // one L2.  (The rows are still swept densely: the eight eighths advance together.)
    const size_t G = gridDim.x;
    size_t w = blockIdx.x;
    if (xcdc) {
        const size_t q = G / 8, r = G % 8, j = blockIdx.x % 8;
        w = j * q + (j < r ? j : r) + blockIdx.x / 8;
    }

    P p[LPT];
    if constexpr (P::LDS_WORDS > 0) {
        P::fill_shared(ptab, tid, kFmBlock);  // published by the first tile barrier
#pragma unroll
        for (int s = 0; s < LPT; s++) p[s].set_shared(ptab);
    }

    // tile position: index, ring slot (index % NB), phase (index % PH), first frame of its frame group and that frame's row offset
    // — kept as running values (a `% 7` per use is twenty scalar instructions on a wave that has nothing to hide them behind)
    struct Pos {
        size_t i;
        int slot, ph;
        size_t fr, row;
    };
    // fps (LPT = 1, 4-byte outputs, few lanes): SEVERAL frames share a one-KiB segment — slot k of a segment holds the block's `bw` lanes of
    // frame (first frame of the tile) + k TS + (segment index), fps = 256 / bw slots — so that a tile is 8 KiB of real samples however few
    // lanes a workgroup owns (16384 lanes: 64 per workgroup, 32 frames per tile) instead of 8 KiB of which three quarters are masked.
    // Each thread's piece of a segment then lies in slot (4 lid) / bw at lane (4 lid) % bw: a per-thread address offset, constant for
    // the launch; threads 0 .. bw - 1 run the tile's 8 fps steps of their lane, slot after slot.
    static_assert(!FPSM || (LPT == 1 && OW == 1), "several frames per segment: one block per workgroup, 4-byte outputs");
    const unsigned fps = FPSM && fps_ > 1 ? fps_ : 1u;
    const unsigned fpt = unsigned(R) * fps;  // frames per tile (frame group)
    const unsigned tslot = fps > 1 ? unsigned(lid * 4) / bw : 0u, tlane = fps > 1 ? unsigned(lid * 4) % bw : unsigned(lid * 4);
    const bool tgeo = tslot < fps;
    const uint32_t xvoff = uint32_t((size_t(tslot) * TS * xl + tlane) * sizeof(In));  // launcher: fits 32 bits
    const size_t yword = fps > 1 ? size_t(tslot) * TS * yl + tlane : size_t(lid * 4);  // words (OW = 1 when fps > 1)
    const bool comp_on = fps == 1 || unsigned(tid) < bw;
    auto next_pos = [&](Pos &q, size_t pitch) {
        q.i++;
        q.slot = q.slot + 1 == NB ? 0 : q.slot + 1;
        if (PH == 1 || ++q.ph == PH) q.ph = 0, q.fr += fpt, q.row += size_t(fpt) * pitch;
    };
    const size_t ypitch = yl * OW;  // words between the frames of y

    for (unsigned rd = 0; rd < rounds; rd++) {
        const size_t rl0 = size_t(rd) * round_lanes;                                    // first lane of this sweep
        const size_t rlanes = lanes - rl0 < round_lanes ? lanes - rl0 : round_lanes;     // its lanes
        if (w * bw >= rlanes) break;  // not even sub-block 0 exists (only beyond the data: later sweeps hold no more lanes than this one)
        // sub-block s: first lane (relative to the sweep) and number of lanes that exist; absent -> clone of sub-block 0
        auto sub_first = [&](int s) -> size_t {
            const size_t f = (size_t(s) * G + w) * bw;
            return f < rlanes ? f : w * bw;
        };
        auto sub_count = [&](int s) -> unsigned {
            size_t f = (size_t(s) * G + w) * bw;
            if (f >= rlanes) f = w * bw;
            return rlanes - f < bw ? unsigned(rlanes - f) : bw;
        };
        auto sub_present = [&](int s) { return (size_t(s) * G + w) * bw < rlanes; };

#pragma unroll
        for (int s = 0; s < LPT; s++) {
            const size_t f = sub_first(s);
            p[s].load(prm, st, slanes, rl0 + f + (unsigned(tid) < sub_count(s) ? size_t(tid) : 0));
        }
        // the state loads must have landed HERE, where the compiler's wait-count pass sees it (see stream_frame_major_lds)
        __builtin_amdgcn_s_waitcnt(0x0F70);

        const size_t ngroups = (frames + fpt - 1) / fpt;  // frame groups of R fps frames
        const size_t ntiles = ngroups * PH;
        const size_t nfull = (frames / fpt) * PH;        // tiles [0, nfull) hold whole frame groups
        const In *xr = x + rl0;
        uint32_t *yr = reinterpret_cast<uint32_t *>(y) + rl0 * OW;

        // This wave's segments of a tile: g = wave + 4 j; in phase ph that is frame offset g / SB of the group and sub-block
        // ph SB + g % SB.  Their lane offsets, counts and complete element / word offsets are constants of the sweep.
        int sfo[RPW];
        size_t sfst[RPW][PH], sxo[RPW][PH], syo[RPW][PH];
        unsigned scnt[RPW][PH];
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int g = wave + 4 * j;
            sfo[j] = g / SB;
#pragma unroll
            for (int ph = 0; ph < PH; ph++) {
                const int s = ph * SB + g % SB;
                sfst[j][ph] = sub_first(s), scnt[j][ph] = sub_count(s);
                sxo[j][ph] = size_t(sfo[j]) * xl + sfst[j][ph];
                syo[j][ph] = (size_t(sfo[j]) * yl + sfst[j][ph]) * OW;
            }
        }
        auto pick = [&](auto &arr, int ph) {  // arr[ph] for a phase known at run time only
            auto v = arr[0];
#pragma unroll
            for (int q = 1; q < PH; q++)
                if (ph == q) v = arr[q];
            return v;
        };

        // requests of the tile at position q (phase static in the steady state: PHS >= 0)
        auto issue = [&](const Pos &q, auto phs, auto full, int j0 = 0, int j1 = 64) {
            constexpr bool FULL = decltype(full)::value;
            constexpr int PHS = decltype(phs)::value;
#pragma unroll
            for (int j = 0; j < RPW; j++) {
                if (j < j0 || j >= j1) continue;
                if (SLP) __builtin_amdgcn_s_sleep(SLP);
                const unsigned cnt = PHS >= 0 ? scnt[j][PHS >= 0 ? PHS : 0] : pick(scnt[j], q.ph);
                const In *base;
                if (FULL || q.fr + sfo[j] < frames)
                    base = xr + q.row + (PHS >= 0 ? sxo[j][PHS >= 0 ? PHS : 0] : pick(sxo[j], q.ph));
                else  // ragged tile: re-request the last frame (static request count)
                    base = xr + (frames - 1) * xl + (PHS >= 0 ? sfst[j][PHS >= 0 ? PHS : 0] : pick(sfst[j], q.ph));
                bool on = tgeo && tlane < cnt;
                uint32_t voff = xvoff;
                if constexpr (!FULL) {
                    if (q.fr + sfo[j] >= frames)
                        on = on && tslot == 0, voff = tlane * uint32_t(sizeof(In));  // the re-requested last frame: slot 0 only
                    else
                        on = on && q.fr + size_t(tslot) * TS + sfo[j] < frames;       // this thread's slot lies beyond the last frame
                }
                if constexpr (XC) {
                    const uint32_t dst = lds_base + uint32_t((q.slot * TS + wave + 4 * j) * kFmBlock * 4);
                    if (on && lid < 60) glds16_s(base, voff, dst);
                    if (lid >= 60) glds16_s(base, on ? voff : 0u, dst);
                } else if (on)
                    glds16_s(base, voff, lds_base + uint32_t((q.slot * TS + wave + 4 * j) * kFmBlock * 4));
            }
        };
        // stores of the tile at position q (q.row counts words of y)
        auto store = [&](const Pos &q, auto phs, auto full, int j0 = 0, int j1 = 64) {
            constexpr bool FULL = decltype(full)::value;
            constexpr int PHS = decltype(phs)::value;
            const uint32_t *o = tout + (q.i & 1) * TS * kFmBlock * OW;
#pragma unroll
            for (int j = 0; j < RPW; j++) {
                if (j < j0 || j >= j1) continue;
                const int g = wave + 4 * j;
                const bool row_ok = FULL || q.fr + sfo[j] < frames;
                const unsigned nv = PHS >= 0 ? scnt[j][PHS >= 0 ? PHS : 0] : pick(scnt[j], q.ph);
                int gsrc = g, lds0 = 0;  // segment (and first word inside it) of the output tile this wave-instruction reads
                uint32_t *base;
                if (row_ok) {
                    base = yr + q.row + (PHS >= 0 ? syo[j][PHS >= 0 ? PHS : 0] : pick(syo[j], q.ph));
                } else {  // a row beyond the data: the tile's last real row instead (fps > 1: frame k of the tile sits in slot k / TS of segment k % TS)
                    const size_t lastf = frames - 1 - q.fr;
                    gsrc = fps > 1 ? int(lastf % TS) : int(lastf) * SB + g % SB;
                    lds0 = fps > 1 ? int(lastf / TS) * int(bw) : 0;
                    base = yr + ((frames - 1) * yl + (PHS >= 0 ? sfst[j][PHS >= 0 ? PHS : 0] : pick(sfst[j], q.ph))) * OW;
                }
#pragma unroll
                for (int h = 0; h < OW; h++) {
                    // piece (h, lid) holds lanes (h 256 + lid 4) / OW ... of the sub-block
                    const unsigned first = fps > 1 ? tlane : unsigned(h * kFmBlock + lid * 4) / OW;
                    bool on = tgeo && first < nv && row_ok && (FULL || q.fr + size_t(tslot) * TS + sfo[j] < frames);
                    int word = h * kFmBlock + lid * 4;   // in the LDS tile
                    size_t gword = fps > 1 ? yword : size_t(word);  // in memory, from `base`
                    if (!row_ok || unsigned(h * kFmBlock) / OW >= nv) {
                        // no piece of this wave-instruction exists (a row beyond the data, or OW = 2 and nv <= 128): thread 0 stores the first
                        // piece of the sub-block's last real row once more (same bytes as its owner stores), so that the instruction is
                        // issued and the hand-counted vmcnt stays exact
                        on = lid == 0;
                        word = 0, gword = 0;
                    }
                    if (on) {
                        const u32x4 v4 = *reinterpret_cast<const u32x4 *>(o + gsrc * OW * kFmBlock + (row_ok ? 0 : lds0) + word);
                        __builtin_nontemporal_store(v4, reinterpret_cast<u32x4 *>(base + gword));
                    }
                }
            }
        };
        auto compute = [&](const Pos &q, auto phase, auto full, auto &&mid) {  // mid(): called half way through the tile's steps
            constexpr bool FULL = decltype(full)::value;
            constexpr int ph = decltype(phase)::value;
            const uint32_t *in = tin + q.slot * TS * kFmBlock;
            uint32_t *o = tout + (q.i & 1) * TS * kFmBlock * OW;
            if (!comp_on) return;  // (several frames per segment) threads beyond the block's lanes own no column
            const size_t left = frames - q.fr;  // frames from this tile's first one on
            // (requesting the NEXT slot's samples before this slot's arithmetic was measured slower: 32768 lanes 0.195 -> 0.23 ms)
            for (unsigned jj = 0; jj < (FPSM ? fps : 1u); jj++) {
            const unsigned joff = jj * bw * (fps > 1);  // this slot's lanes inside a segment
            const size_t f0 = size_t(jj) * R;
            const int nr = FULL || left >= f0 + size_t(R) ? R : left > f0 ? int(left - f0) : 0;  // rows of this slot that exist
            if constexpr (!PRE) {
                static_for<TS>([&](auto gg) {
                    constexpr int g = decltype(gg)::value;
                    constexpr int s = ph * SB + g % SB;
                    if (g == TS / 2 && jj == 0) mid();
                    if (FULL || g / SB < nr)
                        to_words<Out>(step1(p[s], prm, __builtin_bit_cast(In, in[g * kFmBlock + joff + tid])), o + (g * kFmBlock + joff + tid) * OW);
                });
                continue;
            }
            // PRE: the tile's samples of this thread go to registers FIRST, all TS of them, and the results leave after the last step: input
            // ring and output tile are one LDS array to the compiler, so a `read, step, write` loop keeps every read behind the previous
            // step's write and pays the LDS latency once per step.
            In v[TS];
            Out r[TS];
#pragma unroll
            for (int g = 0; g < TS; g++) v[g] = __builtin_bit_cast(In, in[g * kFmBlock + joff + tid]);
            if constexpr (B > 1 && LPT == 1) {  // segments are consecutive frames of one lane
#pragma unroll
                for (int r0 = 0; r0 < TS; r0 += B) {
                    if (!FULL && r0 >= nr) break;
                    typename P::Pre pre[B];
#pragma unroll
                    for (int b = 0; b < B; b++)
                        if (FULL || r0 + b < nr) pre[b] = p[0].pre(prm);
#pragma unroll
                    for (int b = 0; b < B; b++) {
                        const int k = r0 + b;
                        if (FULL || k < nr) r[k] = p[0].step(prm, v[k], pre[b]);
                    }
                    if ((r0 + B == TS / 2 || (B > TS / 2 && r0 == 0)) && jj == 0) mid();
                }
            } else if constexpr (FULL && B == 1 && HasTileOf<P>::value && !SPLIT) {
                tile_of<P, SB, R>(prm, &p[ph * SB], v, r);  // the tile's feed-forward terms ahead of the per-sample chains
            } else {
                static_for<TS>([&](auto gg) {
                    constexpr int g = decltype(gg)::value;  // static: the LPT states stay in registers
                    constexpr int s = ph * SB + g % SB;
                    if (g == TS / 2 && jj == 0) mid();
                    if (FULL || g / SB < nr) r[g] = step1(p[s], prm, v[g]);
                });
            }
#pragma unroll
            for (int g = 0; g < TS; g++)
                if (FULL || g / SB < nr) to_words<Out>(r[g], o + (g * kFmBlock + joff + tid) * OW);
            }
        };
        auto compute_dyn = [&](const Pos &q, auto full) {  // tile phase known at run time only (start-up and drain)
            static_for<PH>([&](auto pp) {
                if (q.ph == decltype(pp)::value) compute(q, pp, full, [] {});
            });
        };

        using Full = std::true_type;
        using Ragged = std::false_type;
        using Dyn = std::integral_constant<int, -1>;
        Pos qi{0, 0, 0, 0, 0}, qc{0, 0, 0, 0, 0};  // next tile to request (rows in elements of x), tile to compute (rows in words of y)
        if constexpr (ONEBAR) {
            // One barrier per tile.  After barrier i every wave's share of tile i has landed (each waited for its own requests) AND
            // every wave is done with tile i - 1: its input slot is free for tile i + NB - 1 and its output tile is complete.  So the
            // request, the stores of tile i - 1 and the LDS reads of tile i all go out together, and the arithmetic follows.
            Pos qs = qc;  // tile to store: one behind qc
            for (; qi.i + 1 < size_t(NB) && qi.i < ntiles; next_pos(qi, xl)) issue(qi, Dyn{}, Ragged{});
            auto slow_iter = [&]() {
                wait_vmcnt<0>();
                lds_barrier();
                if (qi.i < ntiles) issue(qi, Dyn{}, Ragged{}), next_pos(qi, xl);
                if (qc.i >= 1) store(qs, Dyn{}, Ragged{}), next_pos(qs, ypitch);
                compute_dyn(qc, Ragged{});
                next_pos(qc, ypitch);
            };
            while (qc.i < ntiles && (qc.i < size_t(NB) || qc.ph != 0)) slow_iter();
            // steady state: younger than tile ii's requests are the stores of tile ii - NB and the requests and stores of the NB - 2
            // iterations since
            while (qc.i + NB - 1 + PH <= nfull) {
                static_for<PH>([&](auto pp) {
                    constexpr int ph = decltype(pp)::value;
                    using PhI = std::integral_constant<int, (ph + NB - 1) % PH>;
                    using PhS = std::integral_constant<int, (ph + PH - 1) % PH>;
                    if constexpr (SPLIT) {
                        wait_vmcnt<kYoungS>();
                        lds_barrier();
                        issue(qi, PhI{}, Full{}, 0, 1);
                        store(qs, PhS{}, Full{}, 0, 1);
                        compute(qc, pp, Full{}, [&] {
                            issue(qi, PhI{}, Full{}, 1, 2);
                            store(qs, PhS{}, Full{}, 1, 2);
                        });
                    } else {
                        wait_vmcnt<kYoung1>();
                        lds_barrier();
                        issue(qi, PhI{}, Full{});
                        store(qs, PhS{}, Full{});
                        compute(qc, pp, Full{}, [] {});
                    }
                    next_pos(qi, xl);
                    next_pos(qs, ypitch);
                    next_pos(qc, ypitch);
                });
            }
            while (qc.i < ntiles) slow_iter();
            lds_barrier();  // the last output tile is complete
            store(qs, Dyn{}, Ragged{});
        } else {
            for (; qi.i < size_t(NB) && qi.i < ntiles; next_pos(qi, xl)) issue(qi, Dyn{}, Ragged{});
            auto slow_iter = [&]() {  // start-up, drain and ragged tiles: wait for everything
                wait_vmcnt<0>();
                lds_barrier();
                compute_dyn(qc, Ragged{});
                lds_barrier();
                if (qi.i < ntiles) issue(qi, Dyn{}, Ragged{}), next_pos(qi, xl);
                store(qc, Dyn{}, Ragged{});
                next_pos(qc, ypitch);
            };
            while (qc.i < ntiles && (qc.i < size_t(NB) || qc.ph != 0)) slow_iter();
            // steady state: every tile involved is full and every wave has issued exactly RPW requests and RPW * OW stores per
            // past tile, so kYoung younger operations may stay in flight
            while (qc.i + NB + PH <= nfull) {
                static_for<PH>([&](auto pp) {
                    constexpr int ph = decltype(pp)::value;
                    wait_vmcnt<kYoung>();
                    lds_barrier();  // all four waves' segments of the tile have landed
                    compute(qc, pp, Full{}, [] {});
                    lds_barrier();  // output tile complete; the tile's ring slot is free again
                    issue(qi, std::integral_constant<int, (ph + NB) % PH>{}, Full{});
                    next_pos(qi, xl);
                    store(qc, pp, Full{});
                    next_pos(qc, ypitch);
                });
            }
            while (qc.i < ntiles) slow_iter();
        }
#pragma unroll
        for (int s = 0; s < LPT; s++)
            if (sub_present(s) && unsigned(tid) < sub_count(s)) p[s].store(prm, st, slanes, rl0 + sub_first(s) + tid);
        // (no barrier before the next sweep: its first lds_barrier() orders this sweep's last output-tile reads before the compute()
        // that overwrites the tile, and its first requests only touch input slots whose last readers passed the final barrier)
    }
}

// ------------------------------------------------------------------------------------------------------------ host
// Largest LPT a processor is instantiated with (each doubling doubles its state registers, the unrolled loop body and the build time): the
// single biquad sections (P::SWEEP_MAX_LPT = 16) take every lane count up to 2^20 in ONE sweep; `Normal`, the per-lane banks and two-section chains
// (8) up to 524288, and 2^20 as two sweeps over half rows; everything else (4, heavy bodies 2) one sweep only — `sweep_takes` below.
template <class P, class = void>
struct SweepMaxLptOf {
    static constexpr int value = P::COST <= 120 ? 4 : 2;  // (single biquad sections declare 16: biquad_sections.h)
};
template <class P>
struct SweepMaxLptOf<P, std::void_t<decltype(P::SWEEP_MAX_LPT)>> {
    static constexpr int value = P::SWEEP_MAX_LPT;
};
// smallest lane count the sweep kernel takes: 49152, or 24576 for 4-byte outputs, where several frames share a segment (`fps`: 32768 lanes 0.69 of
// the HBM peak against 0.56 on the staged single-wave kernel, 24576 0.52 against 0.50; at 16384 lanes the staged kernel's 36 ns per frame win, 0.46
// against 0.42 — a `v_mad_i64_i32` occupies its SIMD for 16 cycles per 64-lane wave, five per step: 80 cycles x 4096 steps = 0.137 ms whatever the schedule)
constexpr size_t kSweepMinLanes = thr::kSweepMinLanes, kSweepMinLanesFps = thr::kSweepMinLanesFps;

template <class P, int LPT>
int launch_sweep_lpt(const typename P::Params &prm, uint32_t *st, const typename P::In *x, typename P::Out *y, size_t lanes, size_t frames, size_t xl,
                     size_t yl, size_t sp, const SweepGeom &g, hipStream_t s, unsigned xcdc)
{
    constexpr size_t bytes = sweep_lds_bytes<P>();
    // (IDSP_DIAG=1 IDSP_SWEEP_PACE=1: narrow blocks take the full blocks' schedule too; =2: rows off the grid as well)
    static const size_t pace_more = diag_size("IDSP_SWEEP_PACE", 0);
    static const bool no_pace = diag_env("IDSP_SWEEP_NO_PACE") != nullptr;  // IDSP_DIAG=1: nothing paced (8 / 16 blocks per workgroup: nothing on two barriers)
    const bool full = (g.bw == unsigned(kFmBlock) || pace_more >= 1) && (!xcdc || pace_more >= 2) && g.fps <= 1 && !no_pace;  // full blocks on the 64-byte grid: bound by memory
    // Full blocks of the cheap processors up to 4 blocks per workgroup: the paced one-barrier schedule.  8 and 16 blocks per workgroup (a frame is one
    // or two whole tiles, up to sixteen chains per thread): unpaced — the sleep no longer fits the skeleton — and on the plain two-barrier schedule for
    // the processors that declare SWEEP_BIG_TWO_BARRIER (the unclamped i32 DF1, f32 DF2T and `Normal` sections: 2^20 lanes 0.72 / 0.75 / 0.66 against
    // 0.68 / 0.69 / 0.64 on one barrier), on the one-barrier schedule for everything else (clamped sections 0.69-0.71 against 0.54-0.60, two-section
    // chains 0.68-0.72 against 0.65-0.69: profiles/r05_perf_big_lanes.txt).
    if constexpr (LPT <= 4) {
        if constexpr (sweep_cheap<P>()) {
            if (full) {
                if (int rc = ensure_dyn_lds<&stream_frame_major_sweep<P, LPT, kSweepNB, kSweepForm, kSweepPace>>(bytes)) return rc;
                hipLaunchKernelGGL((stream_frame_major_sweep<P, LPT, kSweepNB, kSweepForm, kSweepPace>), dim3(g.grid), dim3(kFmBlock), bytes, s, prm, st, x, y, lanes, frames,
                                   xl, yl, sp, g.bw, g.rounds, g.round_lanes, xcdc, g.fps);
                return launch_status();
            }
        }
    } else if constexpr (SweepBigTwoBarrierOf<P>::value) {
        static const bool xcdc_two = diag_env("IDSP_SWEEP_XCDC_TWO_BARRIER") != nullptr;  // IDSP_DIAG=1: rows off the grid on the two-barrier schedule too
        // In place (y == x) with a clone sub-block (a sub-block wholly beyond the data re-requests sub-block 0's rows to keep the request count
        // static, e.g. 983296 or 1000000 lanes at 16 blocks per workgroup): on the two-barrier schedule the store of tile i goes out while the
        // clone's request of the same x rows may still be in flight, and nothing orders the two.  Never observed (the request is six tile periods
        // old by then), but not guaranteed: such launches take the one-barrier schedule below, which waits for the next tile before it stores.
        const bool clone_in_place = static_cast<const void *>(x) == static_cast<const void *>(y) && size_t(g.grid) * size_t(LPT) * g.bw * g.rounds != lanes;
        if ((full || (xcdc_two && g.bw == unsigned(kFmBlock) && g.fps <= 1)) && !clone_in_place) {
            if (int rc = ensure_dyn_lds<&stream_frame_major_sweep<P, LPT, kSweepNB, 0, 0>>(bytes)) return rc;
            hipLaunchKernelGGL((stream_frame_major_sweep<P, LPT, kSweepNB, 0, 0>), dim3(g.grid), dim3(kFmBlock), bytes, s, prm, st, x, y, lanes, frames, xl, yl, sp, g.bw,
                               g.rounds, g.round_lanes, xcdc, g.fps);
            return launch_status();
        }
    }
    // Full blocks on rows off the 64-byte grid (the whole rounds of a dense tensor of 65536 k + 1 ... 3 lanes, odd pitches): the two-barrier schedule with
    // every request as two instructions (XC above) — 131072 lanes at pitch + 4: 0.62 of the peak against 0.59, 524288: 0.63 against 0.60, 65536: 0.65-0.72
    // against 0.61-0.65 (profiles/r05_exp_sweep_off_grid_full_blocks.txt).  Cheap processors only (one more instantiation per blocks-per-workgroup count).
    static const bool no_xc = diag_env("IDSP_SWEEP_NO_SPLIT_REQUESTS") != nullptr;  // IDSP_DIAG=1: one instruction per request on rows off the grid too
    if constexpr (sweep_cheap<P>()) {
      if (xcdc && g.bw == unsigned(kFmBlock) && g.fps <= 1 && !no_xc) {
        if (int rc = ensure_dyn_lds<&stream_frame_major_sweep<P, LPT, kSweepNB, 0, 0, kSweepT, false, true>>(bytes)) return rc;
        hipLaunchKernelGGL((stream_frame_major_sweep<P, LPT, kSweepNB, 0, 0, kSweepT, false, true>), dim3(g.grid), dim3(kFmBlock), bytes, s, prm, st, x, y, lanes, frames,
                           xl, yl, sp, g.bw, g.rounds, g.round_lanes, xcdc, g.fps);
        return launch_status();
      }
    }
    if (g.fps > 1) {
        if constexpr (LPT == 1 && sizeof(typename P::Out) == 4) {
            if (int rc = ensure_dyn_lds<&stream_frame_major_sweep<P, 1, kSweepNB, kSweepForm, 0, kSweepT, true>>(bytes)) return rc;
            hipLaunchKernelGGL((stream_frame_major_sweep<P, 1, kSweepNB, kSweepForm, 0, kSweepT, true>), dim3(g.grid), dim3(kFmBlock), bytes, s, prm, st, x, y, lanes, frames,
                               xl, yl, sp, g.bw, g.rounds, g.round_lanes, xcdc, g.fps);
            return launch_status();
        } else {
            return fail(IDSP_EINVAL, "internal: several frames per segment with %d blocks per workgroup", LPT);
        }
    }
    if (int rc = ensure_dyn_lds<&stream_frame_major_sweep<P, LPT, kSweepNB, kSweepForm>>(bytes)) return rc;
    hipLaunchKernelGGL((stream_frame_major_sweep<P, LPT, kSweepNB, kSweepForm>), dim3(g.grid), dim3(kFmBlock), bytes, s, prm, st, x, y, lanes, frames, xl, yl, sp, g.bw,
                       g.rounds, g.round_lanes, xcdc, g.fps);
    return launch_status();
}

// Whether the sweep kernel is the one to take for `lanes` lanes of P: when the launch is ONE sweep, or when its sweeps are at least 8 blocks per
// workgroup wide (2 MiB of every row).  Several narrower sweeps per launch are the panel walk again (rows at a stride, 1 MiB or less of each): 2^20 lanes
// in four sweeps of 4 blocks per workgroup 0.54-0.57 of the HBM peak for every family against 0.60-0.62 on round 4's dispatch, which those launches keep
// (profiles/r05_perf_big_lanes.txt).
template <class P>
bool sweep_takes(size_t lanes)
{
    SweepGeom g;
    static const unsigned max_grid = unsigned(diag_size("IDSP_SWEEP_MAX_GRID", 256));
    if (!sweep_geometry(lanes, SweepMaxLptOf<P>::value, g, max_grid ? max_grid : 256u)) return false;
    return g.rounds == 1 || g.lpt >= thr::kSweepMinLptSeveralSweeps || max_grid != 256;  // (a capped grid is the tests' way to reach several sweeps per launch at small sizes)
}

// The launch for `lanes` lanes (a multiple of 4: whole 16-byte pieces; rows need dword alignment only — `global_load_lds_dwordx4` and the
// 16-byte stores take it, round 3 — and off the 64-byte grid the blocks go to the XCDs in contiguous eighths).  Returns IDSP_OK or an error.
template <class P>
int launch_sweep(const typename P::Params &prm, uint32_t *st, const typename P::In *x, typename P::Out *y, size_t lanes, size_t frames, size_t xl, size_t yl,
                 size_t sp, hipStream_t s)
{
    const bool on_grid64 = reinterpret_cast<uintptr_t>(x) % 64 == 0 && reinterpret_cast<uintptr_t>(y) % 64 == 0 && (xl * sizeof(typename P::In)) % 64 == 0 &&
                           (yl * sizeof(typename P::Out)) % 64 == 0;
    static const bool no_xcdc = diag_env("IDSP_SWEEP_NO_XCDC") != nullptr;
    static const bool force_xcdc = diag_env("IDSP_SWEEP_FORCE_XCDC") != nullptr;  // IDSP_DIAG=1: the XCD-contiguous block order on every launch
    const unsigned xcdc = (!on_grid64 && !no_xcdc) || force_xcdc ? 1u : 0u;
    constexpr int kMax = SweepMaxLptOf<P>::value;
    SweepGeom g;
    // (IDSP_DIAG=1 IDSP_SWEEP_MAX_GRID=n: at most n workgroups — small tensors then reach every LPT and several sweeps per launch: tests)
    static const unsigned max_grid = unsigned(diag_size("IDSP_SWEEP_MAX_GRID", 256));
    if (!sweep_geometry(lanes, kMax, g, max_grid ? max_grid : 256u)) return fail(IDSP_EINVAL, "internal: no sweep geometry for %zu lanes", lanes);
    // few lanes per workgroup (below 32768 + lanes in all): several frames per one-KiB segment, so that a tile stays 8 KiB of real samples
    // (IDSP_DIAG=1 IDSP_SWEEP_NO_FPS=1: one frame per segment)
    static const bool no_fps = diag_env("IDSP_SWEEP_NO_FPS") != nullptr;
    if (g.lpt == 1 && sizeof(typename P::Out) == 4 && g.bw <= 128 && !no_fps) {
        unsigned f = 256u / g.bw;
        while (f > 1 && (size_t(f - 1) * kSweepT * (xl > yl ? xl : yl) + 256) * 4 >= (size_t(1) << 32)) f--;  // per-thread byte offsets are 32-bit
        g.fps = f;
    }
    static const char *const names[] = {"stream_frame_major_sweep[1 block/workgroup]", "stream_frame_major_sweep[2 blocks/workgroup]",
                                        "stream_frame_major_sweep[4 blocks/workgroup]", "stream_frame_major_sweep[8 blocks/workgroup]",
                                        "stream_frame_major_sweep[16 blocks/workgroup]"};
    int k = 0;
    while ((1 << k) < g.lpt) k++;
    static const char *const names_x[] = {"stream_frame_major_sweep[1 block/workgroup, XCD-contiguous]", "stream_frame_major_sweep[2 blocks/workgroup, XCD-contiguous]",
                                          "stream_frame_major_sweep[4 blocks/workgroup, XCD-contiguous]", "stream_frame_major_sweep[8 blocks/workgroup, XCD-contiguous]",
                                          "stream_frame_major_sweep[16 blocks/workgroup, XCD-contiguous]"};
    note_kernel(xcdc && g.grid >= 8 ? names_x[k] : names[k], typeid(P).name());
    if constexpr (kMax >= 16) {
        if (g.lpt == 16) return launch_sweep_lpt<P, 16>(prm, st, x, y, lanes, frames, xl, yl, sp, g, s, xcdc && g.grid >= 8);
    }
    if constexpr (kMax >= 8) {
        if (g.lpt == 8) return launch_sweep_lpt<P, 8>(prm, st, x, y, lanes, frames, xl, yl, sp, g, s, xcdc && g.grid >= 8);
    }
    if constexpr (kMax >= 4) {
        if (g.lpt == 4) return launch_sweep_lpt<P, 4>(prm, st, x, y, lanes, frames, xl, yl, sp, g, s, xcdc && g.grid >= 8);
    }
    if constexpr (kMax >= 2) {
        if (g.lpt == 2) return launch_sweep_lpt<P, 2>(prm, st, x, y, lanes, frames, xl, yl, sp, g, s, xcdc && g.grid >= 8);
    }
    return launch_sweep_lpt<P, 1>(prm, st, x, y, lanes, frames, xl, yl, sp, g, s, xcdc && g.grid >= 8);
}

}  // namespace idsp
