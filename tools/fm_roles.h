// fm_roles.h — FrameMajor per-lane recurrences on ROLE waves (round 5 EXPERIMENT: measured equal to the shipped kernel and not used by the library; tools/exp_fm_roles.hip).
//
// Replaces the same triple loop as lane_stream.h (dsp-process/src/process.rs:122-141 driven by `Lanes`,
// dsp-process/src/compose.rs:468-494).  stream_frame_major_lds (lane_stream.h) lets every wave of a 256-lane workgroup do
// all three jobs in turn: request rows into the LDS ring, walk its own lanes, store the finished rows.  vmcnt retires in
// order, so the hand-counted `s_waitcnt vmcnt(N)` that tells a wave "input tile i has landed" also waits for every STORE
// the wave issued before that tile's requests — a slow store acknowledgement shortens the read-ahead of the very wave
// that has to keep the reads coming, and at one wave per SIMD every such wait is exposed (measured, NOTES round 4: the
// C5 kernel with only its requests or only its stores loses 0.8 ms each against the skeleton, with both 2.3-3.3 ms).
// Here the three jobs belong to different waves of one workgroup:
//
//   waves 0-3            COMPUTE: thread t walks lane t of the 256-lane block — LDS in, LDS out, no global memory
//                        traffic besides the state planes at the ends of a block;
//   waves 4 .. 4+NLW-1   LOADERS: `global_load_lds_dwordx4` rows into a ring of NB input tiles; their vmcnt sees
//                        nothing but their own requests, so the ring really is NB - 1 tiles deep at all times;
//   the NSW waves after  STORERS: `ds_read_b128` from the finished output tile, 16-byte nontemporal stores; they never
//                        wait for a memory acknowledgement at all.
//
// (NSW = 0: the NLW mover waves do both jobs — the old coupling, kept as the control of the experiment.)
//
// ONE workgroup barrier per tile of 8 frames.  Between barrier t and barrier t + 1
//   compute   turns input slot t % NB into output slot t % 2,
//   loaders   request tile t + NB - 1 into slot (t - 1) % NB (its readers passed barrier t) and then wait until
//             tile t + 1 has landed (at most (NB - 2) younger tiles of their own outstanding),
//   storers   move output slot (t - 1) % 2 (complete as of barrier t) to memory.
// The tile index t runs over ALL lane blocks a persistent workgroup owns, so the ring stays full across block
// boundaries.  Ragged last tiles re-request the last frame (static request count) and predicate compute and stores;
// a ragged last block (lanes % 256 != 0, lanes % 4 == 0) uses clone pieces exactly as stream_frame_major_lds does.
#pragma once

#include "lane_stream.h"

namespace idsp {

constexpr int kRolesT = 8;         // frames per tile
constexpr int kRolesCompute = 4;   // compute waves = 256 lanes per workgroup

#ifndef IDSP_ROLES_MOVER_PRIO  // experiment hook: s_setprio of the mover waves
#define IDSP_ROLES_MOVER_PRIO 0
#endif

template <class P>
constexpr size_t roles_lds_bytes(int nb)
{
    return (size_t(nb) * kRolesT * kFmBlock + 2 * size_t(kRolesT) * kFmBlock * (sizeof(typename P::Out) / 4) + P::LDS_WORDS) * 4;
}

template <class P, int NB, int NLW, int NSW>
__global__ __launch_bounds__((kRolesCompute + NLW + NSW) * kWave) void stream_frame_major_roles(
    const typename P::Params prm, uint32_t *st, const typename P::In *x, typename P::Out *y,
    const size_t lanes, const size_t frames, const size_t xl, const size_t yl, const size_t slanes, const int order)
{
    using In = typename P::In;
    using Out = typename P::Out;
    static_assert(P::HAS_IN && P::IN_DIV == 1 && sizeof(In) == 4, "role kernel: one 4-byte input per lane and frame");
    static_assert(NB >= 2 && NLW >= 1 && kRolesT % NLW == 0 && (NSW == 0 || kRolesT % NSW == 0), "mover waves divide the tile rows");
    constexpr int OW = sizeof(Out) / 4, B = BatchOf<P>::value, T = kRolesT;
    constexpr int NMW = NSW ? NSW : NLW;  // waves that store
    constexpr int RPL = T / NLW, RPS = T / NMW;
    static_assert(T % B == 0, "tile rows are whole batches");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *tin = smem;                             // [NB][T][256]
    uint32_t *tout = smem + NB * T * kFmBlock;        // [2][T][256 * OW]
    uint32_t *ptab = tout + 2 * T * kFmBlock * OW;    // [P::LDS_WORDS]
    const int tid = threadIdx.x, lid = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)smem;

    if constexpr (P::LDS_WORDS > 0) P::fill_shared(ptab, tid, int(blockDim.x));  // published by barrier 0

    // lane blocks of this workgroup, in the panel order of stream_frame_major_lds (order 3: the upper half of the grid
    // starts half way through its panels)
    const size_t nblocks = (lanes + kFmBlock - 1) / kFmBlock;
    const size_t rounds = (nblocks + gridDim.x - 1) / gridDim.x;
    const size_t ntiles = (frames + T - 1) / T;
    auto block_of = [&](size_t k) -> size_t {
        size_t rr = k;
        if (order == 3) rr = (k + (blockIdx.x >= gridDim.x / 2 ? rounds / 2 : 0)) % rounds;
        if (order == 1) rr = (k + (blockIdx.x & 7) * ((rounds + 7) / 8)) % rounds;
        return size_t(blockIdx.x) + rr * gridDim.x;
    };
    size_t owned = 0;
    for (size_t k = 0; k < rounds; k++) owned += block_of(k) < nblocks;
    const size_t ntotal = owned * ntiles;  // tiles of this workgroup
    if (ntotal == 0) return;
    // (panel, tile) cursor of a role
    struct Cur {
        size_t k, v, blk;
    };
    auto first = [&]() {
        Cur c{0, 0, 0};
        while (block_of(c.k) >= nblocks) c.k++;
        c.blk = block_of(c.k);
        return c;
    };
    auto advance = [&](Cur &c) {  // false: past the last tile
        if (++c.v < ntiles) return true;
        c.v = 0;
        do c.k++;
        while (c.k < rounds && block_of(c.k) >= nblocks);
        if (c.k >= rounds) return false;
        c.blk = block_of(c.k);
        return true;
    };
    // first lane of this thread's 16-byte piece inside a block (clone pieces in a ragged last block)
    auto piece_of = [&](size_t blk) {
        const size_t avail = lanes - blk * kFmBlock;
        return avail < size_t(kFmBlock) && size_t(lid * 4) >= avail ? int(size_t(lid * 4) % avail) : lid * 4;
    };

    if (wave < kRolesCompute) {
        // ------------------------------------------------------------------ compute
        P p;
        if constexpr (P::LDS_WORDS > 0) p.set_shared(ptab);
        Cur c = first();
        int slot = 0, par = 0;
        bool more = true;
        while (more) {
            const size_t lane0 = c.blk * kFmBlock;
            const bool lane_ok = lane0 + size_t(tid) < lanes;
            p.load(prm, st, slanes, lane_ok ? lane0 + tid : lanes - 1);
            for (size_t v = 0; v < ntiles; v++) {
                lds_barrier();  // barrier t: tile t has landed, output slot t % 2 is free
                const uint32_t *in = tin + slot * T * kFmBlock;
                uint32_t *o = tout + par * T * kFmBlock * OW;
                const int ns = frames - v * T < size_t(T) ? int(frames - v * T) : T;
                auto rows = [&](auto full) {
                    constexpr bool FULL = decltype(full)::value;
                    if constexpr (B > 1) {
#pragma unroll
                        for (int r0 = 0; r0 < T; r0 += B) {
                            if (!FULL && r0 >= ns) break;
                            typename P::Pre pre[B];
#pragma unroll
                            for (int b = 0; b < B; b++)
                                if (FULL || r0 + b < ns) pre[b] = p.pre(prm);
#pragma unroll
                            for (int b = 0; b < B; b++) {
                                const int r = r0 + b;
                                if (FULL || r < ns) to_words<Out>(p.step(prm, __builtin_bit_cast(In, in[r * kFmBlock + tid]), pre[b]), o + (r * kFmBlock + tid) * OW);
                            }
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < T; r++)
                            if (FULL || r < ns) to_words<Out>(p.step(prm, __builtin_bit_cast(In, in[r * kFmBlock + tid])), o + (r * kFmBlock + tid) * OW);
                    }
                };
                if (ns == T)
                    rows(std::true_type{});
                else
                    rows(std::false_type{});
                slot = slot + 1 == NB ? 0 : slot + 1;
                par ^= 1;
            }
            if (lane_ok) p.store(prm, st, slanes, lane0 + tid);
            c.v = ntiles - 1;
            more = advance(c);
        }
        lds_barrier();  // barrier ntotal: the last output tile is complete
    } else if (wave < kRolesCompute + NLW) {
        // ------------------------------------------------------------------ loaders (and storers when NSW == 0)
        if (IDSP_ROLES_MOVER_PRIO) __builtin_amdgcn_s_setprio(IDSP_ROLES_MOVER_PRIO);
        const int lw = wave - kRolesCompute;
        Cur c = first();
        int lid4 = piece_of(c.blk);
        size_t cblk = c.blk;
        bool have = true;
        int slot = 0;
        auto issue = [&]() {  // the tile under the cursor into `slot`, then advance both
            if (c.blk != cblk) cblk = c.blk, lid4 = piece_of(c.blk);
            const In *base = x + c.blk * kFmBlock;
#pragma unroll
            for (int j = 0; j < RPL; j++) {
                const int r = lw + NLW * j;
                size_t f = c.v * T + r;
                f = f < frames ? f : frames - 1;  // ragged tile: a static number of requests
                glds16_s(base + f * xl, uint32_t(lid4) * 4, lds_base + uint32_t((slot * T + r) * kFmBlock * 4));
            }
            slot = slot + 1 == NB ? 0 : slot + 1;
            have = advance(c);
        };
        // storing half of a coupled mover (NSW == 0)
        Cur cs = first();
        int sl4 = piece_of(cs.blk);
        size_t sblk = cs.blk;
        int par = 0;
        auto store_tile = [&]() {
            if (cs.blk != sblk) sblk = cs.blk, sl4 = piece_of(cs.blk);
            const uint32_t *o = tout + par * T * kFmBlock * OW;
            uint32_t *yw = reinterpret_cast<uint32_t *>(y) + cs.blk * kFmBlock * OW;
#pragma unroll
            for (int j = 0; j < RPS; j++) {
                const int r = lw + NMW * j;
                const size_t f = cs.v * T + r;
                if (f < frames) {
#pragma unroll
                    for (int h = 0; h < OW; h++) {
                        const u32x4 v4 = *reinterpret_cast<const u32x4 *>(o + (r * OW + h) * kFmBlock + sl4);
                        __builtin_nontemporal_store(v4, reinterpret_cast<u32x4 *>(yw + f * yl * OW + h * kFmBlock + sl4));
                    }
                }
            }
            par ^= 1;
            advance(cs);
        };
        for (int i = 0; i < NB - 1 && have; i++) issue();
        if (have)
            wait_vmcnt<(NB - 2) * RPL>();  // tiles 1 .. NB - 2 may be outstanding: tile 0 has landed
        else
            wait_vmcnt<0>();
        lds_barrier();  // barrier 0
        for (size_t t = 0; t < ntotal; t++) {
            if constexpr (NSW == 0) {
                // coupled: this wave's vmcnt counts its stores too (RPS * OW per tile, interleaved with the requests)
                const bool more = have;
                if (have) issue();
                if (t >= 1) store_tile();
                // younger than tile t + 1's requests, once every slot of the pattern is filled (t >= NB - 1): the requests of
                // tiles t + 2 .. t + NB - 1 and the stores of tiles t - NB + 1 .. t - 1
                if (more && t >= size_t(NB))
                    wait_vmcnt<(NB - 2) * RPL + (NB - 1) * RPS * OW>();
                else
                    wait_vmcnt<0>();
            } else {
                if (have) {
                    issue();  // tile t + NB - 1
                    wait_vmcnt<(NB - 2) * RPL>();  // tiles t + 2 .. t + NB - 1 may be outstanding: tile t + 1 has landed
                } else {
                    wait_vmcnt<0>();
                }
            }
            lds_barrier();  // barrier t + 1
        }
        if constexpr (NSW == 0) store_tile();
    } else {
        // ------------------------------------------------------------------ storers
        if (IDSP_ROLES_MOVER_PRIO) __builtin_amdgcn_s_setprio(IDSP_ROLES_MOVER_PRIO);
        const int sw = wave - kRolesCompute - NLW;
        Cur cs = first();
        int sl4 = piece_of(cs.blk);
        size_t sblk = cs.blk;
        int par = 0;
        auto store_tile = [&]() {
            if (cs.blk != sblk) sblk = cs.blk, sl4 = piece_of(cs.blk);
            const uint32_t *o = tout + par * T * kFmBlock * OW;
            uint32_t *yw = reinterpret_cast<uint32_t *>(y) + cs.blk * kFmBlock * OW;
#pragma unroll
            for (int j = 0; j < RPS; j++) {
                const int r = sw + NMW * j;
                const size_t f = cs.v * T + r;
                if (f < frames) {
#pragma unroll
                    for (int h = 0; h < OW; h++) {
                        const u32x4 v4 = *reinterpret_cast<const u32x4 *>(o + (r * OW + h) * kFmBlock + sl4);
                        __builtin_nontemporal_store(v4, reinterpret_cast<u32x4 *>(yw + f * yl * OW + h * kFmBlock + sl4));
                    }
                }
            }
            par ^= 1;
            advance(cs);
        };
        lds_barrier();  // barrier 0
        for (size_t t = 0; t < ntotal; t++) {
            if (t >= 1) store_tile();  // output tile t - 1
            lds_barrier();             // barrier t + 1
        }
        store_tile();  // output tile ntotal - 1
    }
}

}  // namespace idsp
