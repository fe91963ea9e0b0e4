// dds.hip — cossin(), the Accu phase-accumulator DDS, Lowpass<N> and the
// Lockin mixer over many lanes (reference: src/cossin.rs, src/accu.rs,
// src/lowpass.rs, src/lockin.rs, src/complex.rs).  All integer, bit-exact.
//
// cossin's 128-entry LUT (512 B) is staged once per workgroup into LDS from
// the compile-time table; per-lane gathers then cost an LDS read instead of a
// divergent constant-memory fetch.  Everything after the gather is ~40 VALU
// ops in registers, fused in the lock-in kernel with the mixer and the
// lowpass cascade so a phase never touches memory.
#include "dds_dev.h"

namespace idsp {

// lockin_waves_{iq,arg,norm_sqr}.hip: one lane's work spread over 4 or 6 waves (lockin_waves.h)
int lockin_waves_iq(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, void *y, size_t lanes, size_t frames, int layout,
                    int waves, hipStream_t s, size_t pitch = 0);
int lockin_waves_arg(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, void *y, size_t lanes, size_t frames, int layout,
                     int waves, hipStream_t s, size_t pitch = 0);
int lockin_waves_norm_sqr(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, void *y, size_t lanes, size_t frames,
                          int layout, int waves, hipStream_t s, size_t pitch = 0);

// lockin_stream_{iq,arg,norm_sqr}.hip, lowpass.hip: the stream processors behind the multi-wave kernels (lockin_stream_procs.h)
int lockin_stream_iq(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, bool split, hipStream_t s,
                     size_t pitch = 0);
int lockin_stream_arg(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, hipStream_t s, size_t pitch = 0);
int lockin_stream_norm_sqr(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int64_t *y, size_t lanes, size_t frames, int layout, hipStream_t s,
                           size_t pitch = 0);
int lowpass_stream(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, hipStream_t s);

namespace {

// The multi-wave kernels take FrameMajor always and LaneMajor for whole 16-frame batches on 16-byte aligned rows
// (IDSP_LOCKIN_NO_WAVES=1 keeps everything on the one- / two-thread-per-lane stream kernels below).
// Returns the wave count per 64 lanes, 0 = use the stream kernels.
inline int lockin_waves_for(const void *x, const void *y, size_t lanes, size_t frames, int layout, bool heavy_readout, int arm_weight = 0)
{
    static const bool off = diag_env("IDSP_LOCKIN_NO_WAVES") != nullptr;
    static const int forced = [] {
        const char *e = diag_env("IDSP_LOCKIN_WAVES");
        return e ? atoi(e) : 0;
    }();
    if (off || frames == 0) return 0;
    if (lanes >= (size_t(1) << 28)) return 0;  // 32-bit thread offsets inside a row (8-byte elements, 3 input rows of 4 lanes bytes)
    if (layout == IDSP_LANE_MAJOR && !(frames % 16 == 0 && (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 == 0))
        return 0;
    if (forced == 4 || forced == 6) return forced;
    // FrameMajor up to one workgroup per CU: six waves (four read-out waves) with 16-frame batches — the read-out path is what an interval
    // waits for there (`[Lowpass<1>; 1]` 0.183 -> 0.156 ms, `[Lowpass<2>; 1]`->arg 0.324 -> 0.283 at 16384 lanes x 4096, profiles/r03_perf_c4small.jsonl;
    // `[Lowpass<N>; 2]` is the stage-wave kernel's at these lane counts).  Two 6-wave workgroups with 16-frame batches do not share a CU, so not above.
    if (layout == IDSP_FRAME_MAJOR && lanes <= 16384) return 6;
    // measured at 4096 frames (arg read-out): 6 waves 0.64 ms at 32768 lanes (4 waves: 0.73), 4 waves 1.07 ms at 65536 (6: 1.12)
    // (`arm_weight` = order x cascade: with six and more second-order-equivalents per arm the arm waves are the longer path again and the four-wave
    // form with 16-frame batches wins: `[Lowpass<2>; 4]` -> arg at 32768 lanes 0.58 against 0.74 ms, profiles/r03_perf_c4small_32768.jsonl)
    return heavy_readout && lanes <= kSplitMaxLanes && arm_weight < 6 ? 6 : 4;
}

// LaneMajor rows that are not whole 16-frame batches (round 4): the multi-wave kernel takes the whole batches of every row at the call's
// row pitch and a stream kernel the last frames % 16 behind it (same stream; the state carries over as between two calls).  Rows
// must keep their 16-byte alignment: frames % 4 == 0.  Returns the frames of the first part, 0 = the call is not of that kind.
inline size_t lockin_lm_body(const void *x, const void *y, size_t frames, int layout)
{
    static const bool off = diag_env("IDSP_LOCKIN_NO_LM_TAIL") != nullptr;
    if (off || layout != IDSP_LANE_MAJOR || frames % 16 == 0 || frames % 4 != 0 || frames < 32) return 0;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 != 0) return 0;
    return frames - frames % 16;
}

// ------------------------------------------------------------- processors
// Accu (src/accu.rs:34-41, pre-increment) -> Complex::from_angle (src/complex.rs:237-240)
// CIRCLE: cossin through the full-circle table (dds_dev.h cossin_circle: 9 VALU + 1 LDS instruction instead of ~28 + 1, 16 KiB of LDS
// per workgroup, filled at the start of every workgroup) — taken by FrameMajor calls of 256 frames or more (idsp_dds_i32); the 512-byte
// table otherwise (short calls, LaneMajor — its staged kernel needs the LDS for the 32 KiB tile slot of every wave — and the one-thread-per-lane form).
template <bool CIRCLE>
struct CosTab {
    static constexpr int WORDS = CIRCLE ? kCosCircleWords : (1 << kCossinDepth);
    static __device__ __forceinline__ void fill(uint32_t *sh, int tid, int n)
    {
        if constexpr (CIRCLE)
            fill_cossin_circle(sh, tid, n);
        else
            fill_cossin(sh, tid, n);
    }
    static __device__ __forceinline__ Cplx eval(uint32_t ph, const uint32_t *t)
    {
        if constexpr (CIRCLE)
            return cossin_circle(ph, t);
        else
            return cossin_dev(int32_t(ph), t);
    }
};

template <bool CIRCLE>
struct DdsProcT {
    using In = int32_t;  // unused
    using Out = Cplx;
    static constexpr bool HAS_IN = false;
    static constexpr int LDS_WORDS = CosTab<CIRCLE>::WORDS;
    static constexpr int IN_DIV = 1;
    static constexpr int COST = 100;
    struct Params {
        int32_t unused;
    };
    const uint32_t *lut;
    uint32_t acc, inc;
    static __device__ __forceinline__ void fill_shared(uint32_t *sh, int tid, int n) { CosTab<CIRCLE>::fill(sh, tid, n); }
    __device__ __forceinline__ void set_shared(const uint32_t *sh) { lut = sh; }
    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t lanes, size_t lane)
    {
        acc = st[lane];
        inc = st[lanes + lane];
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane) { st[lane] = acc; }
    __device__ __forceinline__ Out step(const Params &, In)
    {
        acc += inc;
        return CosTab<CIRCLE>::eval(acc, lut);
    }
};
using DdsProc = DdsProcT<false>;

template <bool CIRCLE>
struct DdsSplitProcT {
    using In = int32_t;  // unused
    using Out = int32_t;
    static constexpr bool HAS_IN = false;
    static constexpr int LDS_WORDS = CosTab<CIRCLE>::WORDS;
    static constexpr int IN_DIV = 2;
    static constexpr int COST = 100;
    struct Params {
        int32_t unused;
    };
    const uint32_t *lut;
    uint32_t acc, inc;
    bool q;
    static __device__ __forceinline__ void fill_shared(uint32_t *sh, int tid, int n) { CosTab<CIRCLE>::fill(sh, tid, n); }
    __device__ __forceinline__ void set_shared(const uint32_t *sh) { lut = sh; }
    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t vlanes, size_t vlane)
    {
        q = vlane & 1;
        acc = st[vlane / 2];
        inc = st[vlanes / 2 + vlane / 2];
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t, size_t vlane)
    {
        if (!q) st[vlane / 2] = acc;
    }
    static constexpr int BATCH = 4;
    using Pre = int32_t;
    __device__ __forceinline__ Pre pre(const Params &)
    {
        acc += inc;
        const Cplx c = CosTab<CIRCLE>::eval(acc, lut);
        return q ? c.im : c.re;
    }
    __device__ __forceinline__ void pre_batch(const Params &, Pre (&out)[BATCH])
    {
#pragma unroll
        for (int j = 0; j < BATCH / 2; j++) {
            const uint32_t ph = acc + inc * uint32_t(2 * j + 1) + (q ? inc : 0u);
            const Cplx c = CosTab<CIRCLE>::eval(ph, lut);
            const int32_t mine = q ? c.im : c.re, other = q ? c.re : c.im;
            const int32_t recv = pair_swap(other);
            out[2 * j] = q ? recv : mine;
            out[2 * j + 1] = q ? mine : recv;
        }
        acc += inc * uint32_t(BATCH);
    }
    __device__ __forceinline__ Out step(const Params &, In, const Pre &v) { return v; }
};
using DdsSplitProc = DdsSplitProcT<false>;

typedef int32_t i32x2 __attribute__((ext_vector_type(2)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

// two phases per thread and trip: one dwordx2 load, one dwordx4 store, so that every store instruction
// of a wave writes 1 KiB of whole lines (four phases per thread would split each line over two
// instructions: measured slower).  `vec` needs 16-byte aligned buffers.
__global__ __launch_bounds__(256) void cossin_kernel(const int32_t *phase, Cplx *out, size_t n, bool vec)
{
    __shared__ uint32_t lut[1 << kCossinDepth];
    fill_cossin(lut, threadIdx.x, 256);
    __syncthreads();
    const size_t stride = size_t(gridDim.x) * 256, t = size_t(blockIdx.x) * 256 + threadIdx.x;
    const size_t nv = vec ? n / 2 : 0;
    for (size_t i = t; i < nv; i += stride) {
        const i32x2 p = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(phase) + i);
        const Cplx a = cossin_dev(p.x, lut), b = cossin_dev(p.y, lut);
        __builtin_nontemporal_store(i32x4{a.re, a.im, b.re, b.im}, reinterpret_cast<i32x4 *>(out) + i);
    }
    for (size_t i = nv * 2 + t; i < n; i += stride) out[i] = cossin_dev(phase[i], lut);
}

__global__ __launch_bounds__(256) void atan2_kernel(const Cplx *xy, int32_t *out, size_t n, bool vec)
{
    __shared__ uint32_t tab[32];
    if (threadIdx.x < 32) tab[threadIdx.x] = d_atan2_table[threadIdx.x];
    __syncthreads();
    const size_t stride = size_t(gridDim.x) * 256, t = size_t(blockIdx.x) * 256 + threadIdx.x;
    const size_t nv = vec ? n / 4 : 0;
    for (size_t i = t; i < nv; i += stride) {  // rows are [x, y] = [re, im]
        const i32x4 a = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(xy) + 2 * i);
        const i32x4 b = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(xy) + 2 * i + 1);
        const i32x4 r = {atan2_dev(a.y, a.x, tab), atan2_dev(a.w, a.z, tab), atan2_dev(b.y, b.x, tab), atan2_dev(b.w, b.z, tab)};
        __builtin_nontemporal_store(r, reinterpret_cast<i32x4 *>(out) + i);
    }
    for (size_t i = nv * 4 + t; i < n; i += stride) {
        const Cplx v = xy[i];
        out[i] = atan2_dev(v.im, v.re, tab);
    }
}

// FM discriminator + deemphasis (examples/fm_disc.rs:25-50).  In = Complex<Q32<32>> bits as a 2-vector.
typedef int32_t cplx_bits __attribute__((ext_vector_type(2)));
struct FmDiscProc {
    using In = cplx_bits;
    using Out = int32_t;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 32;  // atan2 reciprocal table
    static constexpr int IN_DIV = 1;
    static constexpr int COST = 130;
    struct Params {
        int32_t carrier;
        bq::SecI32 sec;
    };
    const uint32_t *tab;
    uint32_t has_prev;
    int32_t pre, pim;
    uint32_t s[4];
    static __device__ __forceinline__ void fill_shared(uint32_t *sh, int tid, int n)
    {
        for (int i = tid; i < 32; i += n) sh[i] = d_atan2_table[i];
    }
    __device__ __forceinline__ void set_shared(const uint32_t *sh) { tab = sh; }
    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t lanes, size_t lane)
    {
        has_prev = st[lane];
        pre = int32_t(st[lanes + lane]);
        pim = int32_t(st[2 * lanes + lane]);
#pragma unroll
        for (int w = 0; w < 4; w++) s[w] = st[size_t(3 + w) * lanes + lane];
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane)
    {
        st[lane] = has_prev;
        st[lanes + lane] = uint32_t(pre);
        st[2 * lanes + lane] = uint32_t(pim);
#pragma unroll
        for (int w = 0; w < 4; w++) st[size_t(3 + w) * lanes + lane] = s[w];
    }
    __device__ __forceinline__ Out step(const Params &p, In x)
    {
        int32_t d = 0;
        // `prev.replace(x)`: None -> 0 (fm_disc.rs:33-35)
        const int32_t cim = int32_t(0u - uint32_t(pim));  // conj (src/complex.rs:55-57)
        const int64_t re = int64_t(uint64_t(int64_t(x.x) * pre) - uint64_t(int64_t(x.y) * cim));
        const int64_t im = int64_t(uint64_t(int64_t(x.x) * cim) + uint64_t(int64_t(x.y) * pre));
        const int32_t a = atan2_dev(int32_t(im >> 32), int32_t(re >> 32), tab);
        d = has_prev ? int32_t(uint32_t(a) - uint32_t(p.carrier)) : 0;
        has_prev = 1u;
        pre = x.x, pim = x.y;
        return bq::Df1I32<false>::step(p.sec, s, d);
    }
};

// FM discriminator on role waves (round 3, FRAME_MAJOR).  One thread per lane is ~55 VALU instructions per sample on a single
// wave per SIMD at 65536 lanes (0.44 of the HBM peak, issue-bound) — but only the deemphasis biquad is a recurrence: the
// discriminator `arg(x[f] * conj(x[f-1])) - carrier` needs nothing but the previous INPUT sample, which is in the buffer (the
// state's `prev` only for frame 0).  So a workgroup of NF + 1 waves owns 64 lanes: front wave w computes the discriminator of
// frames [w FPW, (w + 1) FPW) of every tile of T = NF FPW frames (it loads those FPW rows plus the one before them) into an
// LDS tile, the last wave runs the biquad down the previous tile and stores y; one workgroup barrier per tile, tiles double
// buffered, whole tiles without per-frame predicates.  Five waves per SIMD instead of one, the same per-sample functions in the
// same order: results are those of FmDiscProc bit for bit.
// (registers: left alone the compiler spends 245 VGPRs on hoisted loads and ONE workgroup fits a CU — 1.01 ms at the C2 shape,
// slower than the stream kernel; four workgroups per CU need <= 96)
#ifdef IDSP_FMD_ABL_NOSTORE  // tools/exp_lm_ablate.sh (UNIT=dds): timing variants, conditions never true at run time
#define IDSP_FMD_ST_ON (frames == 1)
#else
#define IDSP_FMD_ST_ON true
#endif
#ifdef IDSP_FMD_ABL_NOLOAD
#define IDSP_FMD_LD_ON (frames == 1)
#else
#define IDSP_FMD_LD_ON true
#endif
#ifndef IDSP_FMD_WPE
#define IDSP_FMD_WPE 5
#endif
template <int NF>
__global__ __launch_bounds__(kWave *(NF + 1)) __attribute__((amdgpu_waves_per_eu(IDSP_FMD_WPE, IDSP_FMD_WPE))) void fm_disc_waves_kernel(const FmDiscProc::Params prm, uint32_t *st, const cplx_bits *x, int32_t *y,
                                                                        const size_t lanes, const size_t frames, const size_t xl, const size_t yl)
{
    constexpr int FPW = 8, T = NF * FPW;
    __shared__ int32_t tile[2][T][kWave];
    __shared__ uint32_t tab[32];
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) / kWave), lid = int(threadIdx.x) % kWave;
    if (threadIdx.x < 32) tab[threadIdx.x] = d_atan2_table[threadIdx.x];
    __syncthreads();
    const size_t lane = size_t(blockIdx.x) * kWave + lid;
    const bool active = lane < lanes;
    const size_t la = active ? lane : lanes - 1;  // idle threads of the last workgroup shadow a valid lane, stores masked
    const size_t nfull = frames / T;
    const int ntail = int(frames - nfull * T);
    if (wave < NF) {
        const cplx_bits *xp = x + la;
        const uint32_t has_prev0 = st[la];
        const cplx_bits prev0 = {int32_t(st[lanes + la]), int32_t(st[2 * lanes + la])};
        const int j0 = wave * FPW;  // first frame of this wave inside a tile
        // z = x * conj(prev), d = arg(z) - carrier (FmDiscProc::step, examples/fm_disc.rs:33-41)
        auto disc = [&](cplx_bits c, cplx_bits pv) __attribute__((always_inline)) {
            const int32_t cim = int32_t(0u - uint32_t(pv.y));
            const int64_t re = int64_t(uint64_t(int64_t(c.x) * pv.x) - uint64_t(int64_t(c.y) * cim));
            const int64_t im = int64_t(uint64_t(int64_t(c.x) * cim) + uint64_t(int64_t(c.y) * pv.x));
            return int32_t(uint32_t(atan2_dev(int32_t(im >> 32), int32_t(re >> 32), tab)) - uint32_t(prm.carrier));
        };
        cplx_bits cur[FPW + 1], nxt[FPW + 1];  // [0]: the frame before this wave's first
        auto fetch = [&](size_t k, cplx_bits(&dst)[FPW + 1], auto full) __attribute__((always_inline)) {
            const size_t f0 = k * T + j0;
#pragma unroll
            for (int j = 0; j <= FPW; j++) {
                if (j == 0 && f0 == 0) {
                    dst[0] = prev0;  // frame -1 is the state's `prev`
                } else if ((decltype(full)::value || j0 + j - 1 < ntail) && IDSP_FMD_LD_ON) {
                    dst[j] = nt_load<true>(xp + (f0 + j - 1) * xl);
                }
            }
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
            for (int j = 0; j < FPW; j++) {
                int32_t d = disc(cur[j + 1], cur[j]);
                if (j == 0 && k == 0 && wave == 0) d = has_prev0 ? d : 0;  // `prev` was None: 0 (fm_disc.rs:33-35)
                tile[k & 1][j0 + j][lid] = d;
            }
#pragma unroll
            for (int j = 0; j <= FPW; j++) cur[j] = nxt[j];
            lds_barrier();  // tile k complete; the biquad wave has finished tile k - 1
        }
        if (ntail) {
            for (int j = 0; j < FPW && j0 + j < ntail; j++) {
                int32_t d = disc(cur[j + 1], cur[j]);
                if (j == 0 && nfull == 0 && wave == 0) d = has_prev0 ? d : 0;
                tile[nfull & 1][j0 + j][lid] = d;
            }
            lds_barrier();
        }
        lds_barrier();  // pairs with the biquad wave's last interval
        if (wave == 0 && active) {
            const cplx_bits last = xp[(frames - 1) * xl];
            st[lane] = 1u;
            st[lanes + lane] = uint32_t(last.x);
            st[2 * lanes + lane] = uint32_t(last.y);
        }
    } else {
        uint32_t s[4];
#pragma unroll
        for (int w = 0; w < 4; w++) s[w] = st[size_t(3 + w) * lanes + la];
        int32_t *yp = y + la;
        lds_barrier();  // tile 0
        for (size_t k = 0; k < nfull; k++) {
            int32_t v[T];
#pragma unroll
            for (int f = 0; f < T; f++) v[f] = tile[k & 1][f][lid];
#pragma unroll
            for (int f = 0; f < T; f++) {
                // (no `if (active)`: an idle thread of the last workgroup shadows lane `lanes - 1` — same input, same state, the
                // same value to the same address — and a predicate per store is a branch per sample in the ISA)
                const int32_t o = bq::Df1I32<false>::step(prm.sec, s, v[f]);
                if (IDSP_FMD_ST_ON) nt_store<true>(yp + (k * T + f) * yl, o);
            }
            lds_barrier();
        }
        if (ntail) {
            for (int f = 0; f < ntail; f++) {
                const int32_t o = bq::Df1I32<false>::step(prm.sec, s, tile[nfull & 1][f][lid]);
                if (active) nt_store<true>(yp + (nfull * T + f) * yl, o);
            }
            lds_barrier();
        }
        if (active) {
#pragma unroll
            for (int w = 0; w < 4; w++) st[size_t(3 + w) * lanes + lane] = s[w];
        }
    }
}


// FM discriminator on role waves, LANE_MAJOR (round 4).  Same roles as above — four front waves with eight frames of a 32-frame
// tile each, a fifth wave with the deemphasis biquad one tile behind — but every global access moves whole 128-byte lines, as in
// the LaneMajor lock-in (lockin_waves.h): the input tile (two lines of 16 `Complex<i32>` per lane) arrives by LDS-DMA one tile
// ahead, eight threads per line, pieces swizzled so that a thread's ds_read_b128 of its own row are conflict free; a front-wave
// thread reads its eight samples and the one before them from LDS (wave 0 carries the last sample of the tile before in a
// register), a barrier frees the slot and the requests of the next tile fly during the arithmetic (one slot: 34 KiB of LDS, four
// workgroups per CU — with two slots and one barrier per tile, 51 KiB and three: 0.98 ms); the discriminator values go into one ROW per lane (pitch 36 words), the biquad wave
// turns its row into outputs in place and the wave then re-reads the rows line-wise — eight threads per lane — and stores eight
// whole lines per instruction.  Two workgroup barriers per tile.  Whole tiles on 16-byte aligned rows; everything else stays on
// the tile kernel (stream_lane_major<FmDiscProc>), whose 64 different lines per access are what it is bound by (1.25 ms at
// 65536 lanes x 4096 frames; the FrameMajor role kernel with per-thread vectors on LaneMajor rows: 2.30 ms, round 3).
__global__ __launch_bounds__(kWave * 5) void fm_disc_waves_lm_kernel(const FmDiscProc::Params prm, uint32_t *st, const cplx_bits *x, int32_t *y,
                                                                     const size_t lanes, const size_t frames, const size_t pitch,
                                                                     const unsigned skew, const unsigned skew_shift, const unsigned skew_mod)
{
    constexpr int NF = 4, FPW = 8, T = NF * FPW, RS = T + 4;
    // start-up stagger (lockin_waves.h, "lanes in phase"): workgroup b waits ((b >> shift) % mod) * skew ticks of 10 ns
    if (const long long d = blockIdx.x < 1024 ? (long long)(skew) * ((blockIdx.x >> skew_shift) % skew_mod) : 0) {  // the first round only
        const long long t0 = wall_clock64();
        // (bounded: an s_sleep(8) is at least 0.2 us = 20 ticks, so d / 8 rounds are more than enough even if the counter stood still)
        for (long long spins = d / 8 + 16; spins > 0 && wall_clock64() - t0 < d; spins--) __builtin_amdgcn_s_sleep(8);
    }
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) int32_t tile[2][kWave * RS];
    __shared__ __attribute__((aligned(16))) uint32_t xs[2][kWave * 32];  // [line of the tile][64 rows x 128 bytes]: ONE slot, 34 KiB in all, four workgroups per CU
    __shared__ uint32_t tab[32];
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) / kWave), lid = int(threadIdx.x) % kWave;
    if (threadIdx.x < 32) tab[threadIdx.x] = d_atan2_table[threadIdx.x];
    const size_t lane = size_t(blockIdx.x) * kWave + lid;
    const bool active = lane < lanes;
    const size_t la = active ? lane : lanes - 1;  // idle threads of the last workgroup shadow a valid lane, stores masked
    const size_t ntiles = frames / T;             // launcher: whole tiles of rows `pitch` elements apart (the rest of a row: the tile kernel)
    if (wave < NF) {
        const uint32_t has_prev0 = st[la];
        cplx_bits carry = {int32_t(st[lanes + la]), int32_t(st[2 * lanes + la])};  // wave 0: the sample before the tile
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the state has landed before the first DMA request is counted
        // instruction j = 4 wave + g of the 16 of a tile: line j / 8 of the tile, rows of lanes j % 8 + 8 (lid / 8), piece (lid % 8) ^ (j % 8)
        auto dma = [&](size_t k) {
            if (!IDSP_FMD_LD_ON) return;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int j = 4 * wave + g, line = j >> 3, jj = j & 7;
                size_t gl = size_t(blockIdx.x) * kWave + size_t(jj + 8 * (lid / 8));
                gl = gl < lanes ? gl : lanes - 1;
                glds16(x + gl * pitch + k * T + size_t(line * 16 + ((lid % 8) ^ jj) * 2),
                       uint32_t(reinterpret_cast<uintptr_t>(&xs[line][jj * 256])));
            }
        };
        const uint32_t own = uint32_t((lid % 8) * 8 + lid / 8) * 128, sw = uint32_t(lid % 8);  // own row; piece p sits at 16 (p ^ sw)
        auto piece = [&](int line, int p) {
            return *reinterpret_cast<const i32x4 *>(reinterpret_cast<const char *>(&xs[line][0]) + own + ((uint32_t(p) ^ sw) * 16));
        };
        auto disc = [&](cplx_bits c, cplx_bits pv) __attribute__((always_inline)) {
            const int32_t cim = int32_t(0u - uint32_t(pv.y));
            const int64_t re = int64_t(uint64_t(int64_t(c.x) * pv.x) - uint64_t(int64_t(c.y) * cim));
            const int64_t im = int64_t(uint64_t(int64_t(c.x) * cim) + uint64_t(int64_t(c.y) * pv.x));
            return int32_t(uint32_t(atan2_dev(int32_t(im >> 32), int32_t(re >> 32), tab)) - uint32_t(prm.carrier));
        };
        dma(0);
        wait_vmcnt<0>();
        lds_barrier();  // the table, tile 0
        const int line = wave >> 1, p0 = (wave & 1) * 4;
        for (size_t k = 0; k < ntiles; k++) {
            const int slot = int(k & 1);
            cplx_bits cur[FPW + 1];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const i32x4 a = piece(line, p0 + q);
                cur[1 + 2 * q] = cplx_bits{a.x, a.y}, cur[2 + 2 * q] = cplx_bits{a.z, a.w};
            }
            if (wave == 0) {
                cur[0] = carry;
                const i32x4 a = piece(1, 7);  // frame 31 of this tile: the sample before the next one
                carry = cplx_bits{a.z, a.w};
            } else {
                const i32x4 a = wave == 2 ? piece(0, 7) : piece(line, p0 - 1);
                cur[0] = cplx_bits{a.z, a.w};
            }
            lds_barrier();  // every front wave holds its samples of tile k: the slot is free
            if (k + 1 < ntiles) dma(k + 1);
            int32_t d[FPW];
#pragma unroll
            for (int j = 0; j < FPW; j++) d[j] = disc(cur[j + 1], cur[j]);
            if (k == 0 && wave == 0) d[0] = has_prev0 ? d[0] : 0;  // `prev` was None: 0 (fm_disc.rs:33-35)
            i32x4 *row = reinterpret_cast<i32x4 *>(&tile[slot][lid * RS + wave * FPW]);
            row[0] = i32x4{d[0], d[1], d[2], d[3]};
            row[1] = i32x4{d[4], d[5], d[6], d[7]};
            wait_vmcnt<0>();  // this wave's share of tile k + 1 has landed (it had the whole interval)
            lds_barrier();    // tile k complete, tile k + 1 in LDS; the biquad wave has finished tile k - 1
        }
        if (wave == 0 && active) {
            const cplx_bits last = x[lane * pitch + frames - 1];
            st[lane] = 1u;
            st[lanes + lane] = uint32_t(last.x);
            st[2 * lanes + lane] = uint32_t(last.y);
        }
    } else {
        uint32_t s[4];
#pragma unroll
        for (int w = 0; w < 4; w++) s[w] = st[size_t(3 + w) * lanes + la];
        lds_barrier();  // pairs with the front waves' first barrier
        for (size_t k = 0; k <= ntiles; k++) {
            if (k < ntiles) lds_barrier();  // the front waves' slot hand-over of interval k
            if (k == 0) {
                lds_barrier();  // tile 0 complete
                continue;
            }
            const size_t kt = k - 1;  // the tile this wave turns into outputs during interval k
            int32_t *rowp = &tile[kt & 1][lid * RS];
            int32_t v[T];
#pragma unroll
            for (int q = 0; q < T / 4; q++) {
                const i32x4 a = reinterpret_cast<const i32x4 *>(rowp)[q];
                v[4 * q] = a.x, v[4 * q + 1] = a.y, v[4 * q + 2] = a.z, v[4 * q + 3] = a.w;
            }
#pragma unroll
            for (int f = 0; f < T; f++) v[f] = bq::Df1I32<false>::step(prm.sec, s, v[f]);
#pragma unroll
            for (int q = 0; q < T / 4; q++) reinterpret_cast<i32x4 *>(rowp)[q] = i32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            // line-wise: in instruction u thread t holds frames 4 (t % 8) .. of lane 8 u + t / 8 (LDS operations of one wave stay in order)
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int ll = u * 8 + lid / 8;
                const size_t gl = size_t(blockIdx.x) * kWave + size_t(ll);
                const i32x4 a = *reinterpret_cast<const i32x4 *>(&tile[kt & 1][ll * RS + (lid % 8) * 4]);
                if (gl < lanes && IDSP_FMD_ST_ON) __builtin_nontemporal_store(a, reinterpret_cast<i32x4 *>(y + gl * pitch + kt * T) + lid % 8);
            }
            if (k < ntiles) lds_barrier();  // tile k complete
        }
        if (active) {
#pragma unroll
            for (int w = 0; w < 4; w++) st[size_t(3 + w) * lanes + lane] = s[w];
        }
    }
}


}  // namespace
}  // namespace idsp

using namespace idsp;

extern "C" {

int idsp_cossin_i32(const int32_t *phase, int32_t *out, size_t n, void *stream)
{
    if (n && (!phase || !out)) return fail(IDSP_EINVAL, "phase or out is NULL");
    if (n == 0) return IDSP_OK;
    const bool vec = (reinterpret_cast<uintptr_t>(phase) | reinterpret_cast<uintptr_t>(out)) % 16 == 0;
    size_t blocks = (n / (vec ? 2 : 1) + 255) / 256 + 1;
    if (blocks > 8192) blocks = 8192;
    note_kernel("cossin_kernel");
    hipLaunchKernelGGL(cossin_kernel, dim3(unsigned(blocks)), dim3(256), 0, as_stream(stream), phase,
                       reinterpret_cast<Cplx *>(out), n, vec);
    return launch_status();
}

int idsp_atan2_i32(const int32_t *xy, int32_t *out, size_t n, void *stream)
{
    if (n && (!xy || !out)) return fail(IDSP_EINVAL, "xy or out is NULL");
    if (n == 0) return IDSP_OK;
    const bool vec = (reinterpret_cast<uintptr_t>(xy) | reinterpret_cast<uintptr_t>(out)) % 16 == 0;
    size_t blocks = (n / (vec ? 4 : 1) + 255) / 256 + 1;
    if (blocks > 8192) blocks = 8192;
    note_kernel("atan2_kernel");
    hipLaunchKernelGGL(atan2_kernel, dim3(unsigned(blocks)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const Cplx *>(xy), out, n, vec);
    return launch_status();
}

int idsp_dds_i32(void *state, int32_t *out, size_t lanes, size_t frames, int layout, void *stream)
{
    if (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR) return fail(IDSP_EINVAL, "bad layout %d", layout);
    if (lanes && (!state || (frames && !out))) return fail(IDSP_EINVAL, "state or out is NULL");
    if (lanes == 0) return IDSP_OK;
    static const bool no_circle = diag_env("IDSP_DDS_NO_CIRCLE") != nullptr;
    const bool circle = !no_circle && layout == IDSP_FRAME_MAJOR && frames >= 256;
    if (lanes <= kSplitMaxLanes) {
        DdsSplitProc::Params ps{0};
        if (circle) {
            DdsSplitProcT<true>::Params pc{0};
            return launch_stream<DdsSplitProcT<true>>(pc, state, static_cast<const int32_t *>(nullptr), out, 2 * lanes, frames, layout, as_stream(stream));
        }
        return launch_stream<DdsSplitProc>(ps, state, static_cast<const int32_t *>(nullptr), out, 2 * lanes, frames, layout,
                                           as_stream(stream));
    }
    // one thread per lane (above kSplitMaxLanes) keeps the 512-byte table: 65536 lanes x 4096 run 0.36-0.37 ms with it and 0.46 with the
    // full-circle table, 24- or 16-byte entries alike (profiles/r03_perf_dds_circle.jsonl, r03_perf_dds_one_circle.jsonl)
    DdsProc::Params p{0};
    return launch_stream<DdsProc>(p, state, static_cast<const int32_t *>(nullptr), reinterpret_cast<Cplx *>(out), lanes,
                                  frames, layout, as_stream(stream));
}

size_t idsp_lockin_state_words(const idsp_lockin_i32 *cfg)
{
    if (lockin_cfg_check(cfg)) return 0;
    return size_t(2 + 2 * cfg->cascade * cfg->order * 2);
}

int idsp_lockin_i32_process(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes,
                            size_t frames, int layout, void *stream)
{
    int rc = lockin_cfg_check(cfg);
    if (rc) return rc;
    if ((rc = check_stream_args(cfg, 1, state, x, y, lanes, frames, layout))) return rc;
    if (lanes == 0) return IDSP_OK;
    if (const int waves = lockin_waves_for(x, y, lanes, frames, layout, false))
        return lockin_waves_iq(cfg, state, x, y, lanes, frames, layout, waves, as_stream(stream));
    if (const size_t body = lockin_lm_body(x, y, frames, layout)) {
        if (const int waves = lockin_waves_for(x, y, lanes, body, layout, false)) {
            if ((rc = lockin_waves_iq(cfg, state, x, y, lanes, body, layout, waves, as_stream(stream), frames))) return rc;
            rc = lockin_stream_iq(cfg, state, x + body, y + 2 * body, lanes, frames - body, layout, false, as_stream(stream), frames);
            if (rc == IDSP_OK) note_kernel("lockin_waves_kernel + stream kernel (last frames % 16)");
            return rc;
        }
    }
    // too few lanes to give every SIMD a wave: put the I and Q arms on separate threads (both layouts)
    return lockin_stream_iq(cfg, state, x, y, lanes, frames, layout, lanes <= kSplitMaxLanes, as_stream(stream));
}

int idsp_lockin_i32_arg(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                        int layout, void *stream)
{
    int rc = lockin_cfg_check(cfg);
    if (rc) return rc;
    if ((rc = check_stream_args(cfg, 1, state, x, y, lanes, frames, layout))) return rc;
    if (lanes == 0) return IDSP_OK;
    if (const int waves = lockin_waves_for(x, y, lanes, frames, layout, true, cfg->order * cfg->cascade))
        return lockin_waves_arg(cfg, state, x, y, lanes, frames, layout, waves, as_stream(stream));
    if (const size_t body = lockin_lm_body(x, y, frames, layout)) {
        if (const int waves = lockin_waves_for(x, y, lanes, body, layout, true, cfg->order * cfg->cascade)) {
            if ((rc = lockin_waves_arg(cfg, state, x, y, lanes, body, layout, waves, as_stream(stream), frames))) return rc;
            rc = lockin_stream_arg(cfg, state, x + body, y + body, lanes, frames - body, layout, as_stream(stream), frames);
            if (rc == IDSP_OK) note_kernel("lockin_waves_kernel + stream kernel (last frames % 16)");
            return rc;
        }
    }
    return lockin_stream_arg(cfg, state, x, y, lanes, frames, layout, as_stream(stream));
}

int idsp_lockin_i32_norm_sqr(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int64_t *y, size_t lanes,
                             size_t frames, int layout, void *stream)
{
    int rc = lockin_cfg_check(cfg);
    if (rc) return rc;
    if ((rc = check_stream_args(cfg, 1, state, x, y, lanes, frames, layout))) return rc;
    if (lanes == 0) return IDSP_OK;
    if (const int waves = lockin_waves_for(x, y, lanes, frames, layout, false))
        return lockin_waves_norm_sqr(cfg, state, x, y, lanes, frames, layout, waves, as_stream(stream));
    if (const size_t body = lockin_lm_body(x, y, frames, layout)) {
        if (const int waves = lockin_waves_for(x, y, lanes, body, layout, false)) {
            if ((rc = lockin_waves_norm_sqr(cfg, state, x, y, lanes, body, layout, waves, as_stream(stream), frames))) return rc;
            rc = lockin_stream_norm_sqr(cfg, state, x + body, y + body, lanes, frames - body, layout, as_stream(stream), frames);
            if (rc == IDSP_OK) note_kernel("lockin_waves_kernel + stream kernel (last frames % 16)");
            return rc;
        }
    }
    return lockin_stream_norm_sqr(cfg, state, x, y, lanes, frames, layout, as_stream(stream));
}

int idsp_fm_disc_i32(const idsp_fm_disc *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                     int layout, void *stream)
{
    int rc = check_stream_args(cfg, 1, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if (cfg->deemph.frac < 0 || cfg->deemph.frac > 31) return fail(IDSP_EINVAL, "deemph frac = %d not in 0..31", cfg->deemph.frac);
    if (lanes == 0 || frames == 0) return IDSP_OK;
    FmDiscProc::Params p;
    p.carrier = cfg->carrier;
    for (int i = 0; i < 5; i++) p.sec.ba[i] = cfg->deemph.ba[i];
    p.sec.frac = cfg->deemph.frac;
    p.sec.u = 0, p.sec.mn = INT32_MIN, p.sec.mx = INT32_MAX;
    // FRAME_MAJOR: the discriminator on four front waves, the deemphasis biquad on a fifth (IDSP_DIAG=1 IDSP_FM_DISC_WAVES=0: the
    // one-thread-per-lane stream kernel; = 3: three front waves)
    // Up to 81920 lanes: beyond, the stream kernel has two waves per SIMD itself and fewer instructions per sample (no LDS hand-over,
    // no second read of the previous row): 131072 lanes 1.40 ms against 1.52 (profiles/r03_perf_fm_disc.jsonl)
    static const size_t forced = diag_size("IDSP_FM_DISC_WAVES", ~size_t(0));
    const size_t nf = forced != ~size_t(0) ? forced : lanes <= 81920 ? 4 : 0;
    if (layout == IDSP_FRAME_MAJOR && nf && frames >= 8) {
        const unsigned grid = unsigned((lanes + kWave - 1) / kWave);
        if (nf == 3) {
            note_kernel("fm_disc_waves_kernel<3>");
            hipLaunchKernelGGL((fm_disc_waves_kernel<3>), dim3(grid), dim3(kWave * 4), 0, as_stream(stream), p, static_cast<uint32_t *>(state),
                               reinterpret_cast<const cplx_bits *>(x), y, lanes, frames, lanes, lanes);
        } else {
            note_kernel("fm_disc_waves_kernel<4>");
            hipLaunchKernelGGL((fm_disc_waves_kernel<4>), dim3(grid), dim3(kWave * 5), 0, as_stream(stream), p, static_cast<uint32_t *>(state),
                               reinterpret_cast<const cplx_bits *>(x), y, lanes, frames, lanes, lanes);
        }
        return launch_status();
    }
    // LANE_MAJOR, whole 32-frame tiles on 16-byte aligned rows: the line-wise role kernel (IDSP_DIAG=1 IDSP_FM_DISC_LM_WAVES=0: never)
    static const bool no_lm_waves = diag_size("IDSP_FM_DISC_LM_WAVES", 1) == 0;
    // Rows of any multiple of 16 frames from 32 up: the whole tiles here, the last 16 frames of a row that is not whole tiles on the tile kernel
    // behind it (same stream, rows at the call's pitch, the state carries over as between two calls).  Multiples of 16 keep the input lines on the
    // 128-byte grid; rows off it lose more to straddled lines than the role waves win (65536 lanes x 4100 / 4104 / 4124 frames: 1.52 / 1.37 / 1.51 ms
    // against 1.29 on the tile kernel; 4112 frames: 1.01).
    if (layout == IDSP_LANE_MAJOR && !no_lm_waves && frames >= 32 && frames % 16 == 0 && lanes < (size_t(1) << 28) &&
        (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 == 0) {
        const size_t body = frames - frames % 32, tail = frames - body;
        note_kernel("fm_disc_waves_lm_kernel");
        // Start-up stagger (lockin_waves.h, "lanes in phase"): arithmetic alone 0.535 ms at 65536 lanes x 4096 frames, with the requests
        // 0.60, with the stores 0.555 — and with both 0.92: every lane reads and writes the same offset of its rows at the same time.
        // Four groups of CUs 12 us apart: 0.926 -> 0.834 ms (6 us 0.884, 24 us 0.827, 48 us 0.97; groups per XCD or per
        // workgroup pair nothing; tools/exp_fm_disc_skew.sh).  IDSP_DIAG=1 IDSP_FMD_SKEW / _SHIFT / _MOD override.
        static const size_t sk_forced = diag_size("IDSP_FMD_SKEW", ~size_t(0));
        static const unsigned sk_shift = unsigned(diag_size("IDSP_FMD_SKEW_SHIFT", 4)) & 31u, sk_mod = unsigned(diag_size("IDSP_FMD_SKEW_MOD", 4));  // shift 0..31
        const unsigned grid = unsigned((lanes + kWave - 1) / kWave);
        const unsigned sk_ticks = sk_forced != ~size_t(0) ? unsigned(sk_forced) : grid >= thr::kStaggerMinWorkgroups && frames >= thr::kStaggerMinFrames && frames <= thr::kStaggerMaxFrames && stagger_tuned_device() ? thr::kFmDiscStaggerTicks : 0u;
        hipLaunchKernelGGL(fm_disc_waves_lm_kernel, dim3(grid), dim3(kWave * 5), 0, as_stream(stream), p,
                           static_cast<uint32_t *>(state), reinterpret_cast<const cplx_bits *>(x), y, lanes, body, frames, sk_ticks, sk_shift, sk_mod ? sk_mod : 1u);
        if (int rc = launch_status()) return rc;
        if (tail == 0) return IDSP_OK;
        rc = launch_stream<FmDiscProc>(p, state, reinterpret_cast<const cplx_bits *>(x) + body, y + body, lanes, tail, layout, as_stream(stream), Pitch{frames, frames});
        if (rc == IDSP_OK) note_kernel("fm_disc_waves_lm_kernel + stream_lane_major (last frames % 32)");
        return rc;
    }
    return launch_stream<FmDiscProc>(p, state, reinterpret_cast<const cplx_bits *>(x), y, lanes, frames, layout, as_stream(stream));
}

int idsp_lowpass_i32(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes,
                     size_t frames, int layout, void *stream)
{
    int rc = lockin_cfg_check(cfg);
    if (rc) return rc;
    if ((rc = check_stream_args(cfg, 1, state, x, y, lanes, frames, layout))) return rc;
    if (lanes == 0) return IDSP_OK;
    return lowpass_stream(cfg, state, x, y, lanes, frames, layout, as_stream(stream));
}

}  // extern "C"
