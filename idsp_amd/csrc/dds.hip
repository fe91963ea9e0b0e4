// dds.hip — cossin(), the Accu phase-accumulator DDS, Lowpass<N> and the
// Lockin mixer over many lanes (reference: src/cossin.rs, src/accu.rs,
// src/lowpass.rs, src/lockin.rs, src/complex.rs).  All integer, bit-exact.
//
// cossin's 128-entry LUT (512 B) is staged once per workgroup into LDS from
// the compile-time table; per-lane gathers then cost an LDS read instead of a
// divergent constant-memory fetch.  Everything after the gather is ~40 VALU
// ops in registers, fused in the lock-in kernel with the mixer and the
// lowpass cascade so a phase never touches memory.
#include "atan2_table.h"
#include "cossin_table.h"
#include "biquad_sections.h"
#include "lane_stream.h"

namespace idsp {
namespace {

__device__ const uint32_t d_cossin_table[1 << kCossinDepth] = {
#define T8(i) kCossinTable[i], kCossinTable[i + 1], kCossinTable[i + 2], kCossinTable[i + 3], \
              kCossinTable[i + 4], kCossinTable[i + 5], kCossinTable[i + 6], kCossinTable[i + 7]
    T8(0),  T8(8),  T8(16), T8(24), T8(32), T8(40), T8(48),  T8(56),
    T8(64), T8(72), T8(80), T8(88), T8(96), T8(104), T8(112), T8(120)
#undef T8
};

// Largest lane count that still runs the I and Q arms on two threads (IDSP_SPLIT_MAX_LANES overrides).
inline size_t split_max_lanes()
{
    static const size_t v = [] {
        const char *e = getenv("IDSP_SPLIT_MAX_LANES");
        return e ? size_t(strtoull(e, nullptr, 10)) : size_t(40960);
    }();
    return v;
}
#define kSplitMaxLanes split_max_lanes()

struct Cplx {
    int32_t re, im;
};
static_assert(sizeof(Cplx) == 8, "Complex<i32> is [re, im]");

// src/cossin.rs:14-67
__device__ __forceinline__ Cplx cossin_dev(int32_t phase_in, const uint32_t *lut)
{
    constexpr int kAlign = 32 - 16 - 1;  // ALIGN_MSB
    uint32_t octant = uint32_t(phase_in);
    uint32_t ph = uint32_t(phase_in);
    if (octant & (1u << 29)) ph = ~ph;  // phase = pi/4 - phase
    ph = (ph << 3) >> (32 - kCossinDepth - kAlign);
    const uint32_t lookup = lut[ph >> kAlign];
    int32_t p = int32_t(ph & ((1u << kAlign) - 1u)) - (1 << (kAlign - 1));
    constexpr int32_t kPi4 = 51471;  // (FRAC_PI_4 * 65536.0) as i32
    const int32_t dphi = (p * kPi4) >> 16;
    int32_t c = int32_t(lookup & 0xffffu) + (1 << 16);
    int32_t s = int32_t(lookup >> 16);
    const int32_t dcos = (s * dphi) >> kCossinDepth;
    const int32_t dsin = (c * dphi) >> (kCossinDepth + 1);
    c = int32_t(uint32_t(c) << (kAlign - 1)) - dcos;
    s = int32_t(uint32_t(s) << kAlign) + dsin;
    octant ^= octant >> 1;
    if (octant & (1u << 29)) {
        const int32_t t = c;
        c = s;
        s = t;
    }
    if (octant & (1u << 30)) c = int32_t(0u - uint32_t(c));
    if (octant & (1u << 31)) s = int32_t(0u - uint32_t(s));
    return Cplx{c, s};
}

__device__ __forceinline__ void fill_cossin(uint32_t *sh, int tid, int nthreads)
{
    for (int i = tid; i < (1 << kCossinDepth); i += nthreads) sh[i] = d_cossin_table[i];
}

// src/lowpass.rs:47-78; all i64 arithmetic wraps (the library is built with
// -fwrapv, so plain signed arithmetic has Rust release semantics and the two
// products map onto v_mad_i64_i32).
template <int N>
__device__ __forceinline__ int32_t lowpass_step(const int32_t (&k)[2], int64_t (&s)[N], int32_t x)
{
    int64_t d = int64_t(__builtin_elementwise_sub_sat(x, int32_t(s[0] >> 32))) * int64_t(k[0]);
    int32_t y;
    if constexpr (N == 1) {
        s[0] += d;
        y = int32_t(s[0] >> 32);
        s[0] += d;
    } else {
        d += int64_t(int32_t(s[1] >> 32)) * int64_t(k[1]);
        s[1] += d;
        s[0] += s[1];
        y = int32_t(s[0] >> 32);
        s[0] += s[1];
        s[1] += d;
    }
    return y;
}

struct LpParams {
    int32_t k[IDSP_LOCKIN_MAX_CASCADE][2];
};

template <int N, int K>
struct LpBank {
    int64_t s[K][N];
    __device__ __forceinline__ void load(const uint32_t *st, size_t lanes, size_t lane, int word0)
    {
#pragma unroll
        for (int c = 0; c < K; c++)
#pragma unroll
            for (int j = 0; j < N; j++) {
                const size_t w = size_t(word0 + (c * N + j) * 2);
                s[c][j] = int64_t(uint64_t(st[w * lanes + lane]) | (uint64_t(st[(w + 1) * lanes + lane]) << 32));
            }
    }
    __device__ __forceinline__ void store(uint32_t *st, size_t lanes, size_t lane, int word0) const
    {
#pragma unroll
        for (int c = 0; c < K; c++)
#pragma unroll
            for (int j = 0; j < N; j++) {
                const size_t w = size_t(word0 + (c * N + j) * 2);
                st[w * lanes + lane] = uint32_t(uint64_t(s[c][j]));
                st[(w + 1) * lanes + lane] = uint32_t(uint64_t(s[c][j]) >> 32);
            }
    }
    // `[Lowpass<N>; K]` array composition (dsp-process/src/compose.rs:84-93)
    __device__ __forceinline__ int32_t step(const LpParams &p, int32_t x)
    {
#pragma unroll
        for (int c = 0; c < K; c++) x = lowpass_step<N>(p.k[c], s[c], x);
        return x;
    }
};

// value of the other thread of an adjacent-thread pair (v_mov_b32 quad_perm:[1,0,3,2])
__device__ __forceinline__ int32_t pair_swap(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true); }

// ------------------------------------------------------------- processors
template <int N, int K>
struct LowpassProc {
    using In = int32_t;
    using Out = int32_t;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 0;
    static constexpr int IN_DIV = 1;
    static constexpr int COST = 40 * N * K;
    using Params = LpParams;
    LpBank<N, K> b;
    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t lanes, size_t lane) { b.load(st, lanes, lane, 0); }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane) { b.store(st, lanes, lane, 0); }
    __device__ __forceinline__ Out step(const Params &p, In x) { return b.step(p, x); }
};

// Accu (src/accu.rs:34-41, pre-increment) -> Complex::from_angle (src/complex.rs:237-240)
struct DdsProc {
    using In = int32_t;  // unused
    using Out = Cplx;
    static constexpr bool HAS_IN = false;
    static constexpr int LDS_WORDS = 1 << kCossinDepth;
    static constexpr int IN_DIV = 1;
    static constexpr int COST = 100;
    struct Params {
        int32_t unused;
    };
    const uint32_t *lut;
    uint32_t acc, inc;
    static __device__ __forceinline__ void fill_shared(uint32_t *sh, int tid, int n) { fill_cossin(sh, tid, n); }
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
        return cossin_dev(int32_t(acc), lut);
    }
};

// src/lockin.rs:30-39 -> :17-27.  Mixer `x * Q32<32>` =
// ((q as i64 * x as i64) >> 32) as i32 (dsp-fixedpoint/src/lib.rs:449-456).
template <int N, int K>
struct LockinProc {
    using In = int32_t;
    using Out = Cplx;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 1 << kCossinDepth;
    static constexpr int IN_DIV = 1;
    static constexpr int COST = 110 + 80 * N * K;
    using Params = LpParams;
    const uint32_t *lut;
    uint32_t acc, inc;
    LpBank<N, K> bi, bq;
    static __device__ __forceinline__ void fill_shared(uint32_t *sh, int tid, int n) { fill_cossin(sh, tid, n); }
    __device__ __forceinline__ void set_shared(const uint32_t *sh) { lut = sh; }
    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t lanes, size_t lane)
    {
        acc = st[lane];
        inc = st[lanes + lane];
        bi.load(st, lanes, lane, 2);
        bq.load(st, lanes, lane, 2 + 2 * N * K);
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane)
    {
        st[lane] = acc;
        bi.store(st, lanes, lane, 2);
        bq.store(st, lanes, lane, 2 + 2 * N * K);
    }
    static constexpr int BATCH = 4;
    using Pre = Cplx;
    __device__ __forceinline__ Pre pre(const Params &)
    {
        acc += inc;
        return cossin_dev(int32_t(acc), lut);
    }
    __device__ __forceinline__ Out step(const Params &p, In x, const Pre &lo)
    {
        const int32_t xi = __mulhi(lo.re, x);
        const int32_t xq = __mulhi(lo.im, x);
        return Cplx{bi.step(p, xi), bq.step(p, xq)};
    }
};

// I/Q arms on two adjacent threads ("virtual lanes" 2*lane + iq): used when the
// lane count alone cannot give every SIMD a wave (C4: 32768 lanes = 512 waves).
// Both threads step the same phase accumulator and evaluate cossin; each runs
// one arm of the mixer + lowpass cascade and writes one word of Complex<i32>,
// so a wave still stores 256 contiguous bytes per frame.
template <int N, int K>
struct LockinSplitProc {
    using In = int32_t;
    using Out = int32_t;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 1 << kCossinDepth;
    static constexpr int IN_DIV = 2;
    static constexpr int COST = 70 + 40 * N * K;
    using Params = LpParams;
    const uint32_t *lut;
    uint32_t acc, inc;
    bool q;
    LpBank<N, K> b;
    static __device__ __forceinline__ void fill_shared(uint32_t *sh, int tid, int n) { fill_cossin(sh, tid, n); }
    __device__ __forceinline__ void set_shared(const uint32_t *sh) { lut = sh; }
    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t vlanes, size_t vlane)
    {
        const size_t lanes = vlanes / 2, lane = vlane / 2;
        q = vlane & 1;
        acc = st[lane];
        inc = st[lanes + lane];
        b.load(st, lanes, lane, 2 + (q ? 2 * N * K : 0));
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t vlanes, size_t vlane)
    {
        const size_t lanes = vlanes / 2, lane = vlane / 2;
        if (!q) st[lane] = acc;
        b.store(st, lanes, lane, 2 + (q ? 2 * N * K : 0));
    }
    static constexpr int BATCH = 4;
    using Pre = int32_t;  // this arm's LO component
    __device__ __forceinline__ Pre pre(const Params &)
    {
        acc += inc;
        const Cplx lo = cossin_dev(int32_t(acc), lut);
        return q ? lo.im : lo.re;
    }
    // The two threads of a lane evaluate the LO of alternate frames (thread q: frames 2j + q) and
    // swap the component the partner needs with one DPP move: 16 instead of 32 cossin
    // instructions per frame and thread on a VALU-bound kernel.
    __device__ __forceinline__ void pre_batch(const Params &, Pre (&out)[BATCH])
    {
#pragma unroll
        for (int j = 0; j < BATCH / 2; j++) {
            const uint32_t ph = acc + inc * uint32_t(2 * j + 1) + (q ? inc : 0u);
            const Cplx lo = cossin_dev(int32_t(ph), lut);
            const int32_t mine = q ? lo.im : lo.re, other = q ? lo.re : lo.im;
            const int32_t recv = pair_swap(other);
            out[2 * j] = q ? recv : mine;
            out[2 * j + 1] = q ? mine : recv;
        }
        acc += inc * uint32_t(BATCH);
    }
    __device__ __forceinline__ Out step(const Params &p, In x, const Pre &lo) { return b.step(p, __mulhi(lo, x)); }
};

struct DdsSplitProc {
    using In = int32_t;  // unused
    using Out = int32_t;
    static constexpr bool HAS_IN = false;
    static constexpr int LDS_WORDS = 1 << kCossinDepth;
    static constexpr int IN_DIV = 2;
    static constexpr int COST = 100;
    struct Params {
        int32_t unused;
    };
    const uint32_t *lut;
    uint32_t acc, inc;
    bool q;
    static __device__ __forceinline__ void fill_shared(uint32_t *sh, int tid, int n) { fill_cossin(sh, tid, n); }
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
        const Cplx c = cossin_dev(int32_t(acc), lut);
        return q ? c.im : c.re;
    }
    __device__ __forceinline__ void pre_batch(const Params &, Pre (&out)[BATCH])
    {
#pragma unroll
        for (int j = 0; j < BATCH / 2; j++) {
            const uint32_t ph = acc + inc * uint32_t(2 * j + 1) + (q ? inc : 0u);
            const Cplx c = cossin_dev(int32_t(ph), lut);
            const int32_t mine = q ? c.im : c.re, other = q ? c.re : c.im;
            const int32_t recv = pair_swap(other);
            out[2 * j] = q ? recv : mine;
            out[2 * j + 1] = q ? mine : recv;
        }
        acc += inc * uint32_t(BATCH);
    }
    __device__ __forceinline__ Out step(const Params &, In, const Pre &v) { return v; }
};

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

// src/atan2.rs:6-82, all integer.  tab[0..16) = reciprocal bases, tab[16..32) = slopes.
__device__ __forceinline__ uint32_t mul_q31(uint32_t x, uint32_t y) { return uint32_t((uint64_t(x) * uint64_t(y)) >> 31); }

__device__ __forceinline__ int32_t atan2_dev(int32_t y, int32_t x, const uint32_t *tab)
{
    uint32_t k = 0;
    if (y < 0) {
        y = y == INT32_MIN ? INT32_MAX : -y;  // saturating_neg
        k ^= 0xffffffffu;
    }
    if (x < 0) {
        x = x == INT32_MIN ? INT32_MAX : -x;
        k ^= 0x7fffffffu;
    }
    if (y > x) {
        const int32_t t = y;
        y = x;
        x = t;
        k ^= 0x3fffffffu;
    }
    // divi(y, x), y <= x: normalise x to [1, 2) in Q1.31, LUT reciprocal seed + one Newton step
    uint32_t q = 0;
    if (x != 0) {
        const int shift = __builtin_clz(uint32_t(x));
        const uint32_t yn = uint32_t(y) << shift, xn = uint32_t(x) << shift;
        constexpr int kFrac = 31 - kAtan2DiviDepth;
        const uint32_t rem = xn & ((1u << kFrac) - 1u);
        const uint32_t idx = (xn << 1) >> (1 + kFrac);
        const uint32_t step = uint32_t((int64_t(int32_t(tab[16 + idx])) * int64_t(rem)) >> kFrac);
        const uint32_t r0 = tab[idx] + step;
        q = mul_q31(yn, mul_q31(r0, 0u - mul_q31(xn, r0)));
    }
    // atani(q): odd polynomial q * P(q^2 / 4), Horner in Q32<32> from the highest coefficient
    const int32_t x2 = int32_t((int64_t(q) * int64_t(q)) >> 32);
    int32_t r = 0;
    constexpr int32_t kAtani[6] = {0x0517c2cd, -0x06c6496b, 0x0fbdb021, -0x25b32e0a, 0x43b34c81, -0x3bc823dd};
#pragma unroll
    for (int i = 5; i >= 0; i--) r = int32_t(uint32_t(int32_t((int64_t(r) * int64_t(x2)) >> 32)) + uint32_t(kAtani[i]));
    const uint32_t a = uint32_t((int64_t(r) * int64_t(q)) >> 28);
    return int32_t(a ^ k);
}

__device__ const uint32_t d_atan2_table[32] = {
    kAtan2Base[0], kAtan2Base[1], kAtan2Base[2], kAtan2Base[3], kAtan2Base[4], kAtan2Base[5], kAtan2Base[6], kAtan2Base[7],
    kAtan2Base[8], kAtan2Base[9], kAtan2Base[10], kAtan2Base[11], kAtan2Base[12], kAtan2Base[13], kAtan2Base[14], kAtan2Base[15],
    uint32_t(kAtan2Slope[0]), uint32_t(kAtan2Slope[1]), uint32_t(kAtan2Slope[2]), uint32_t(kAtan2Slope[3]),
    uint32_t(kAtan2Slope[4]), uint32_t(kAtan2Slope[5]), uint32_t(kAtan2Slope[6]), uint32_t(kAtan2Slope[7]),
    uint32_t(kAtan2Slope[8]), uint32_t(kAtan2Slope[9]), uint32_t(kAtan2Slope[10]), uint32_t(kAtan2Slope[11]),
    uint32_t(kAtan2Slope[12]), uint32_t(kAtan2Slope[13]), uint32_t(kAtan2Slope[14]), uint32_t(kAtan2Slope[15])};

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

// Lock-in with the polar read-out fused on the same thread: `Lockin::process(..)` (src/lockin.rs:30-39)
// followed by `Complex::<i32>::arg()` (src/complex.rs:254-256, MODE 0, i32) or `norm_sqr()`
// (src/complex.rs:214-217, MODE 1, i64 with the wrapping sum of a release build).  Saves the 8 byte/sample
// Complex<i32> round trip through HBM of `lockin` + `atan2`.
template <int N, int K, int MODE>
struct LockinPolarProc {
    using In = int32_t;
    using Out = std::conditional_t<MODE == 0, int32_t, int64_t>;
    static constexpr bool HAS_IN = true;
    static constexpr int kLut = 1 << kCossinDepth;
    static constexpr int LDS_WORDS = kLut + (MODE == 0 ? 32 : 0);  // cossin table, atan2 reciprocal table
    static constexpr int IN_DIV = 1;
    static constexpr int COST = 110 + 80 * N * K + (MODE == 0 ? 80 : 10);
    using Params = LpParams;
    const uint32_t *lut;
    uint32_t acc, inc;
    LpBank<N, K> bi, bq;
    static __device__ __forceinline__ void fill_shared(uint32_t *sh, int tid, int n)
    {
        fill_cossin(sh, tid, n);
        if constexpr (MODE == 0)
            for (int i = tid; i < 32; i += n) sh[kLut + i] = d_atan2_table[i];
    }
    __device__ __forceinline__ void set_shared(const uint32_t *sh) { lut = sh; }
    __device__ __forceinline__ void load(const Params &, const uint32_t *st, size_t lanes, size_t lane)
    {
        acc = st[lane];
        inc = st[lanes + lane];
        bi.load(st, lanes, lane, 2);
        bq.load(st, lanes, lane, 2 + 2 * N * K);
    }
    __device__ __forceinline__ void store(const Params &, uint32_t *st, size_t lanes, size_t lane)
    {
        st[lane] = acc;
        bi.store(st, lanes, lane, 2);
        bq.store(st, lanes, lane, 2 + 2 * N * K);
    }
    static constexpr int BATCH = 4;
    using Pre = Cplx;
    __device__ __forceinline__ Pre pre(const Params &)
    {
        acc += inc;
        return cossin_dev(int32_t(acc), lut);
    }
    __device__ __forceinline__ Out step(const Params &p, In x, const Pre &lo)
    {
        const int32_t re = bi.step(p, __mulhi(lo.re, x));
        const int32_t im = bq.step(p, __mulhi(lo.im, x));
        if constexpr (MODE == 0)
            return atan2_dev(im, re, lut + kLut);
        else
            return int64_t(uint64_t(int64_t(re) * re) + uint64_t(int64_t(im) * im));
    }
};
template <int N, int K>
using LockinArgProc = LockinPolarProc<N, K, 0>;
template <int N, int K>
using LockinNormSqrProc = LockinPolarProc<N, K, 1>;

// FrameMajor lock-in -> arg with the work of one lane spread over four or six waves.  A single thread per lane runs
// ~150 VALU instructions per frame (cossin ~20, two arms ~30 each, atan2 ~70), many of them multi-pass 64-bit
// operations, on one wave per SIMD at the C4 lane counts, where a SIMD issues an instruction every 6-10 cycles
// instead of every 3-5.  Here a workgroup is 64 lanes x 4 (6) waves: wave 0 runs the I arm, wave 1 the Q arm,
// the other 2 (4) the LO (cossin) and the atan2 of every 2nd (4th) frame; cos/sin and the arm outputs travel through
// LDS and every store instruction still writes one contiguous 256-byte row.  The stages are software-pipelined
// over batches of kPairB frames -- arms of batch n beside LO of batch n + 1 and atan2 of batch n - 1 in one
// barrier interval, double-buffered in LDS.
constexpr int kPairB = 8;

// LM: LaneMajor rows (whole batches and 16-byte aligned rows only): an arm thread reads its lane's 8 samples of a
// batch as two 16-byte vectors, a polar thread writes its 4 (2) consecutive phases as one 16 (8) byte vector.
template <int N, int K, int kPairWaves, bool LM>  // kPairWaves = 2 arm waves + 2 or 4 polar waves
__global__ __launch_bounds__(kPairWaves * kWave) void lockin_arg_pair_fm(const LpParams prm, uint32_t *st, const int32_t *x,
                                                                         int32_t *y, const size_t lanes, const size_t frames)
{
    constexpr int B = kPairB, kLut = 1 << kCossinDepth;
    __shared__ uint32_t lut[kLut];
    __shared__ uint32_t tab[32];
    __shared__ Cplx lo[2][B][kWave];
    __shared__ int32_t arm[2][2][B][kWave];  // [buffer][I/Q][frame][lane]
    const int w = threadIdx.x / kWave, lid = threadIdx.x % kWave;
    const bool arm_wave = w < 2;  // wave-uniform role
    constexpr int P = kPairWaves - 2, C = kPairB / P;  // polar waves, each takes frames b = r * C + j, j < C, of a batch
    static_assert(kPairB % P == 0 && kPairB == 8, "batch splits evenly over the polar waves");
    const int r = arm_wave ? w : w - 2;  // arm waves: I / Q; polar waves: frame group
    const size_t lane = size_t(blockIdx.x) * kWave + lid;
    const bool active = lane < lanes;
    const size_t la = active ? lane : lanes - 1;  // idle threads of the last workgroup shadow a valid lane, stores masked
    fill_cossin(lut, threadIdx.x, kPairWaves * kWave);
    if (threadIdx.x < 32) tab[threadIdx.x] = d_atan2_table[threadIdx.x];
    const uint32_t acc0 = st[la], inc = st[lanes + la];
    LpBank<N, K> bank;
    if (arm_wave) bank.load(st, lanes, la, 2 + (r ? 2 * N * K : 0));
    // row base pointers are wave-uniform and the lane offset is one 32-bit register
    const uint32_t lo32 = uint32_t(la), lane32 = uint32_t(lane);
    uint32_t phase = acc0;  // accumulator before the batch whose LO is produced next
    int32_t xn[B];
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    typedef int32_t i32xc __attribute__((ext_vector_type(C)));
    auto fetch = [&](size_t f0, auto full) {
        if constexpr (LM) {
            const i32x4 *row = reinterpret_cast<const i32x4 *>(x + la * frames + f0);
            const i32x4 a = row[0], c = row[1];
            xn[0] = a.x, xn[1] = a.y, xn[2] = a.z, xn[3] = a.w, xn[4] = c.x, xn[5] = c.y, xn[6] = c.z, xn[7] = c.w;
        } else {
#pragma unroll
            for (int b = 0; b < B; b++) {
                const int32_t *row = x + (f0 + b) * lanes;
                xn[b] = (decltype(full)::value || f0 + b < frames) ? row[lo32] : 0;
            }
        }
    };
    auto lo_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < C; j++) {
            const int b = r * C + j;
            lo[buf][b][lid] = cossin_dev(int32_t(phase + inc * uint32_t(b + 1)), lut);
        }
        phase += inc * uint32_t(B);
    };
    auto arg_stage = [&](size_t f0, int buf, int nb, auto full) {
        if constexpr (LM) {
            i32xc v;
#pragma unroll
            for (int j = 0; j < C; j++) v[j] = atan2_dev(arm[buf][1][r * C + j][lid], arm[buf][0][r * C + j][lid], tab);
            if (active) *reinterpret_cast<i32xc *>(y + lane * frames + f0 + r * C) = v;
        } else {
#pragma unroll
            for (int j = 0; j < C; j++) {
                const int b = r * C + j;
                if ((decltype(full)::value || b < nb) && active) {
                    int32_t *row = y + (f0 + b) * lanes;
                    row[lane32] = atan2_dev(arm[buf][1][b][lid], arm[buf][0][b][lid], tab);
                }
            }
        }
    };
    auto iter = [&](size_t n, int nb, auto full, auto first) {
        const size_t f0 = n * B;
        const int buf = int(n & 1);
        if (arm_wave) {
            int32_t xv[B];
#pragma unroll
            for (int b = 0; b < B; b++) xv[b] = xn[b];
            if (f0 + 2 * B <= frames)
                fetch(f0 + B, std::true_type{});
            else if (f0 + B < frames)
                fetch(f0 + B, std::false_type{});
            const int32_t *lo_mine = reinterpret_cast<const int32_t *>(&lo[buf][0][lid]) + r;  // this arm's LO component
#pragma unroll
            for (int b = 0; b < B; b++)
                if (decltype(full)::value || b < nb) arm[buf][r][b][lid] = bank.step(prm, __mulhi(lo_mine[b * kWave * 2], xv[b]));
        } else {
            lo_stage(buf ^ 1);
            if constexpr (!decltype(first)::value) arg_stage(f0 - B, buf ^ 1, B, std::true_type{});
        }
        __syncthreads();
    };
    if (arm_wave) {
        if (frames >= size_t(B))
            fetch(0, std::true_type{});
        else
            fetch(0, std::false_type{});
    }
    __syncthreads();  // tables
    if (!arm_wave) lo_stage(0);
    __syncthreads();
    const size_t nfull = frames / B;
    const int tail = int(frames % B);
    if (nfull) {
        iter(0, B, std::true_type{}, std::true_type{});
        for (size_t n = 1; n < nfull; n++) iter(n, B, std::true_type{}, std::false_type{});
    }
    if (tail) {
        if (nfull)
            iter(nfull, tail, std::false_type{}, std::false_type{});
        else
            iter(0, tail, std::false_type{}, std::true_type{});
        if (!arm_wave) arg_stage(nfull * B, int(nfull & 1), tail, std::false_type{});
    } else if (!arm_wave) {
        arg_stage((nfull - 1) * B, int((nfull - 1) & 1), B, std::true_type{});
    }
    if (active && arm_wave) {
        if (r == 0) st[lane] = acc0 + inc * uint32_t(frames);
        bank.store(st, lanes, lane, 2 + (r ? 2 * N * K : 0));
    }
}

template <int N, int K>
int launch_lockin_arg_pair(const LpParams &p, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout,
                           hipStream_t s)
{
    const dim3 grid(unsigned((lanes + kWave - 1) / kWave));
    uint32_t *st = static_cast<uint32_t *>(state);
    // measured at 4096 frames, FrameMajor: 6 waves per 64 lanes 0.64 ms at 32768 lanes (4 waves: 0.73), 4 waves 1.07 ms at
    // 65536 (6: 1.12)
    const bool six = lanes <= kSplitMaxLanes;
    if (layout == IDSP_LANE_MAJOR) {
        if (six)
            hipLaunchKernelGGL((lockin_arg_pair_fm<N, K, 6, true>), grid, dim3(6 * kWave), 0, s, p, st, x, y, lanes, frames);
        else
            hipLaunchKernelGGL((lockin_arg_pair_fm<N, K, 4, true>), grid, dim3(4 * kWave), 0, s, p, st, x, y, lanes, frames);
    } else {
        if (six)
            hipLaunchKernelGGL((lockin_arg_pair_fm<N, K, 6, false>), grid, dim3(6 * kWave), 0, s, p, st, x, y, lanes, frames);
        else
            hipLaunchKernelGGL((lockin_arg_pair_fm<N, K, 4, false>), grid, dim3(4 * kWave), 0, s, p, st, x, y, lanes, frames);
    }
    return launch_status();
}

int lockin_cfg_check(const idsp_lockin_i32 *c)
{
    if (!c) return fail(IDSP_EINVAL, "cfg is NULL");
    if (c->order != 1 && c->order != 2) return fail(IDSP_EINVAL, "Lowpass order %d not in {1,2} (src/lowpass.rs:75)", c->order);
    if (c->cascade < 1 || c->cascade > IDSP_LOCKIN_MAX_CASCADE) return fail(IDSP_EINVAL, "cascade %d not in 1..4", c->cascade);
    return IDSP_OK;
}

LpParams lp_params(const idsp_lockin_i32 *c)
{
    LpParams p;
    for (int i = 0; i < IDSP_LOCKIN_MAX_CASCADE; i++) {
        p.k[i][0] = c->k[i][0];
        p.k[i][1] = c->k[i][1];
    }
    return p;
}

template <template <int, int> class Proc, class OutT>
int dispatch_nk(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, OutT *y, size_t lanes, size_t frames,
                int layout, hipStream_t s)
{
    const LpParams p = lp_params(cfg);
#define IDSP_CASE(N, K) \
    if (cfg->order == N && cfg->cascade == K) return launch_stream<Proc<N, K>>(p, state, x, y, lanes, frames, layout, s)
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

using namespace idsp;

extern "C" {

int idsp_cossin_i32(const int32_t *phase, int32_t *out, size_t n, void *stream)
{
    if (n && (!phase || !out)) return fail(IDSP_EINVAL, "phase or out is NULL");
    if (n == 0) return IDSP_OK;
    const bool vec = (reinterpret_cast<uintptr_t>(phase) | reinterpret_cast<uintptr_t>(out)) % 16 == 0;
    size_t blocks = (n / (vec ? 2 : 1) + 255) / 256 + 1;
    if (blocks > 8192) blocks = 8192;
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
    hipLaunchKernelGGL(atan2_kernel, dim3(unsigned(blocks)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const Cplx *>(xy), out, n, vec);
    return launch_status();
}

int idsp_dds_i32(void *state, int32_t *out, size_t lanes, size_t frames, int layout, void *stream)
{
    if (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR) return fail(IDSP_EINVAL, "bad layout %d", layout);
    if (lanes && (!state || (frames && !out))) return fail(IDSP_EINVAL, "state or out is NULL");
    if (lanes == 0) return IDSP_OK;
    if (lanes <= kSplitMaxLanes) {
        DdsSplitProc::Params ps{0};
        return launch_stream<DdsSplitProc>(ps, state, static_cast<const int32_t *>(nullptr), out, 2 * lanes, frames, layout,
                                           as_stream(stream));
    }
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
    // too few lanes to give every SIMD a wave: put the I and Q arms on separate threads (both layouts)
    if (lanes <= kSplitMaxLanes)
        return dispatch_nk<LockinSplitProc, int32_t>(cfg, state, x, y, 2 * lanes, frames, layout, as_stream(stream));
    return dispatch_nk<LockinProc, Cplx>(cfg, state, x, reinterpret_cast<Cplx *>(y), lanes, frames, layout, as_stream(stream));
}

int idsp_lockin_i32_arg(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                        int layout, void *stream)
{
    int rc = lockin_cfg_check(cfg);
    if (rc) return rc;
    if ((rc = check_stream_args(cfg, 1, state, x, y, lanes, frames, layout))) return rc;
    if (lanes == 0) return IDSP_OK;
    static const bool pair = !getenv("IDSP_LOCKIN_ARG_NO_PAIR");
    // LaneMajor takes the multi-wave kernel for whole batches on 16-byte aligned rows, the generic stream kernel otherwise
    const bool lm_ok = frames % kPairB == 0 && (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 == 0;
    if (pair && frames && (layout == IDSP_FRAME_MAJOR || lm_ok)) {
        const LpParams p = lp_params(cfg);
#define IDSP_CASE(N, K) \
    if (cfg->order == N && cfg->cascade == K) return launch_lockin_arg_pair<N, K>(p, state, x, y, lanes, frames, layout, as_stream(stream))
        IDSP_CASE(1, 1);
        IDSP_CASE(1, 2);
        IDSP_CASE(1, 3);
        IDSP_CASE(1, 4);
        IDSP_CASE(2, 1);
        IDSP_CASE(2, 2);
        IDSP_CASE(2, 3);
        IDSP_CASE(2, 4);
#undef IDSP_CASE
    }
    return dispatch_nk<LockinArgProc, int32_t>(cfg, state, x, y, lanes, frames, layout, as_stream(stream));
}

int idsp_lockin_i32_norm_sqr(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int64_t *y, size_t lanes,
                             size_t frames, int layout, void *stream)
{
    int rc = lockin_cfg_check(cfg);
    if (rc) return rc;
    if ((rc = check_stream_args(cfg, 1, state, x, y, lanes, frames, layout))) return rc;
    if (lanes == 0) return IDSP_OK;
    return dispatch_nk<LockinNormSqrProc, int64_t>(cfg, state, x, y, lanes, frames, layout, as_stream(stream));
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
    return launch_stream<FmDiscProc>(p, state, reinterpret_cast<const cplx_bits *>(x), y, lanes, frames, layout, as_stream(stream));
}

int idsp_lowpass_i32(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes,
                     size_t frames, int layout, void *stream)
{
    int rc = lockin_cfg_check(cfg);
    if (rc) return rc;
    if ((rc = check_stream_args(cfg, 1, state, x, y, lanes, frames, layout))) return rc;
    if (lanes == 0) return IDSP_OK;
    return dispatch_nk<LowpassProc, int32_t>(cfg, state, x, y, lanes, frames, layout, as_stream(stream));
}

}  // extern "C"
