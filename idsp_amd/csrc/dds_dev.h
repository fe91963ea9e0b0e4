// dds_dev.h — device helpers shared by the cossin / DDS / lock-in translation units: the cossin and atan2
// tables, `cossin()` (src/cossin.rs:14-67), `atan2()` (src/atan2.rs:6-82), `Lowpass<N>` (src/lowpass.rs:47-78) and
// the `[Lowpass<N>; K]` bank.  Everything sits in an anonymous namespace: each translation unit gets its own copy of
// the constant tables (no relocatable device code in this build).
#pragma once
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
        const char *e = diag_env("IDSP_SPLIT_MAX_LANES");
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

// (Round 3 also had `cossin_wide`: the same function on a pre-shifted table {c << 14, s << 15, c << 8, s << 9} stored 8 times against
// bank conflicts, both interpolation products one v_mul_hi_i32, branch-free unmap — 23 VALU + 1 LDS.  Superseded by the form below and
// removed; tools/ubench_cossin.hip keeps it as V6 with its timing.)
// The same function once more, with the whole octant logic in the table (round 3, later; tools/ubench_cossin.hip compares it
// with the two above on the GPU, tests/test_gpu_* through every caller): 9 VALU + 1 LDS instruction per evaluation where
// round 3's pre-shifted-table form took 23 + 1.
// cossin.rs:50-66 makes each output component one of c', s', -c', -s' with c' = (c << 14) - ((s * dphi) >> 7) and
// s' = (s << 15) + ((c * dphi) >> 8), which of them being a function of the top three phase bits alone.  Every one of the
// four is  A + floor(P / 2^32)  or  A - floor(P / 2^32)  with P = B * (dphi << 16), and -floor(P / 2^32) = floor((-P + 2^32 - 1) / 2^32),
// so each is the HIGH WORD of ONE v_mad_i64_i32:  hi32(B' * d + (A' << 32 | L)),  B' = +-B, A' = +-A, and L any low word that
// carries exactly when the term is subtracted and P is not a whole multiple of 2^32.  With the product written as
// Bh * d17 (Bh = B' / 2: B is (s << 9) or (c << 8), even; d17 = dphi << 17) P is a whole multiple of 2^24 and |Bh| < 2^24, so
// L = Bh ITSELF does it: added term, 0 < Bh < 2^24 never carries; subtracted term, Bh as an unsigned word lies in
// [2^32 - 2^24, 2^32) and carries iff P mod 2^32 >= 2^24 iff P mod 2^32 != 0.  (Wrapping exactly like the reference's `wrapping_neg`.)
// The multiply-add reads Bh twice — as the multiplier and as the low half of the addend pair {Bh, A'} — so a table entry is
// {Bh_re, A_re, Bh_im, A_im}: 16 bytes, one ds_read_b128 whose result registers ARE the two addend pairs.  The table is indexed
// by the top TEN phase bits (octant and table index as they stand: the index reversal `phase = !phase` of odd octants,
// cossin.rs:27-29, is folded in): 1024 x 16 bytes.  What is left on the VALU is dphi (the reflection of the low bits in odd
// octants, the multiply by PI/4 — doubled, so that the truncation mask leaves dphi << 17 —), the entry address and the two
// multiply-adds.  Random entries do meet in banks; see the callers for where that matters.
// (First version, same round: {K, A} pairs with K = 0 / 0xffffffff and B beside them, 24-byte entries, two LDS reads.)
constexpr int kCosCircleEntries = 1 << (kCossinDepth + 3);
constexpr int kCosCircleWords = kCosCircleEntries * 4;  // 16 KiB
__device__ __forceinline__ void fill_cossin_circle(uint32_t *sh, int tid, int nthreads)
{
    for (int e = tid; e < kCosCircleEntries; e += nthreads) {
        const uint32_t x29 = (e >> kCossinDepth) & 1, x30 = (e >> (kCossinDepth + 1)) & 1, x31 = (e >> (kCossinDepth + 2)) & 1;
        const int raw = e & ((1 << kCossinDepth) - 1);
        const uint32_t lookup = d_cossin_table[x29 ? (1 << kCossinDepth) - 1 - raw : raw];
        const uint32_t c = (lookup & 0xffffu) + (1u << 16), s = lookup >> 16;
        // octant ^= octant >> 1 (cossin.rs:59): bit 29 swaps, bit 30 negates cos, bit 31 negates sin — after the swap
        const uint32_t sw = x29 ^ x30, neg[2] = {x30 ^ x31, x31};
        uint32_t *o = sh + e * 4;
#pragma unroll
        for (int comp = 0; comp < 2; comp++) {  // 0: re, 1: im
            const bool is_sin = (comp == 1) != (sw != 0);
            const uint32_t a = is_sin ? s << 15 : c << 14, bh = is_sin ? c << 7 : s << 8;
            const bool sub = is_sin == (neg[comp] != 0);  // c' subtracts its term, s' adds it; negation flips that
            o[2 * comp] = sub ? 0u - bh : bh;
            o[2 * comp + 1] = neg[comp] ? 0u - a : a;
        }
    }
}
// split in two so that a caller with several phases in hand can issue all table reads before the first multiply-add (the
// compiler keeps each evaluation's read behind the previous evaluation's s_waitcnt otherwise)
typedef uint32_t CosCircleEntry __attribute__((ext_vector_type(4)));  // {Bh_re, A_re, Bh_im, A_im}
__device__ __forceinline__ CosCircleEntry cossin_circle_fetch(uint32_t x, const uint32_t *tab)
{
    static_assert(kCossinDepth == 7, "entry byte offset = phase bits 22..31 << 4");
    return *reinterpret_cast<const CosCircleEntry *>(reinterpret_cast<const char *>(tab) + ((x >> 18) & 0x3ff0u));
}
__device__ __forceinline__ Cplx cossin_circle_finish(uint32_t x, const CosCircleEntry &e)
{
    const uint32_t xx = x ^ uint32_t(__builtin_amdgcn_sbfe(int32_t(x), 29, 1));  // low bits reflected in odd octants
    const int32_t t2 = int32_t(__builtin_amdgcn_ubfe(xx, 7, 15)) * (2 * 51471) - 16384 * (2 * 51471);  // 2 (p - 2^14) PI4
    const int64_t d17 = int64_t(int32_t(uint32_t(t2) & 0xfffe0000u));                                   // dphi << 17
    const uint64_t re = uint64_t(int64_t(int32_t(e.x)) * d17) + ((uint64_t(e.y) << 32) | e.x);
    const uint64_t im = uint64_t(int64_t(int32_t(e.z)) * d17) + ((uint64_t(e.w) << 32) | e.z);
    return Cplx{int32_t(uint32_t(re >> 32)), int32_t(uint32_t(im >> 32))};
}
__device__ __forceinline__ Cplx cossin_circle(uint32_t x, const uint32_t *tab) { return cossin_circle_finish(x, cossin_circle_fetch(x, tab)); }

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
    using Params = LpParams;
    static constexpr int kArmWords = 2 * N * K;  // state words of one arm
    static constexpr bool kSixWaves = true;      // the six-wave form of lockin_waves_kernel is instantiated for this bank
    static constexpr bool kExtLo = false;        // phase form: `Accu` -> cossin inside the kernel
    static const char *name() { return nullptr; }
    static __device__ __forceinline__ int32_t mix(int32_t lo, int32_t x) { return __mulhi(lo, x); }  // src/lockin.rs:34-37
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

// src/atan2.rs:6-82, all integer.  tab[0..16) = reciprocal bases, tab[16..32) = slopes.
__device__ __forceinline__ uint32_t mul_q31(uint32_t x, uint32_t y) { return uint32_t((uint64_t(x) * uint64_t(y)) >> 31); }

// `i32::saturating_neg` as one instruction (v_sub_i32 ... clamp)
__device__ __forceinline__ int32_t sat_neg(int32_t v) { return __builtin_elementwise_sub_sat(int32_t(0), v); }
// leading zero count with the hardware's defined result for 0 (0xffffffff: a shift by it uses the low five bits)
__device__ __forceinline__ uint32_t ffbh_u32(uint32_t v)
{
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

__device__ __forceinline__ int32_t atan2_dev(int32_t y, int32_t x, const uint32_t *tab)
{
    // octant unmap mask (src/atan2.rs:66-82): y < 0 -> ^ !0, x < 0 -> ^ i32::MAX, y > x -> ^ (i32::MAX >> 1); |v| with
    // saturation is max(v, saturating_neg(v)).  Branch-free: 13 instructions where the three `if`s compiled to 20 with 64-bit
    // compares on the halves of the products they came from (round 3, ISA of fm_disc_waves_kernel).
    uint32_t k = uint32_t(y >> 31) ^ (uint32_t(x >> 31) >> 1);
    y = y > sat_neg(y) ? y : sat_neg(y);
    x = x > sat_neg(x) ? x : sat_neg(x);
    if (y > x) k ^= 0x3fffffffu;
    const int32_t lo = y < x ? y : x, hi = y < x ? x : y;
    // divi(lo, hi), lo <= hi: normalise hi to [1, 2) in Q1.31, LUT reciprocal seed + one Newton step.  hi == 0 (then lo == 0) needs
    // no branch: the shift count is 31, every operand 0 and q = 0 as in the reference's early return.
    uint32_t q;
    {
        const uint32_t shift = ffbh_u32(uint32_t(hi)) & 31u;
        const uint32_t yn = uint32_t(lo) << shift, xn = uint32_t(hi) << shift;
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

inline int lockin_cfg_check(const idsp_lockin_i32 *c)
{
    if (!c) return fail(IDSP_EINVAL, "cfg is NULL");
    if (c->order != 1 && c->order != 2) return fail(IDSP_EINVAL, "Lowpass order %d not in {1,2} (src/lowpass.rs:75)", c->order);
    if (c->cascade < 1 || c->cascade > IDSP_LOCKIN_MAX_CASCADE) return fail(IDSP_EINVAL, "cascade %d not in 1..4", c->cascade);
    return IDSP_OK;
}

inline LpParams lp_params(const idsp_lockin_i32 *c)
{
    LpParams p;
    for (int i = 0; i < IDSP_LOCKIN_MAX_CASCADE; i++) {
        p.k[i][0] = c->k[i][0];
        p.k[i][1] = c->k[i][1];
    }
    return p;
}

}  // namespace
}  // namespace idsp
