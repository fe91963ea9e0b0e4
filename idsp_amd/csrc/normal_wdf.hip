// normal_wdf.hip — C-ABI entry points (include/idsp_hip.h) of the `iir::wdf::Wdf` wave-digital allpass sections (src/iir/wdf.rs).
// (`iir::normal::Normal`: normal.hip — a translation unit of its own for the parallel build.)
#include <cmath>

#include "biquad_sections.h"

namespace idsp {
namespace {

// ------------------------------------------------------------------- Wdf
constexpr int kWdfMaxSections = 4;  // sections fused per launch; longer chains run in passes

// `i32 * Q32<32>` (dsp-fixedpoint/src/lib.rs:449-456)
__device__ __forceinline__ int32_t mulq32(int32_t c, int32_t a) { return int32_t((int64_t(c) * int64_t(a)) >> 32); }

// One two-port adaptor `Tpa::adapt` (src/iir/wdf.rs:65-100) as host-precomputed integer coefficients.
// Every variant is  c = s * (x1 - x0),  p = (c * a) >> 32,  o0 = p + k00*x0 + k01*x1,  o1 = p + k10*x0 + k11*x1
// in wrapping 32-bit arithmetic (small integer multiples are exact mod 2^32):
//   A : s=+1, o0 = p - x0 + 2 x1, o1 = p + x1        D : s=-1, o0 = p + x0 - 2 x1, o1 = p - x1
//   B = B1: s=-1, o0 = p + x1, o1 = p + x0            C = C1: s=+1, o0 = p - x1,  o1 = p - x0
//   X : a=0, o0 = x1, o1 = x0                         Z : a=0, o0 = x0, o1 = x1
// (B1 / C1 only reassociate the same wrapping sums.)  No per-sample decoding of the architecture
// nibbles: a first version with a uniform switch per adaptor spent its time in scalar compares,
// branches and SGPR spills (2.9 ms instead of the 0.4 ms of a biquad at the C2 shape).
struct Adaptor {
    int32_t s, a, k00, k01, k10, k11;
};

template <int NMAX>
struct WdfParamsT {
    int32_t n[kWdfMaxSections];
    Adaptor ad[kWdfMaxSections][NMAX];
};

inline Adaptor adaptor_of(uint32_t nib, int32_t a)
{
    switch (nib) {
        case 0xA: return {1, a, -1, 2, 0, 1};
        case 0xB: case 0xE: return {-1, a, 0, 1, 1, 0};
        case 0xC: case 0xF: return {1, a, 0, -1, -1, 0};
        case 0xD: return {-1, a, 1, -2, 0, -1};
        case 0x1: return {0, 0, 0, 1, 1, 0};
        default: return {0, 0, 1, 0, 0, 1};
    }
}

// K sections padded to NMAX adaptors each.  Padding adaptors are Z on scratch state slots: the fold of
// src/iir/wdf.rs:153-169 writes adaptor i's first output into slot i-1 and ends with `*z = x`, and a Z
// adaptor at index n does exactly that final store (o0 = x -> z[n-1]); what flows on afterwards only touches
// slots >= n, which are never written back.  The loops below are therefore fully static.
template <int K, int NMAX>
struct WdfChain {
    static constexpr int MAX_U = 8;  // ~10 VALU per adaptor: keep the unrolled window inside the instruction cache
    using In = int32_t;
    using Out = int32_t;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 0;
    static constexpr int IN_DIV = 1;
    static constexpr int COST = 200;  // never the LDS-DMA path
    using Params = WdfParamsT<NMAX>;
    int32_t z[K][NMAX];

    // state words: sections in order, N words each
    __device__ __forceinline__ void load(const Params &p, const uint32_t *st, size_t lanes, size_t lane)
    {
        int w = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
#pragma unroll
            for (int i = 0; i < NMAX; i++) z[k][i] = i < p.n[k] ? int32_t(st[size_t(w + i) * lanes + lane]) : 0;
            w += p.n[k];
        }
    }
    __device__ __forceinline__ void store(const Params &p, uint32_t *st, size_t lanes, size_t lane)
    {
        int w = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
#pragma unroll
            for (int i = 0; i < NMAX; i++)
                if (i < p.n[k]) st[size_t(w + i) * lanes + lane] = uint32_t(z[k][i]);
            w += p.n[k];
        }
    }
    __device__ __forceinline__ Out step(const Params &p, In xin)
    {
        uint32_t x = uint32_t(xin);
#pragma unroll
        for (int k = 0; k < K; k++) {
            uint32_t y = 0;
#pragma unroll
            for (int i = 0; i < NMAX; i++) {
                const Adaptor &c = p.ad[k][i];
                const uint32_t x0 = x, x1 = uint32_t(z[k][i]);
                const uint32_t pr = uint32_t(mulq32(int32_t((x1 - x0) * uint32_t(c.s)), c.a));
                const uint32_t o0 = pr + x0 * uint32_t(c.k00) + x1 * uint32_t(c.k01);
                const uint32_t o1 = pr + x0 * uint32_t(c.k10) + x1 * uint32_t(c.k11);
                if (i == 0)
                    y = o0;
                else
                    z[k][i - 1] = int32_t(o0);
                x = o1;
            }
            z[k][NMAX - 1] = int32_t(x);
            x = y;
        }
        return int32_t(x);
    }
};

int wdf_cfg_check(const idsp_wdf *c, size_t n)
{
    for (size_t k = 0; k < n; k++)
        if (c[k].n < 1 || c[k].n > IDSP_WDF_MAX_ORDER) return fail(IDSP_EINVAL, "section %zu: Wdf order N = %d not in 1..%d", k, c[k].n, IDSP_WDF_MAX_ORDER);
    return IDSP_OK;
}

template <int K, int NMAX>
int wdf_launch_n(const idsp_wdf *c, uint32_t *st, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, hipStream_t s)
{
    WdfParamsT<NMAX> p{};
    for (int k = 0; k < K; k++) {
        p.n[k] = c[k].n;
        uint32_t m = c[k].m;
        for (int i = 0; i < NMAX; i++, m >>= 4) p.ad[k][i] = i < c[k].n ? adaptor_of(m & 0xf, c[k].a[i]) : adaptor_of(0, 0);
    }
    return launch_stream<WdfChain<K, NMAX>>(p, st, x, y, lanes, frames, layout, s);
}

template <int K, int NMAX>
WdfParamsT<NMAX> wdf_params(const idsp_wdf *c)
{
    WdfParamsT<NMAX> p{};
    for (int k = 0; k < K; k++) {
        p.n[k] = c[k].n;
        uint32_t m = c[k].m;
        for (int i = 0; i < NMAX; i++, m >>= 4) p.ad[k][i] = i < c[k].n ? adaptor_of(m & 0xf, c[k].a[i]) : adaptor_of(0, 0);
    }
    return p;
}

// K sections over two waves per 64 lanes (lane_stream.h, stream_frame_major_duo): the first KA on wave 0, the rest on wave 1,
// whose state planes start behind the first KA sections' words.  ~80 VALU instructions per sample for the 7th-order chain of the
// reference's embedded bench: 0.81 -> 0.70 ms at 65536 lanes x 4096 frames (profiles/r03_wdf_duo.log).
template <int KA, int KB, int NMAX>
int wdf_launch_duo_n(const idsp_wdf *c, uint32_t *st, const int32_t *x, int32_t *y, size_t lanes, size_t frames, hipStream_t s)
{
    size_t wa = 0;
    for (int k = 0; k < KA; k++) wa += size_t(c[k].n);
    return launch_duo<WdfChain<KA, NMAX>, WdfChain<KB, NMAX>>(wdf_params<KA, NMAX>(c), wdf_params<KB, NMAX>(c + KA), st, st + wa * lanes, x, y, lanes, frames, s,
                                                               Pitch{});
}
template <int KA, int KB>
int wdf_launch_duo(const idsp_wdf *c, uint32_t *st, const int32_t *x, int32_t *y, size_t lanes, size_t frames, hipStream_t s)
{
    int nmax = 0;
    for (int k = 0; k < KA + KB; k++) nmax = c[k].n > nmax ? c[k].n : nmax;
    if (nmax <= 2) return wdf_launch_duo_n<KA, KB, 2>(c, st, x, y, lanes, frames, s);
    if (nmax <= 4) return wdf_launch_duo_n<KA, KB, 4>(c, st, x, y, lanes, frames, s);
    return 1;  // sections of order 5 .. 8: the two-wave form would index its state through scratch; the caller keeps one wave
}

template <int K>
int wdf_launch(const idsp_wdf *c, uint32_t *st, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, hipStream_t s)
{
    if constexpr (K >= 2) {
        if (duo_wanted(5, lanes, layout)) {
            const int rc = wdf_launch_duo<(K + 1) / 2, K / 2>(c, st, x, y, lanes, frames, s);
            if (rc <= 0) return rc;
        }
    }
    int nmax = 0;
    for (int k = 0; k < K; k++) nmax = c[k].n > nmax ? c[k].n : nmax;
    if (nmax <= 2) return wdf_launch_n<K, 2>(c, st, x, y, lanes, frames, layout, s);
    if (nmax <= 4) return wdf_launch_n<K, 4>(c, st, x, y, lanes, frames, layout, s);
    return wdf_launch_n<K, IDSP_WDF_MAX_ORDER>(c, st, x, y, lanes, frames, layout, s);
}

}  // namespace
}  // namespace idsp

using namespace idsp;
using namespace idsp::bq;

extern "C" {

int idsp_wdf_quantize(int n, uint32_t m, const double *g, idsp_wdf *out)
{
    if (!g || !out) return fail(IDSP_EINVAL, "g or out is NULL");
    if (n < 1 || n > IDSP_WDF_MAX_ORDER) return fail(IDSP_EINVAL, "Wdf order N = %d not in 1..%d", n, IDSP_WDF_MAX_ORDER);
    out->n = n, out->m = m;
    for (int i = 0; i < IDSP_WDF_MAX_ORDER; i++) out->a[i] = 0;
    uint32_t mm = m;
    for (int i = 0; i < n; i++, mm >>= 4) {
        // `Tpa::quantize` (src/iir/wdf.rs:50-62): a in [-0.5, 0], then Q32::<32>::from_f64
        double a;
        switch (mm & 0xf) {
            case 0xA: a = g[i] - 1.0; break;
            case 0xB: case 0xE: a = -g[i]; break;
            case 0xC: case 0xF: a = g[i]; break;
            case 0xD: a = -1.0 - g[i]; break;
            default: a = 0.0; break;  // Z, X
        }
        if (!(a >= -0.5 && a <= 0.0)) return fail(IDSP_EOUTOFRANGE, "parameter `g[%d]` is out of range", i);
        const double r = std::round(a * 4294967296.0);
        out->a[i] = r <= -2147483648.0 ? INT32_MIN : int32_t(r);
    }
    return IDSP_OK;
}

size_t idsp_wdf_state_words(const idsp_wdf *cfg, size_t n_sections)
{
    if (!cfg || wdf_cfg_check(cfg, n_sections)) return 0;
    size_t w = 0;
    for (size_t k = 0; k < n_sections; k++) w += size_t(cfg[k].n);
    return w;
}

int idsp_wdf_i32(const idsp_wdf *cfg, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                 int layout, void *stream)
{
    int rc = check_stream_args(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if ((rc = wdf_cfg_check(cfg, n))) return rc;
    if (lanes == 0 || frames == 0) return IDSP_OK;
    hipStream_t s = as_stream(stream);
    if (n == 0) {  // empty chain: identity (compose.rs:63-65)
        if (x != y) IDSP_HIP_TRY(hipMemcpyAsync(y, x, lanes * frames * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        return IDSP_OK;
    }
    uint32_t *st = static_cast<uint32_t *>(state);
    const int32_t *src = x;
    size_t done = 0;
    while (done < n) {  // stage-major passes like run_chain
        const size_t k = n - done < size_t(kWdfMaxSections) ? n - done : size_t(kWdfMaxSections);
        switch (k) {
            case 1: rc = wdf_launch<1>(cfg + done, st, src, y, lanes, frames, layout, s); break;
            case 2: rc = wdf_launch<2>(cfg + done, st, src, y, lanes, frames, layout, s); break;
            case 3: rc = wdf_launch<3>(cfg + done, st, src, y, lanes, frames, layout, s); break;
            default: rc = wdf_launch<4>(cfg + done, st, src, y, lanes, frames, layout, s); break;
        }
        if (rc) return rc;
        for (size_t j = 0; j < k; j++) st += size_t(cfg[done + j].n) * lanes;
        done += k;
        src = y;
    }
    return IDSP_OK;
}

}  // extern "C"
