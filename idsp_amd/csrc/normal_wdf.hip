// normal_wdf.hip — C-ABI entry points (include/idsp_hip.h) of `iir::normal::Normal` (device code in
// biquad_sections.h) and the `iir::wdf::Wdf` wave-digital allpass sections (src/iir/wdf.rs).
#include <cmath>

#include "biquad_sections.h"

namespace idsp {
namespace {

// ------------------------------------------------------------------- Wdf
constexpr int kWdfMaxSections = 4;  // sections fused per launch; longer chains run in passes

struct WdfParams {
    idsp_wdf sec[kWdfMaxSections];
};

// `i32 * Q32<32>` (dsp-fixedpoint/src/lib.rs:449-456)
__device__ __forceinline__ int32_t mulq32(int32_t c, int32_t a) { return int32_t((int64_t(c) * int64_t(a)) >> 32); }
__device__ __forceinline__ int32_t wadd32(int32_t a, int32_t b) { return int32_t(uint32_t(a) + uint32_t(b)); }
__device__ __forceinline__ int32_t wsub32(int32_t a, int32_t b) { return int32_t(uint32_t(a) - uint32_t(b)); }

// `Tpa::adapt` (src/iir/wdf.rs:65-100); x = [x0, x1] -> [o0, o1].  The nibble is wave-uniform.
__device__ __forceinline__ void tpa_adapt(uint32_t nib, int32_t a, int32_t x0, int32_t x1, int32_t &o0, int32_t &o1)
{
    switch (nib) {
        case 0xA: {
            const int32_t c = wsub32(x1, x0), y = wadd32(mulq32(c, a), x1);
            o0 = wadd32(y, c), o1 = y;
            break;
        }
        case 0xB: {
            const int32_t c = wsub32(x0, x1), y = wadd32(mulq32(c, a), x1);
            o0 = y, o1 = wadd32(y, c);
            break;
        }
        case 0xE: {
            const int32_t c = wsub32(x0, x1), y = mulq32(c, a);
            o0 = wadd32(y, x1), o1 = wadd32(y, x0);
            break;
        }
        case 0x1: o0 = x1, o1 = x0; break;
        case 0xC: {
            const int32_t c = wsub32(x1, x0), y = wsub32(mulq32(c, a), x1);
            o0 = y, o1 = wadd32(y, c);
            break;
        }
        case 0xF: {
            const int32_t c = wsub32(x1, x0), y = mulq32(c, a);
            o0 = wsub32(y, x1), o1 = wsub32(y, x0);
            break;
        }
        case 0xD: {
            const int32_t c = wsub32(x0, x1), y = wsub32(mulq32(c, a), x1);
            o0 = wadd32(y, c), o1 = y;
            break;
        }
        default: o0 = x0, o1 = x1; break;  // Tpa::Z
    }
}

template <int K>
struct WdfChain {
    using In = int32_t;
    using Out = int32_t;
    static constexpr bool HAS_IN = true;
    static constexpr int LDS_WORDS = 0;
    static constexpr int IN_DIV = 1;
    static constexpr int COST = 40 * K;
    using Params = WdfParams;
    int32_t z[K][IDSP_WDF_MAX_ORDER];

    // state words: sections in order, N words each
    __device__ __forceinline__ void load(const Params &p, const uint32_t *st, size_t lanes, size_t lane)
    {
        int w = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
#pragma unroll
            for (int i = 0; i < IDSP_WDF_MAX_ORDER; i++) z[k][i] = i < p.sec[k].n ? int32_t(st[size_t(w + i) * lanes + lane]) : 0;
            w += p.sec[k].n;
        }
    }
    __device__ __forceinline__ void store(const Params &p, uint32_t *st, size_t lanes, size_t lane)
    {
        int w = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
#pragma unroll
            for (int i = 0; i < IDSP_WDF_MAX_ORDER; i++)
                if (i < p.sec[k].n) st[size_t(w + i) * lanes + lane] = uint32_t(z[k][i]);
            w += p.sec[k].n;
        }
    }
    // src/iir/wdf.rs:153-169: adaptor i maps [x, z_i] -> [o0, o1]; o0 is the section output for i = 0 and
    // the new z_{i-1} otherwise, o1 travels on as x; the last x becomes z_{N-1}
    __device__ __forceinline__ Out step(const Params &p, In x)
    {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const idsp_wdf &c = p.sec[k];
            int32_t y = 0;
            uint32_t m = c.m;
#pragma unroll
            for (int i = 0; i < IDSP_WDF_MAX_ORDER; i++) {
                if (i < c.n) {
                    int32_t o0, o1;
                    tpa_adapt(m & 0xf, c.a[i], x, z[k][i], o0, o1);
                    if (i == 0)
                        y = o0;
                    else
                        z[k][i - 1] = o0;
                    x = o1;
                    m >>= 4;
                }
            }
#pragma unroll
            for (int i = 0; i < IDSP_WDF_MAX_ORDER; i++)
                if (i == c.n - 1) z[k][i] = x;
            x = y;
        }
        return x;
    }
};

int wdf_cfg_check(const idsp_wdf *c, size_t n)
{
    for (size_t k = 0; k < n; k++)
        if (c[k].n < 1 || c[k].n > IDSP_WDF_MAX_ORDER) return fail(IDSP_EINVAL, "section %zu: Wdf order N = %d not in 1..%d", k, c[k].n, IDSP_WDF_MAX_ORDER);
    return IDSP_OK;
}

template <int K>
int wdf_launch(const idsp_wdf *c, uint32_t *st, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, hipStream_t s)
{
    WdfParams p{};
    for (int k = 0; k < K; k++) p.sec[k] = c[k];
    return launch_stream<WdfChain<K>>(p, st, x, y, lanes, frames, layout, s);
}

}  // namespace
}  // namespace idsp

using namespace idsp;
using namespace idsp::bq;

extern "C" {

int idsp_normal_i32_df1(const idsp_biquad_i32 *cfg, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes,
                        size_t frames, int layout, void *stream)
{
    return entry_i32<NormalI32, idsp_biquad_i32, FillI32>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

int idsp_normal_f32_df1(const idsp_biquad_f32 *cfg, size_t n, void *state, const float *x, float *y, size_t lanes,
                        size_t frames, int layout, void *stream)
{
    return entry_f32<NormalF32, idsp_biquad_f32, FillF32>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

int idsp_normal_f64_df1(const idsp_biquad_f64 *cfg, size_t n, void *state, const double *x, double *y, size_t lanes,
                        size_t frames, int layout, void *stream)
{
    return entry_f64<NormalF64, idsp_biquad_f64, FillF64>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

int idsp_normal_from_sos(const double sos[6], double out[5])
{
    if (!sos || !out) return fail(IDSP_EINVAL, "sos or out is NULL");
    // src/iir/normal.rs:62-76
    const double a0 = 1.0 / sos[3];
    const double p2 = -0.5 * sos[4];
    const double pq = sos[3] * sos[5] - p2 * p2;
    if (!(pq >= 0.0)) return fail(IDSP_EINVAL, "Normal::from: real poles (assert!(pq >= 0.0), src/iir/normal.rs:69)");
    out[0] = sos[0] * a0, out[1] = sos[1] * a0, out[2] = sos[2] * a0;
    out[3] = p2 * a0, out[4] = std::sqrt(pq) * a0;
    return IDSP_OK;
}

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
