// hbf_f64.hip — the half-band decimator / interpolator cascades and the four `type_fir!` same-rate FIR types on f64 samples with
// f64 taps (`EvenSymmetric<[f64; M]>` etc. are generic in the sample type, src/hbf.rs:46-68,70-138,142-236).  One workgroup per
// lane walks the lane in chunks of 2048 high-rate samples; all stages of a cascade run inside the chunk with the inter-stage
// streams in LDS (the reference's `Major` scratch, dsp-process/src/compose.rs:581-593), each preceded by the history the
// reference keeps with `copy_within` (src/hbf.rs:103,182-183,224).  Tap counts and taps are runtime values (any cascade the
// C ABI can describe); arithmetic is `get()` verbatim: sum_k (new_k +/- old_k) * tap_k from -0.0, sequential, nothing fused
// (-ffp-contract=off).  f64 is a completeness path (no BASELINE configuration uses it): plain code, no layout-specific kernels.
#include <cstring>

#include "common.h"
#include "hbf_taps.h"

namespace idsp {
namespace {

constexpr int kT = 256;      // threads per workgroup
constexpr int kCh = 2048;    // high-rate samples per chunk and lane

struct H64Args {
    int32_t stages;
    int32_t m[IDSP_HBF_MAX_STAGES];
    int32_t buf_a[IDSP_HBF_MAX_STAGES];  // LDS offset (doubles): decimator even stream / interpolator x stream, history first
    int32_t buf_b[IDSP_HBF_MAX_STAGES];  // decimator odd stream
    int32_t st_off[IDSP_HBF_MAX_STAGES + 1];  // state VALUE offset of each stage (+ total)
    double taps[IDSP_HBF_MAX_STAGES][IDSP_HBF_MAX_TAPS];
};

__device__ __forceinline__ double ld_state(const uint32_t *st, size_t v, size_t lanes, size_t lane)
{
    const uint64_t u = (uint64_t(st[(2 * v + 1) * lanes + lane]) << 32) | st[(2 * v) * lanes + lane];
    return __longlong_as_double((long long)u);
}
__device__ __forceinline__ void st_state(uint32_t *st, size_t v, size_t lanes, size_t lane, double d)
{
    const uint64_t u = (uint64_t)__double_as_longlong(d);
    st[(2 * v) * lanes + lane] = uint32_t(u);
    st[(2 * v + 1) * lanes + lane] = uint32_t(u >> 32);
}

// history <-> state for a list of streams described by (offset, length) pairs per stage
template <bool DEC>
__device__ void hist_io(const H64Args &a, double *lds, uint32_t *st, size_t lanes, size_t lane, bool load)
{
    for (int w = threadIdx.x; w < a.st_off[a.stages]; w += kT) {
        int s = 0;
        while (w >= a.st_off[s + 1]) s++;
        const int M = a.m[s], j = w - a.st_off[s];
        double *p;
        if (DEC)
            p = j < M - 1 ? lds + a.buf_a[s] + j : lds + a.buf_b[s] + (j - (M - 1));
        else
            p = lds + a.buf_a[s] + j;
        if (load)
            *p = ld_state(st, size_t(w), lanes, lane);
        else
            st_state(st, size_t(w), lanes, lane, *p);
    }
}

// after a chunk with `nin` high-rate samples: every stream's last H samples become its history (src/hbf.rs:182-183,224)
template <bool DEC>
__device__ void roll(const H64Args &a, double *lds, int nin)
{
    double keep[2];  // at most 2 words per thread: <= 5 stages x (3 x 32 - 2) values / 256 threads
    int c = 0;
    for (int w = threadIdx.x; w < a.st_off[a.stages]; w += kT, c++) {
        int s = 0;
        while (w >= a.st_off[s + 1]) s++;
        const int M = a.m[s], j = w - a.st_off[s];
        const int ns = DEC ? nin >> (s + 1) : (nin >> a.stages) << s;  // samples this stage consumed
        if (DEC)
            keep[c] = j < M - 1 ? lds[a.buf_a[s] + ns + j] : lds[a.buf_b[s] + ns + (j - (M - 1))];
        else
            keep[c] = lds[a.buf_a[s] + ns + j];
    }
    __syncthreads();
    c = 0;
    for (int w = threadIdx.x; w < a.st_off[a.stages]; w += kT, c++) {
        int s = 0;
        while (w >= a.st_off[s + 1]) s++;
        const int M = a.m[s], j = w - a.st_off[s];
        if (DEC) {
            if (j < M - 1)
                lds[a.buf_a[s] + j] = keep[c];
            else
                lds[a.buf_b[s] + (j - (M - 1))] = keep[c];
        } else {
            lds[a.buf_a[s] + j] = keep[c];
        }
    }
    __syncthreads();
}

// `HbfDec` cascade (src/hbf.rs:156-192,385-421).  x[(idx(f,l))*R + k], y[idx(f,l)]
__global__ __launch_bounds__(kT) void hbf_dec_f64_kernel(const H64Args a, uint32_t *st, const double *x, double *y, const size_t lanes,
                                                         const size_t frames, const int lane_major)
{
    extern __shared__ __attribute__((aligned(16))) double lds64[];
    double *lds = lds64;
    const size_t lane = blockIdx.x;
    const int S = a.stages, R = 1 << S, tid = threadIdx.x;
    hist_io<true>(a, lds, st, lanes, lane, true);
    __syncthreads();
    const size_t ch = size_t(kCh >> S);  // output frames per chunk
    for (size_t f0 = 0; f0 < frames; f0 += ch) {
        const int nf = int(frames - f0 < ch ? frames - f0 : ch), nin = nf * R;
        {
            const int M0 = a.m[0], ppf = R / 2;
            double *E0 = lds + a.buf_a[0] + (M0 - 1), *O0 = lds + a.buf_b[0] + (2 * M0 - 1);
            for (int q = tid; q < nin / 2; q += kT) {
                const size_t f = f0 + size_t(q / ppf);
                const size_t base = (lane_major ? lane * frames + f : f * lanes + lane) * size_t(R) + size_t(q % ppf) * 2;
                E0[q] = x[base];
                O0[q] = x[base + 1];
            }
        }
        __syncthreads();
        int n = nin;
        for (int s = 0; s < S; s++) {
            n >>= 1;
            const int M = a.m[s];
            const double *E = lds + a.buf_a[s], *O = lds + a.buf_b[s];
            double *En = nullptr, *On = nullptr;
            if (s + 1 < S) {
                En = lds + a.buf_a[s + 1] + (a.m[s + 1] - 1);
                On = lds + a.buf_b[s + 1] + (2 * a.m[s + 1] - 1);
            }
            for (int i = tid; i < n; i += kT) {
                double acc = -0.0;
                for (int k = 0; k < M; k++) acc = acc + (O[i + 2 * M - 1 - k] + O[i + k]) * a.taps[s][k];
                const double out = acc + E[i];
                if (s + 1 == S)
                    y[lane_major ? lane * frames + f0 + size_t(i) : (f0 + size_t(i)) * lanes + lane] = out;
                else if (i & 1)
                    On[i >> 1] = out;  // `ChunkIn<_, 2>`: consecutive outputs pair up as the next [even, odd]
                else
                    En[i >> 1] = out;
            }
            __syncthreads();
        }
        roll<true>(a, lds, nin);
    }
    hist_io<true>(a, lds, st, lanes, lane, false);
}

// `HbfInt` cascade (src/hbf.rs:200-236,476-512).  x[idx(f,l)], y[idx(f,l)*R + k]
__global__ __launch_bounds__(kT) void hbf_int_f64_kernel(const H64Args a, uint32_t *st, const double *x, double *y, const size_t lanes,
                                                         const size_t frames, const int lane_major)
{
    extern __shared__ __attribute__((aligned(16))) double lds64[];
    double *lds = lds64;
    const size_t lane = blockIdx.x;
    const int S = a.stages, R = 1 << S, tid = threadIdx.x;
    hist_io<false>(a, lds, st, lanes, lane, true);
    __syncthreads();
    const size_t ch = size_t(kCh >> S);  // input frames per chunk
    for (size_t f0 = 0; f0 < frames; f0 += ch) {
        const int nf = int(frames - f0 < ch ? frames - f0 : ch);
        {
            double *X0 = lds + a.buf_a[0] + (2 * a.m[0] - 1);
            for (int i = tid; i < nf; i += kT) X0[i] = x[lane_major ? lane * frames + f0 + size_t(i) : (f0 + size_t(i)) * lanes + lane];
        }
        __syncthreads();
        int n = nf;
        for (int s = 0; s < S; s++) {
            const int M = a.m[s];
            const double *X = lds + a.buf_a[s];
            double *Xn = s + 1 < S ? lds + a.buf_a[s + 1] + (2 * a.m[s + 1] - 1) : nullptr;
            for (int i = tid; i < n; i += kT) {
                double acc = -0.0;
                for (int k = 0; k < M; k++) acc = acc + (X[i + 2 * M - 1 - k] + X[i + k]) * a.taps[s][k];
                const double ctr = X[i + M];  // centre tap: identity
                if (s + 1 == S) {
                    const int j = 2 * i;  // chunk-local output sample; j and j + 1 lie in one frame (R >= 2)
                    const size_t f = f0 + size_t(j / R);
                    double *dst = y + (lane_major ? lane * frames + f : f * lanes + lane) * size_t(R) + size_t(j % R);
                    dst[0] = acc;
                    dst[1] = ctr;
                } else {
                    Xn[2 * i] = acc;  // `ChunkOut<_, 2>`: the pairs flatten into the next stage's input
                    Xn[2 * i + 1] = ctr;
                }
            }
            n *= 2;
            __syncthreads();
        }
        roll<false>(a, lds, nf * R);
    }
    hist_io<false>(a, lds, st, lanes, lane, false);
}

struct F64Args {
    int32_t m, odd, sym;
    double taps[IDSP_HBF_MAX_TAPS];
};
// `type_fir!` (src/hbf.rs:70-138): window of 2M + odd samples ending at the current input
__global__ __launch_bounds__(kT) void fir_sym_f64_kernel(const F64Args a, uint32_t *st, const double *x, double *y, const size_t lanes,
                                                         const size_t frames, const int lane_major)
{
    __shared__ double buf[2 * IDSP_HBF_MAX_TAPS + kCh];
    const size_t lane = blockIdx.x;
    const int M = a.m, len = 2 * M - 1 + a.odd, tid = threadIdx.x;
    for (int w = tid; w < len; w += kT) buf[w] = ld_state(st, size_t(w), lanes, lane);
    __syncthreads();
    for (size_t f0 = 0; f0 < frames; f0 += kCh) {
        const int n = int(frames - f0 < size_t(kCh) ? frames - f0 : size_t(kCh));
        for (int i = tid; i < n; i += kT) buf[len + i] = x[lane_major ? lane * frames + f0 + size_t(i) : (f0 + size_t(i)) * lanes + lane];
        __syncthreads();
        for (int i = tid; i < n; i += kT) {
            const double *w = buf + i;
            double acc = -0.0;
            for (int k = 0; k < M; k++) {
                const double nw = w[2 * M - 1 + a.odd - k], od = w[k];
                acc = acc + (a.sym ? nw + od : nw - od) * a.taps[k];
            }
            y[lane_major ? lane * frames + f0 + size_t(i) : (f0 + size_t(i)) * lanes + lane] = (a.odd && a.sym) ? acc + w[M] : acc;
        }
        __syncthreads();
        double keep = 0.0;
        if (tid < len) keep = buf[n + tid];  // len <= 64 < kT
        __syncthreads();
        if (tid < len) buf[tid] = keep;
        __syncthreads();
    }
    for (int w = tid; w < len; w += kT) st_state(st, size_t(w), lanes, lane, buf[w]);
}

bool cfg_ok(const idsp_hbf_cascade_f64 *c)
{
    if (!c || c->stages < 1 || c->stages > IDSP_HBF_MAX_STAGES) return false;
    for (int s = 0; s < c->stages; s++)
        if (c->m[s] < 1 || c->m[s] > IDSP_HBF_MAX_TAPS) return false;
    return true;
}

int fill(int tap_set, int stages, bool dec, idsp_hbf_cascade_f64 *out)
{
    if (!out || tap_set < 0 || tap_set > 1 || stages < 1 || stages > 5) return fail(IDSP_EINVAL, "tap_set 0..1, stages 1..5, out != NULL");
    std::memset(out, 0, sizeof(*out));
    out->stages = stages;
    for (int s = 0; s < stages; s++) {
        const int t = hbf_tuple_index(dec, stages, s);
        out->m[s] = kHbfM[tap_set][t];
        for (int k = 0; k < out->m[s]; k++) out->taps[s][k] = double(kHbfTaps[tap_set][t][k]);
    }
    return IDSP_OK;
}

template <class K>
int launch(K kernel, const idsp_hbf_cascade_f64 *cfg, bool dec, void *state, const double *x, double *y, size_t lanes, size_t frames, int layout,
           void *stream)
{
    if (!cfg_ok(cfg)) return fail(IDSP_EINVAL, "invalid hbf cascade (stages 1..5, taps 1..32 per stage)");
    if (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR) return fail(IDSP_EINVAL, "bad layout %d", layout);
    if (lanes && (!state || (frames && (!x || !y)))) return fail(IDSP_EINVAL, "state, x or y is NULL");
    if (lanes > (size_t(1) << 31) - 1 || frames > (size_t(1) << 40)) return fail(IDSP_EINVAL, "lanes/frames out of range");
    if (lanes == 0 || frames == 0) return IDSP_OK;
    H64Args a;
    std::memset(&a, 0, sizeof(a));
    a.stages = cfg->stages;
    int off = 0, sv = 0;
    for (int s = 0; s < cfg->stages; s++) {
        const int M = cfg->m[s];
        a.m[s] = M;
        a.st_off[s] = sv;
        for (int k = 0; k < M; k++) a.taps[s][k] = cfg->taps[s][k];
        if (dec) {
            const int n = kCh >> (s + 1);
            a.buf_a[s] = off, off += (M - 1) + n + 2;
            a.buf_b[s] = off, off += (2 * M - 1) + n + 2;
            sv += 3 * M - 2;
        } else {
            const int n = (kCh >> cfg->stages) << s;
            a.buf_a[s] = off, off += (2 * M - 1) + n + 2;
            sv += 2 * M - 1;
        }
    }
    a.st_off[cfg->stages] = sv;
    const size_t bytes = size_t(off) * sizeof(double);
    if (bytes > 64 * 1024)
        IDSP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    note_kernel(dec ? "hbf_dec_f64_kernel" : "hbf_int_f64_kernel");
    hipLaunchKernelGGL(kernel, dim3(unsigned(lanes)), dim3(kT), bytes, as_stream(stream), a, static_cast<uint32_t *>(state), x, y, lanes, frames,
                       layout == IDSP_LANE_MAJOR ? 1 : 0);
    return launch_status();
}

}  // namespace
}  // namespace idsp

using namespace idsp;

extern "C" {

int idsp_hbf_dec_cascade_f64(int tap_set, int stages, idsp_hbf_cascade_f64 *out) { return fill(tap_set, stages, true, out); }
int idsp_hbf_int_cascade_f64(int tap_set, int stages, idsp_hbf_cascade_f64 *out) { return fill(tap_set, stages, false, out); }

size_t idsp_hbf_dec_state_words_f64(const idsp_hbf_cascade_f64 *cfg)
{
    if (!cfg_ok(cfg)) return 0;
    size_t w = 0;
    for (int s = 0; s < cfg->stages; s++) w += size_t(3 * cfg->m[s] - 2);
    return 2 * w;
}
size_t idsp_hbf_int_state_words_f64(const idsp_hbf_cascade_f64 *cfg)
{
    if (!cfg_ok(cfg)) return 0;
    size_t w = 0;
    for (int s = 0; s < cfg->stages; s++) w += size_t(2 * cfg->m[s] - 1);
    return 2 * w;
}
size_t idsp_fir_sym_state_words_f64(const idsp_fir_sym_f64 *cfg)
{
    if (!cfg || cfg->kind < 0 || cfg->kind > 3 || cfg->m < 1 || cfg->m > IDSP_HBF_MAX_TAPS) return 0;
    return 2 * size_t(2 * cfg->m - 1 + (cfg->kind == IDSP_FIR_ODD_SYMMETRIC || cfg->kind == IDSP_FIR_ODD_ANTISYMMETRIC ? 1 : 0));
}

int idsp_hbf_dec_f64(const idsp_hbf_cascade_f64 *cfg, void *state, const double *x, double *y, size_t lanes, size_t frames, int layout, void *stream)
{
    return launch(hbf_dec_f64_kernel, cfg, true, state, x, y, lanes, frames, layout, stream);
}
int idsp_hbf_int_f64(const idsp_hbf_cascade_f64 *cfg, void *state, const double *x, double *y, size_t lanes, size_t frames, int layout, void *stream)
{
    return launch(hbf_int_f64_kernel, cfg, false, state, x, y, lanes, frames, layout, stream);
}

int idsp_fir_sym_f64_process(const idsp_fir_sym_f64 *cfg, void *state, const double *x, double *y, size_t lanes, size_t frames, int layout,
                             void *stream)
{
    if (!idsp_fir_sym_state_words_f64(cfg)) return fail(IDSP_EINVAL, "invalid FIR configuration (kind 0..3, m 1..32)");
    if (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR) return fail(IDSP_EINVAL, "bad layout %d", layout);
    if (lanes && (!state || (frames && (!x || !y)))) return fail(IDSP_EINVAL, "state, x or y is NULL");
    if (lanes > (size_t(1) << 31) - 1) return fail(IDSP_EINVAL, "lanes out of range");
    if (lanes == 0 || frames == 0) return IDSP_OK;
    F64Args a;
    a.m = cfg->m;
    a.odd = (cfg->kind == IDSP_FIR_ODD_SYMMETRIC || cfg->kind == IDSP_FIR_ODD_ANTISYMMETRIC) ? 1 : 0;
    a.sym = (cfg->kind == IDSP_FIR_ODD_SYMMETRIC || cfg->kind == IDSP_FIR_EVEN_SYMMETRIC) ? 1 : 0;
    for (int k = 0; k < IDSP_HBF_MAX_TAPS; k++) a.taps[k] = k < cfg->m ? cfg->taps[k] : 0.0;
    note_kernel("fir_sym_f64_kernel");
    hipLaunchKernelGGL(fir_sym_f64_kernel, dim3(unsigned(lanes)), dim3(kT), 0, as_stream(stream), a, static_cast<uint32_t *>(state), x, y, lanes,
                       frames, layout == IDSP_LANE_MAJOR ? 1 : 0);
    return launch_status();
}

}  // extern "C"
