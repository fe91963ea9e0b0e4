// api_util.hip — library/device utilities, coefficient ingestion and the
// built-in half-band tap sets of the C ABI (include/idsp_hip.h).  Host code
// only; nothing here is on the per-sample path.
#include <cxxabi.h>

#include <cmath>
#include <cstring>

#include "common.h"
#include "hbf_taps.h"

namespace idsp {

char *last_error_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

namespace {
struct LastKernel {
    const char *kernel = nullptr, *detail = nullptr, *also = nullptr;
    char text[768] = {0};
};
LastKernel &last_kernel()
{
    static thread_local LastKernel k;
    return k;
}
}  // namespace

// two pointer stores per launch; the text is built only when idsp_last_kernel() is called
void note_kernel(const char *kernel, const char *detail)
{
    LastKernel &k = last_kernel();
    k.kernel = kernel;
    k.detail = detail;
    k.also = nullptr;
}
void note_kernel_also(const char *also) { last_kernel().also = also; }
const char *noted_kernel() { return last_kernel().kernel; }

namespace {

// float -> Q: `(v * 2^F).round()` half away from zero, then a saturating
// `as i32` with NaN -> 0 (dsp-fixedpoint/src/num_traits_impl.rs:32-46; the
// scale 1/DELTA is an exact power of two, lib.rs:220-224).
int32_t to_q32(double v, int frac)
{
    const double s = std::round(std::ldexp(v, frac));
    if (std::isnan(s)) return 0;
    if (s >= 2147483647.0) return INT32_MAX;
    if (s <= -2147483648.0) return INT32_MIN;
    return int32_t(s);
}

int hbf_fill(int tap_set, int stages, bool dec, idsp_hbf_cascade_f32 *out)
{
    if (!out) return fail(IDSP_EINVAL, "out is NULL");
    if (tap_set < 0 || tap_set > 1) return fail(IDSP_EINVAL, "tap_set %d not in {0,1}", tap_set);
    if (stages < 1 || stages > IDSP_HBF_MAX_STAGES) return fail(IDSP_EINVAL, "stages %d not in 1..5", stages);
    std::memset(out, 0, sizeof(*out));
    out->stages = stages;
    for (int s = 0; s < stages; s++) {
        // decimator: highest-rate stage (tuple index stages-1) first, src/hbf.rs:412-421;
        // interpolator: tuple index 0 first, src/hbf.rs:503-512
        const int t = hbf_tuple_index(dec, stages, s);
        out->m[s] = kHbfM[tap_set][t];
        for (int k = 0; k < out->m[s]; k++) out->taps[s][k] = kHbfTaps[tap_set][t][k];
    }
    return IDSP_OK;
}

}  // namespace

bool hbf_cfg_ok(const idsp_hbf_cascade_f32 *c)
{
    if (!c || c->stages < 1 || c->stages > IDSP_HBF_MAX_STAGES) return false;
    for (int s = 0; s < c->stages; s++)
        if (c->m[s] < 1 || c->m[s] > IDSP_HBF_MAX_TAPS) return false;
    return true;
}

// idsp_device_copy: workgroup w moves the pieces [w per, (w + 1) per) front to back, 8 x 256 of them per step, nontemporal
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_chunk_kernel(u32x4_t *dst, const u32x4_t *src, size_t n)
{
    constexpr int U = 8;
    const size_t per = (n + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n ? lo + per : n;
    size_t i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; u++) __builtin_nontemporal_store(v[u], dst + i + u * 256);
    }
    for (; i < hi; i += 256) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
__global__ __launch_bounds__(256) void copy_bytes_kernel(unsigned char *dst, const unsigned char *src, size_t n)
{
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) dst[i] = src[i];
}

}  // namespace idsp

using namespace idsp;

extern "C" {

int idsp_version(void) { return IDSP_ABI_VERSION; }

const char *idsp_last_error(void) { return last_error_buf(); }

const char *idsp_last_kernel(void)
{
    auto &k = last_kernel();
    if (!k.kernel) return "";
    if (k.detail) {
        // `detail` is typeid(Processor).name(): demangle it for the reader
        int status = 0;
        char *dm = abi::__cxa_demangle(k.detail, nullptr, nullptr, &status);
        snprintf(k.text, sizeof(k.text), "%s<%s>%s", k.kernel, status == 0 && dm ? dm : k.detail, k.also ? k.also : "");
        free(dm);
    } else {
        snprintf(k.text, sizeof(k.text), "%s%s", k.kernel, k.also ? k.also : "");
    }
    return k.text;
}

int idsp_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(IDSP_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int idsp_device_set(int device)
{
    IDSP_HIP_TRY(hipSetDevice(device));
    return IDSP_OK;
}

int idsp_device_alloc(void **ptr, size_t bytes)
{
    if (!ptr) return fail(IDSP_EINVAL, "ptr is NULL");
    IDSP_HIP_TRY(hipMalloc(ptr, bytes));
    return IDSP_OK;
}

int idsp_device_free(void *ptr)
{
    IDSP_HIP_TRY(hipFree(ptr));
    return IDSP_OK;
}

int idsp_device_memset(void *ptr, int value, size_t bytes, void *stream)
{
    IDSP_HIP_TRY(hipMemsetAsync(ptr, value, bytes, as_stream(stream)));
    return IDSP_OK;
}

int idsp_device_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream)
{
    IDSP_HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return IDSP_OK;
}

int idsp_device_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream)
{
    IDSP_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return IDSP_OK;
}

int idsp_device_copy(void *dst_dev, const void *src_dev, size_t bytes, void *stream)
{
    if (bytes && (!dst_dev || !src_dev)) return fail(IDSP_EINVAL, "dst or src is NULL");
    if (bytes == 0 || dst_dev == src_dev) return IDSP_OK;
    auto *d = static_cast<unsigned char *>(dst_dev);
    auto *s = static_cast<const unsigned char *>(src_dev);
    if ((s < d + bytes) && (d < s + bytes)) return fail(IDSP_EINVAL, "overlapping buffers");
    hipStream_t q = as_stream(stream);
    // head up to dst's 16-byte grid and everything when the two are not congruent mod 16: byte kernel
    const uintptr_t da = reinterpret_cast<uintptr_t>(d), sa = reinterpret_cast<uintptr_t>(s);
    if ((da ^ sa) & 15) {
        hipLaunchKernelGGL(copy_bytes_kernel, dim3(unsigned((bytes + 255) / 256 > 65535 * 16 ? 65535 * 16 : (bytes + 255) / 256)), dim3(256), 0, q, d, s, bytes);
        return launch_status();
    }
    const size_t head = (16 - (da & 15)) & 15, h = head < bytes ? head : bytes;
    if (h) hipLaunchKernelGGL(copy_bytes_kernel, dim3(1), dim3(256), 0, q, d, s, h);
    const size_t n16 = (bytes - h) / 16, tail = bytes - h - n16 * 16;
    if (n16) {
        // chunk per workgroup: 2048 workgroups of 256 threads, 8 pieces per thread and step (tools/ubench_copy_big.hip: 5.1-5.6 TB/s
        // at 1-16 GiB against 4.2-5.1 for a grid-stride sweep)
        const unsigned grid = unsigned(n16 < 2048u * 2048u ? (n16 + 2047) / 2048 : 2048);
        hipLaunchKernelGGL(copy_chunk_kernel, dim3(grid), dim3(256), 0, q, reinterpret_cast<u32x4_t *>(d + h), reinterpret_cast<const u32x4_t *>(s + h), n16);
    }
    if (tail) hipLaunchKernelGGL(copy_bytes_kernel, dim3(1), dim3(256), 0, q, d + h + n16 * 16, s + h + n16 * 16, tail);
    return launch_status();
}

int idsp_stream_sync(void *stream)
{
    IDSP_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return IDSP_OK;
}

int idsp_device_sync(void)
{
    IDSP_HIP_TRY(hipDeviceSynchronize());
    return IDSP_OK;
}

// src/iir/biquad.rs:545-566 then :570-576
int idsp_biquad_i32_from_sos(const double sos[6], int frac, idsp_biquad_i32 *out)
{
    if (!sos || !out) return fail(IDSP_EINVAL, "sos or out is NULL");
    if (frac < 0 || frac > 31) return fail(IDSP_EINVAL, "frac = %d not in 0..31", frac);
    const double a0 = 1.0 / sos[3];
    const double ba[5] = {sos[0] * a0, sos[1] * a0, sos[2] * a0, -sos[4] * a0, -sos[5] * a0};
    for (int i = 0; i < 5; i++) out->ba[i] = to_q32(ba[i], frac);
    out->frac = frac;
    return IDSP_OK;
}

int idsp_biquad_f32_from_sos(const float sos[6], idsp_biquad_f32 *out)
{
    if (!sos || !out) return fail(IDSP_EINVAL, "sos or out is NULL");
    const float a0 = 1.0f / sos[3];
    out->ba[0] = sos[0] * a0;
    out->ba[1] = sos[1] * a0;
    out->ba[2] = sos[2] * a0;
    out->ba[3] = -sos[4] * a0;
    out->ba[4] = -sos[5] * a0;
    return IDSP_OK;
}

int idsp_biquad_f32_from_sos_f64(const double sos[6], idsp_biquad_f32 *out)
{
    if (!sos || !out) return fail(IDSP_EINVAL, "sos or out is NULL");
    const double a0 = 1.0 / sos[3];
    const double ba[5] = {sos[0] * a0, sos[1] * a0, sos[2] * a0, -sos[4] * a0, -sos[5] * a0};
    for (int i = 0; i < 5; i++) out->ba[i] = float(ba[i]);
    return IDSP_OK;
}

int idsp_biquad_f64_from_sos(const double sos[6], idsp_biquad_f64 *out)
{
    if (!sos || !out) return fail(IDSP_EINVAL, "sos or out is NULL");
    const double a0 = 1.0 / sos[3];
    out->ba[0] = sos[0] * a0;
    out->ba[1] = sos[1] * a0;
    out->ba[2] = sos[2] * a0;
    out->ba[3] = -sos[4] * a0;
    out->ba[4] = -sos[5] * a0;
    return IDSP_OK;
}

int idsp_hbf_dec_cascade(int tap_set, int stages, idsp_hbf_cascade_f32 *out) { return hbf_fill(tap_set, stages, true, out); }
int idsp_hbf_int_cascade(int tap_set, int stages, idsp_hbf_cascade_f32 *out) { return hbf_fill(tap_set, stages, false, out); }

// src/hbf.rs:424-448
int idsp_hbf_dec_response_length(const idsp_hbf_cascade_f32 *cfg)
{
    if (!hbf_cfg_ok(cfg)) return fail(IDSP_EINVAL, "invalid hbf cascade");
    int n = 0;
    for (int s = 0; s < cfg->stages; s++) n = n / 2 + 2 * cfg->m[s] - 1;
    return n;
}

// src/hbf.rs:515-539
int idsp_hbf_int_response_length(const idsp_hbf_cascade_f32 *cfg)
{
    if (!hbf_cfg_ok(cfg)) return fail(IDSP_EINVAL, "invalid hbf cascade");
    int n = 0;
    for (int s = 0; s < cfg->stages; s++) n = (n + 2 * cfg->m[s] - 1) * 2;
    return n;
}

size_t idsp_hbf_dec_state_words(const idsp_hbf_cascade_f32 *cfg)
{
    if (!hbf_cfg_ok(cfg)) return 0;
    size_t w = 0;
    for (int s = 0; s < cfg->stages; s++) w += size_t(3 * cfg->m[s] - 2);
    return w;
}

size_t idsp_hbf_int_state_words(const idsp_hbf_cascade_f32 *cfg)
{
    if (!hbf_cfg_ok(cfg)) return 0;
    size_t w = 0;
    for (int s = 0; s < cfg->stages; s++) w += size_t(2 * cfg->m[s] - 1);
    return w;
}

}  // extern "C"
