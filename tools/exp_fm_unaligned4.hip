// exp_fm_unaligned4.hip — round-3 experiment: FrameMajor rows that are only 4-byte aligned (dense tensors whose lane count is not a
// multiple of four: row f starts 4 (lanes % 4) f bytes off the 16-byte grid).  Does the LDS-DMA kernel's `global_load_lds_dwordx4`
// / 16-byte store pair work on such rows at all, and how fast, against the register-window kernel (4-byte accesses) that
// launch_stream takes for them today?  Compares the outputs word for word before timing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fwrapv -ffp-contract=off -fno-slp-vectorize -w -Iinclude -Iidsp_amd/csrc tools/exp_fm_unaligned4.hip -o build/exp_fm_un4
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "biquad_sections.h"

namespace idsp {
char *last_error_buf() { static thread_local char b[512]; return b; }
int fail(int code, const char *, ...) { return code; }
void note_kernel(const char *, const char *) {}
}  // namespace idsp
using namespace idsp;
using P = bq::Chain<bq::Df1I32<false>, 1>;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class F>
float timeit(F &&launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int i = 0; i < 30; i++) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 10) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main()
{
    const size_t frames = 4096, cap = 66000;
    P::Params prm{};
    prm.sec[0] = {{1 << 26, 1 << 27, 1 << 26, 1 << 29, -(1 << 28)}, 30, 0, 0, 0};
    int32_t *x, *y, *yr;
    uint32_t *st, *st2;
    CK(hipMalloc(&x, cap * frames * 4));
    CK(hipMalloc(&y, cap * frames * 4));
    CK(hipMalloc(&yr, cap * frames * 4));
    CK(hipMalloc(&st, cap * 16));
    CK(hipMalloc(&st2, cap * 16));
    std::vector<int32_t> hx(cap * frames);
    uint32_t s = 12345;
    for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = int32_t(s) >> 4; }
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    constexpr size_t bytes = (size_t(7) * kLdsT * kFmBlock + 2 * kLdsT * kFmBlock) * 4;
    auto k0 = stream_frame_major_lds<P, 7, 1, false, false>;
    auto k1 = stream_frame_major_lds<P, 7, 1, false, true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    struct Sh { size_t lanes, pitch, off; };
    for (auto sh : std::vector<Sh>{{65536, 65536, 0}, {65536, 65536, 1}, {65536, 65537, 0}, {65536, 65538, 0}, {65536, 65539, 3}, {65280, 65281, 0}}) {
        const unsigned grid = unsigned(sh.lanes / kFmBlock);
        const int32_t *xs = x + sh.off;
        int32_t *ys = y + sh.off, *yrs = yr + sh.off;
        // reference: register-window kernel
        CK(hipMemset(st, 0, cap * 16));
        CK(hipMemset(st2, 0, cap * 16));
        CK(hipMemset(y, 0x55, cap * frames * 4));
        CK(hipMemset(yr, 0x55, cap * frames * 4));
        hipLaunchKernelGGL((stream_frame_major<P, 24>), dim3(grid), dim3(kFmBlock), 0, 0, prm, st, xs, yrs, sh.lanes, frames, sh.pitch, sh.pitch, 0);
        hipLaunchKernelGGL(k1, dim3(grid), dim3(kFmBlock), bytes, 0, prm, st2, xs, ys, sh.lanes, frames, sh.pitch, sh.pitch, sh.lanes);
        CK(hipDeviceSynchronize());
        std::vector<int32_t> a(cap * frames), b(cap * frames);
        CK(hipMemcpy(a.data(), y, a.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), yr, b.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < a.size(); i++) bad += a[i] != b[i];
        const float tr = timeit([&] { hipLaunchKernelGGL((stream_frame_major<P, 24>), dim3(grid), dim3(kFmBlock), 0, 0, prm, st, xs, yrs, sh.lanes, frames, sh.pitch, sh.pitch, 0); });
        const float trx = timeit([&] { hipLaunchKernelGGL((stream_frame_major<P, 24>), dim3(grid), dim3(kFmBlock), 0, 0, prm, st, xs, yrs, sh.lanes, frames, sh.pitch, sh.pitch, 1); });
        const float t0 = timeit([&] { hipLaunchKernelGGL(k0, dim3(grid), dim3(kFmBlock), bytes, 0, prm, st2, xs, ys, sh.lanes, frames, sh.pitch, sh.pitch, sh.lanes); });
        const float t1 = timeit([&] { hipLaunchKernelGGL(k1, dim3(grid), dim3(kFmBlock), bytes, 0, prm, st2, xs, ys, sh.lanes, frames, sh.pitch, sh.pitch, sh.lanes); });
        const double gb = double(sh.lanes) * frames * 8 / 1e9;
        printf("{\"lanes\": %zu, \"pitch\": %zu, \"offset_words\": %zu, \"mismatching_words\": %zu, \"window_frac\": %.3f, \"window_xcdc_frac\": %.3f, \"lds_frac\": %.3f, \"lds_xcdc_frac\": %.3f}\n",
               sh.lanes, sh.pitch, sh.off, bad, gb / (tr * 1e-3) / 8000, gb / (trx * 1e-3) / 8000, gb / (t0 * 1e-3) / 8000, gb / (t1 * 1e-3) / 8000);
        fflush(stdout);
    }
    return 0;
}
