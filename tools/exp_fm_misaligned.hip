// exp_fm_misaligned.hip — round-3 experiment: FrameMajor rows that do not start on 64-byte boundaries (dense 65000-lane
// tensors).  Runs the library's LDS-DMA kernel standalone with its plain block order and with XCD-contiguous lane blocks
// (template parameter XCDC).  The log of the round (profiles/r03_exp_fm_misaligned.jsonl) also holds the two variants that
// were built and removed: per-row rotation of the piece -> thread assignment, with and without the XCD-contiguous order.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fwrapv -ffp-contract=off -fno-slp-vectorize -w -Iinclude -Iidsp_amd/csrc tools/exp_fm_misaligned.hip -o build/exp_fm_mis
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

template <bool XCDC>
float run(const P::Params &prm, uint32_t *st, const int32_t *x, int32_t *y, size_t lanes, size_t frames, size_t pitch)
{
    constexpr size_t bytes = (size_t(7) * kLdsT * kFmBlock + 2 * kLdsT * kFmBlock) * 4;
    auto k = stream_frame_major_lds<P, 7, 1, false, XCDC>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    const unsigned grid = unsigned((lanes + kFmBlock - 1) / kFmBlock);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int i = 0; i < 40; i++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(grid), dim3(kFmBlock), bytes, 0, prm, st, x, y, lanes, frames, pitch, pitch, lanes);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 20) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main()
{
    const size_t frames = 4096;
    P::Params prm{};
    prm.sec[0] = {{1 << 26, 1 << 27, 1 << 26, 1 << 29, -(1 << 28)}, 30, 0, 0, 0};
    int32_t *x, *y;
    uint32_t *st;
    CK(hipMalloc(&x, 66000 * frames * 4));
    CK(hipMalloc(&y, 66000 * frames * 4));
    CK(hipMalloc(&st, 66000 * 16));
    CK(hipMemset(x, 1, 66000 * frames * 4));
    CK(hipMemset(st, 0, 66000 * 16));
    const char *variant = "xcd-contiguous";
    for (auto sh : std::vector<std::pair<size_t, size_t>>{{65536, 65536}, {65536, 65544}, {65536, 65540}, {65000, 65000}, {65532, 65532}}) {
        const float p = run<false>(prm, st, x, y, sh.first, frames, sh.second), r = run<true>(prm, st, x, y, sh.first, frames, sh.second);
        const double gb = double(sh.first) * frames * 8 / 1e9;
        printf("{\"variant\": \"%s\", \"lanes\": %zu, \"pitch\": %zu, \"plain_frac\": %.3f, \"variant_frac\": %.3f}\n", variant, sh.first, sh.second, gb / (p * 1e-3) / 8000,
               gb / (r * 1e-3) / 8000);
        fflush(stdout);
    }
    return 0;
}
