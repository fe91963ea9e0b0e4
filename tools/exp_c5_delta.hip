// exp_c5_delta.hip — round-2 experiment: LDS-DMA kernel on a persistent grid of 256 workgroups, output at x + size + delta
// inside ONE allocation, delta swept; is the placement sensitivity of profiles/r02_exp_c5_place.jsonl a function of the
// virtual offset?   build/exp_c5_delta <lanes> <NB 6|7|8>
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
using P = bq::Chain<bq::Df2tF32<false>, 1>;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NB>
float run(const P::Params &prm, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames, unsigned grid)
{
    constexpr size_t bytes = (size_t(NB) * kLdsT * kFmBlock + 2 * kLdsT * kFmBlock) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_frame_major_lds<P, NB, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int i = 0; i < 14; i++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((stream_frame_major_lds<P, NB, 1, false>), dim3(grid), dim3(kFmBlock), bytes, 0, prm, st, x, y, lanes, frames, lanes, lanes, lanes);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 6) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char **argv)
{
    const size_t lanes = argc > 1 ? atoll(argv[1]) : 1048576, frames = 4096;
    const int nb = argc > 2 ? atoi(argv[2]) : 7;
    const size_t n = lanes * frames * 4;
    char *buf;
    uint32_t *st;
    CK(hipMalloc(&buf, 2 * n + (64u << 20)));
    CK(hipMalloc(&st, lanes * 8));
    CK(hipMemset(buf, 0, n));
    CK(hipMemset(st, 0, lanes * 8));
    P::Params prm{};
    prm.sec[0] = {{0.001f, 0.002f, 0.001f, 1.9f, -0.91f}, 0.f, -1e30f, 1e30f};
    const double gb = double(lanes) * frames * 8 / 1e9;
    const size_t deltas[] = {0, 4096, 65536, 262144, 524288, 1u << 20, 3u << 19, 2u << 20, 5u << 19, 3u << 20, 4u << 20, 6u << 20, 8u << 20, 12u << 20, 16u << 20, 24u << 20, 32u << 20, (32u << 20) + 65536, 48u << 20};
    printf("{\"lanes\": %zu, \"nb\": %d, \"x\": \"%p\", \"frac_by_delta\": {", lanes, nb, (void *)buf);
    bool first = true;
    for (size_t d : deltas) {
        float *y = reinterpret_cast<float *>(buf + n + d);
        const float ms = nb == 6 ? run<6>(prm, st, (const float *)buf, y, lanes, frames, 256) : nb == 8 ? run<8>(prm, st, (const float *)buf, y, lanes, frames, 256)
                                                                                                       : run<7>(prm, st, (const float *)buf, y, lanes, frames, 256);
        printf("%s\"%zu\": %.3f", first ? "" : ", ", d, gb / (ms * 1e-3) / 8000);
        first = false;
        fflush(stdout);
    }
    printf("}}\n");
    return 0;
}
