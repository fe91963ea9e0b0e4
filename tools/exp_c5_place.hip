// exp_c5_place.hip — round-2 experiment: how much does the placement of the output buffer move the LDS-DMA kernel at
// the C5 shard shapes?  One input buffer, eight separately allocated output buffers (with odd-sized dummy allocations
// in between so that they land at different offsets), every ring depth 4..8.
//   build/exp_c5_place <lanes> <LPT 1|2|4> <grid (0 = lanes / 256 / LPT)>
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

template <int NB, int LPT>
float run(const P::Params &prm, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames, unsigned grid)
{
    constexpr size_t bytes = (size_t(NB) * kLdsT * kFmBlock + 2 * kLdsT * kFmBlock) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_frame_major_lds<P, NB, LPT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int i = 0; i < 24; i++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((stream_frame_major_lds<P, NB, LPT, false>), dim3(grid), dim3(kFmBlock), bytes, 0, prm, st, x, y, lanes, frames, lanes, lanes, lanes);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 12) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

template <int LPT>
void sweep(const P::Params &prm, uint32_t *st, const float *x, std::vector<float *> &ys, size_t lanes, size_t frames, unsigned grid)
{
    const double gb = double(lanes) * frames * 8 / 1e9;
    auto row = [&](int nb, auto f) {
        printf("{\"lanes\": %zu, \"lpt\": %d, \"grid\": %u, \"nb\": %d, \"frac\": [", lanes, LPT, grid, nb);
        float worst = 0, best = 1e9;
        for (size_t k = 0; k < ys.size(); k++) {
            const float ms = f(ys[k]);
            worst = std::max(worst, ms), best = std::min(best, ms);
            printf("%s%.3f", k ? ", " : "", gb / (ms * 1e-3) / 8000);
        }
        printf("], \"worst\": %.3f, \"best\": %.3f}\n", gb / (worst * 1e-3) / 8000, gb / (best * 1e-3) / 8000);
        fflush(stdout);
    };
    row(4, [&](float *y) { return run<4, LPT>(prm, st, x, y, lanes, frames, grid); });
    row(5, [&](float *y) { return run<5, LPT>(prm, st, x, y, lanes, frames, grid); });
    row(6, [&](float *y) { return run<6, LPT>(prm, st, x, y, lanes, frames, grid); });
    row(7, [&](float *y) { return run<7, LPT>(prm, st, x, y, lanes, frames, grid); });
    row(8, [&](float *y) { return run<8, LPT>(prm, st, x, y, lanes, frames, grid); });
}

int main(int argc, char **argv)
{
    const size_t lanes = argc > 1 ? atoll(argv[1]) : 131072, frames = 4096;
    const int lpt = argc > 2 ? atoi(argv[2]) : 2;
    unsigned grid = argc > 3 ? atoi(argv[3]) : 0;
    if (!grid) grid = unsigned(lanes / kFmBlock / lpt);
    const size_t n = lanes * frames * 4;
    float *x;
    uint32_t *st;
    CK(hipMalloc(&x, n));
    CK(hipMalloc(&st, lanes * 8));
    CK(hipMemset(x, 0, n));
    CK(hipMemset(st, 0, lanes * 8));
    std::vector<float *> ys;
    for (int k = 0; k < 8; k++) {
        void *dummy;
        CK(hipMalloc(&dummy, size_t(k + 1) * ((3u << 20) + 4096 * 17)));  // shifts the next allocation
        float *y;
        CK(hipMalloc(&y, n));
        ys.push_back(y);
    }
    ys.push_back(x);  // in place
    P::Params prm{};
    prm.sec[0] = {{0.001f, 0.002f, 0.001f, 1.9f, -0.91f}, 0.f, -1e30f, 1e30f};
    if (lpt == 1) sweep<1>(prm, st, x, ys, lanes, frames, grid);
    else if (lpt == 2) sweep<2>(prm, st, x, ys, lanes, frames, grid);
    else sweep<4>(prm, st, x, ys, lanes, frames, grid);
    return 0;
}
