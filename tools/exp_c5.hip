// exp_c5.hip — experiment (round 2): what bounds the LDS-DMA FrameMajor kernel beyond 65536 lanes?
// Runs the library's own kernel (stream_frame_major_lds<Chain<Df1I32>, NB>) standalone with the lane
// count, the row pitch, the grid (persistent workgroups) and the ring depth as free parameters.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fwrapv -fno-slp-vectorize -Iinclude -Iidsp_amd/csrc \
//         tools/exp_c5.hip -o build/exp_c5
//   build/exp_c5 <lanes> <frames> <pitch (0 = lanes)> <grid (0 = lanes/256)> <NB 3..8> [in-place 0/1] [y offset bytes] [LPT 1 2 4 8 16]
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
#ifndef EXP_RUN
#define EXP_RUN false
#endif
#ifdef EXP_CLAMP
using P = bq::Chain<bq::Df1I32<true>, 1>;
#else
using P = bq::Chain<bq::Df1I32<false>, 1>;
#endif
static int g_adj = 0;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NB, int LPT = 1>
float run(const P::Params &prm, uint32_t *st, const int32_t *x, int32_t *y, size_t lanes, size_t frames, size_t pitch, unsigned grid, int iters, int adj = 0)
{
    constexpr size_t kSeg = (LPT > kLdsT) ? LPT : kLdsT;
    constexpr size_t bytes = (size_t(NB) * kSeg * kFmBlock + 2 * kSeg * kFmBlock) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_frame_major_lds<P, NB, LPT, EXP_RUN>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int i = 0; i < iters + 40; i++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((stream_frame_major_lds<P, NB, LPT, EXP_RUN>), dim3(grid), dim3(kFmBlock), bytes, 0, prm, st, x, y, lanes, frames, pitch, pitch, lanes);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 40) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char **argv)
{
    const size_t lanes = argc > 1 ? atoll(argv[1]) : 65536, frames = argc > 2 ? atoll(argv[2]) : 4096;
    size_t pitch = argc > 3 ? atoll(argv[3]) : 0;
    unsigned grid = argc > 4 ? atoi(argv[4]) : 0;
    const int nb = argc > 5 ? atoi(argv[5]) : 7;
    const bool inplace = argc > 6 && atoi(argv[6]);
    const size_t yoff = argc > 7 ? size_t(atoll(argv[7])) : 0;
    const int lpt = argc > 8 ? atoi(argv[8]) : 1;
    g_adj = argc > 9 ? atoi(argv[9]) : 0;
    if (!pitch) pitch = lanes;
    if (!grid) grid = unsigned(lanes / kFmBlock / lpt);
    const size_t n = pitch * frames;
    int32_t *x, *y;
    uint32_t *st;
    const bool separate = yoff == size_t(-1);  // y from its own hipMalloc (what two framework allocations look like)
    CK(hipMalloc(&x, n * 4 + (inplace || separate ? 0 : n * 4 + yoff + 4096)));
    y = inplace ? x : reinterpret_cast<int32_t *>(reinterpret_cast<char *>(x) + n * 4 + yoff);
    if (separate) CK(hipMalloc(&y, n * 4));
    CK(hipMalloc(&st, lanes * 16));
    CK(hipMemset(x, 1, n * 4));
    CK(hipMemset(st, 0, lanes * 16));
    P::Params prm{};
    prm.sec[0] = {{1 << 20, 1 << 21, 1 << 20, 1 << 30, -(1 << 29)}, 30, 3, -(1 << 30), 1 << 30};
    float ms = 0;
    if (lpt > 1) {
        if (nb != 7 && nb != 5) { printf("LPT > 1: NB 5 or 7\n"); return 1; }
        switch (lpt * 10 + nb) {
            case 27: ms = run<7, 2>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
            case 47: ms = run<7, 4>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
            case 87: ms = run<7, 8>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
            case 167: ms = run<7, 16>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
            case 25: ms = run<5, 2>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
            case 45: ms = run<5, 4>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
            case 85: ms = run<5, 8>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
            case 165: ms = run<5, 16>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
            default: printf("LPT 2 4 8 16\n"); return 1;
        }
    } else
    switch (nb) {
        case 3: ms = run<3>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
        case 4: ms = run<4>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
        case 5: ms = run<5>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
        case 6: ms = run<6>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
        case 7: ms = run<7>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
        case 8: ms = run<8>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
        case 9: ms = run<9>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
        case 10: ms = run<10>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
        case 11: ms = run<11>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
        case 12: ms = run<12>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
        case 14: ms = run<14>(prm, st, x, y, lanes, frames, pitch, grid, 20); break;
        default: printf("NB 3..8\n"); return 1;
    }
    const double gbs = double(lanes) * frames * 8 / (ms * 1e-3) / 1e9;
    printf("{\"lanes\": %zu, \"frames\": %zu, \"pitch\": %zu, \"grid\": %u, \"nb\": %d, \"inplace\": %d, \"yoff\": %zu, \"lpt\": %d, \"adj\": %d, \"ms\": %.4f, \"GB/s\": %.0f, \"frac\": %.3f}\n",
           lanes, frames, pitch, grid, nb, int(inplace), yoff, lpt, g_adj, ms, gbs, gbs / 8000);
    return 0;
}
