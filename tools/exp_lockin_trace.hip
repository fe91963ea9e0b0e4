// Where does an interval of the multi-wave lock-in kernel go?  Builds idsp_amd/csrc/lockin_waves.h with IDSP_LW_TRACE (cycle
// counter sums per wave and phase, workgroup 0) and runs the C4 shape (32768 lanes x 4096 frames, [Lowpass<2>; 2], FrameMajor
// DMA input, 16-frame batches) plus the one-workgroup-per-CU shape (16384 lanes).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fwrapv -ffp-contract=off -fno-slp-vectorize -Iinclude -Iidsp_amd/csrc tools/exp_lockin_trace.hip -o build/exp_lockin_trace
#ifndef NO_TRACE
#define IDSP_LW_TRACE 1
#endif
#include "lockin_waves.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace idsp;

#define CHK(x)                                                  \
    do {                                                        \
        hipError_t e_ = (x);                                    \
        if (e_ != hipSuccess) {                                 \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                           \
        }                                                       \
    } while (0)

template <int W, int B, int MODE>
int run(size_t lanes, size_t frames, const char *name)
{
    using Out = typename LwOut<MODE>::type;
    LpParams p{};
    for (int i = 0; i < 4; i++) p.k[i][0] = 1 << 20, p.k[i][1] = -(1 << 27);
    uint32_t *st;
    int32_t *x;
    Out *y;
    CHK(hipMalloc(&st, 18 * lanes * 4));
    CHK(hipMalloc(&x, lanes * frames * 4));
    CHK(hipMalloc(&y, lanes * frames * sizeof(Out)));
    std::vector<uint32_t> hs(18 * lanes);
    for (size_t i = 0; i < hs.size(); i++) hs[i] = uint32_t(i * 2654435761u);
    CHK(hipMemcpy(st, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemset(x, 1, lanes * frames * 4));
    auto k = lockin_waves_kernel<LpBank<2, 2>, W, IN_FM_DMA, MODE, B>;
    int occ = 0;
    CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, W * 64, 0));
    hipFuncAttributes fa;
    CHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k)));
    std::printf("[%s] occupancy API: %d workgroups per CU; %d registers, %zu B static LDS, %d max threads per block\n", name, occ, fa.numRegs, fa.sharedSizeBytes, fa.maxThreadsPerBlock);
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k, dim3(unsigned(lanes / 64)), dim3(W * 64), 0, 0, p, st, x, y, lanes, frames, static_cast<const int32_t *>(nullptr), 0u, frames);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k, dim3(unsigned(lanes / 64)), dim3(W * 64), 0, 0, p, st, x, y, lanes, frames, static_cast<const int32_t *>(nullptr), 0u, frames);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double nint = double((frames + B - 1) / B);
    std::printf("%s: %zu lanes x %zu frames, W=%d B=%d: %.4f ms per launch = %.0f ns per interval\n", name, lanes, frames, W, B, ms / 10, ms / 10 * 1e6 / nint);
#ifdef IDSP_LW_TRACE
    unsigned long long tr[8][8];
    CHK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_lw_trace), sizeof(tr)));
    std::printf("  cycle-counter ticks per interval (s_memtime, 100 MHz): arm waves: [dma issue, row load, lowpass chain, row store, dma wait, barrier]; read-out waves: [x read, cossin+mix, row store, row load (arm out), global stores, barrier]\n");
    for (int w = 0; w < W; w++) {
        std::printf("  wave %d (%s):", w, w < 2 ? "arm" : "read-out");
        for (int i = 0; i < 6; i++) std::printf(" %8.1f", double(tr[w][i]) / nint);
        std::printf("   | whole kernel: %llu s_memtime ticks in %.1f us (s_memrealtime) = %.3f ticks per ns\n", tr[w][6], double(tr[w][7]) * 0.01, double(tr[w][6]) / (double(tr[w][7]) * 10.0));
    }
    {
        // every workgroup: duration, CU (XCC, SE, CU from HW_ID) and the SIMD of each wave
        const size_t nwg = lanes / 64;
        std::vector<unsigned long long> wg(4096 * 8 * 3);
        CHK(hipMemcpyFromSymbol(wg.data(), HIP_SYMBOL(g_lw_wg), wg.size() * 8));
        unsigned long long t0 = ~0ull;
        for (size_t b = 0; b < nwg; b++) t0 = wg[(b * 8) * 3] < t0 ? wg[(b * 8) * 3] : t0;
        if (getenv("LW_DUMP")) {
            for (size_t b = 0; b < nwg; b++) {
                const unsigned long long *e = &wg[(b * 8) * 3];
                const unsigned hw = unsigned(e[2]), xcc = unsigned(e[2] >> 32) & 0xf;
                std::printf("  wg %4zu xcc %u se %u cu %2u  start %7.1f us  end %7.1f us  simd of waves:", b, xcc, (hw >> 13) & 7, (hw >> 8) & 15, double(e[0] - t0) * 0.01, double(e[1] - t0) * 0.01);
                for (int w = 0; w < W; w++) std::printf(" %u", (unsigned(wg[(b * 8 + w) * 3 + 2]) >> 4) & 3);
                std::printf("\n");
            }
        }
        // histogram of end times
        double lo = 1e30, hi = 0, sum = 0;
        for (size_t b = 0; b < nwg; b++) {
            const double d = double(wg[(b * 8) * 3 + 1] - wg[(b * 8) * 3]) * 0.01;
            lo = d < lo ? d : lo, hi = d > hi ? d : hi, sum += d;
        }
        std::printf("  workgroup durations: min %.1f us, mean %.1f us, max %.1f us\n", lo, sum / double(nwg), hi);
    }
#endif
    hipFree(st), hipFree(x), hipFree(y);
    return 0;
}

int main()
{
    run<4, 16, MODE_IQ>(16384, 4096, "one workgroup per CU");
    run<4, 16, MODE_IQ>(32768, 4096, "C4");
    run<4, 8, MODE_IQ>(32768, 4096, "C4, 8-frame batches");
    run<6, 16, MODE_IQ>(32768, 4096, "C4, 6 waves");
    run<6, 16, MODE_ARG>(32768, 4096, "C4 arg, 6 waves");
    return 0;
}
