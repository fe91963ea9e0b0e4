// tune_lds.hip — per-processor tuning of the LDS-DMA FrameMajor kernel (round 2): ring depth NB x addressing form RUN
// at the C2 shape (65536 lanes x 4096 frames), four placements of the output (adjacent to x, own allocation, 48 KiB
// further, in place).  Prints one JSON line per (processor, NB, RUN) with the four times; the choice written into
// P::LDS_RING / P::LDS_RUN is the combination with the best WORST placement.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fwrapv -fno-slp-vectorize -Iinclude -Iidsp_amd/csrc \
//         tools/tune_lds.hip -o build/tune_lds
//   build/tune_lds [processor index, default all]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "biquad_sections.h"

namespace idsp {
char *last_error_buf() { static thread_local char b[512]; return b; }
int fail(int code, const char *, ...) { return code; }
void note_kernel(const char *, const char *) {}
}  // namespace idsp

using namespace idsp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr size_t kLanes = 65536, kFrames = 4096;

template <class P, int NB, bool RUN>
float run1(const typename P::Params &prm, uint32_t *st, const typename P::In *x, typename P::Out *y)
{
    constexpr size_t bytes = (size_t(NB) * kLdsT * kFmBlock + 2 * kLdsT * kFmBlock) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_frame_major_lds<P, NB, 1, RUN>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int i = 0; i < 50; i++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((stream_frame_major_lds<P, NB, 1, RUN>), dim3(kLanes / kFmBlock), dim3(kFmBlock), bytes, 0, prm, st, x, y, kLanes, kFrames, kLanes, kLanes, kLanes);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 30) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

// the register-window kernel the launcher uses for processors above the LDS cost limit (same four placements)
template <class P>
float run_regwin(const typename P::Params &prm, uint32_t *st, const typename P::In *x, typename P::Out *y)
{
    constexpr int U = MaxU<P>::value;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int i = 0; i < 50; i++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((stream_frame_major<P, U>), dim3(kLanes / kFmBlock), dim3(kFmBlock), 0, 0, prm, st, x, y, kLanes, kFrames, kLanes, kLanes, 0);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 30) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

struct Bufs {
    char *x, *yadj, *yown, *y48;
    uint32_t *st;
};

template <class P, int NB, bool RUN>
void combo(const char *name, const typename P::Params &prm, const Bufs &b)
{
    using In = typename P::In;
    using Out = typename P::Out;
    const float t0 = run1<P, NB, RUN>(prm, b.st, reinterpret_cast<const In *>(b.x), reinterpret_cast<Out *>(b.yadj));
    const float t1 = run1<P, NB, RUN>(prm, b.st, reinterpret_cast<const In *>(b.x), reinterpret_cast<Out *>(b.yown));
    const float t2 = run1<P, NB, RUN>(prm, b.st, reinterpret_cast<const In *>(b.x), reinterpret_cast<Out *>(b.y48));
    const float t3 = run1<P, NB, RUN>(prm, b.st, reinterpret_cast<const In *>(b.x), reinterpret_cast<Out *>(b.x));
    const float worst = std::max(std::max(t0, t1), std::max(t2, t3));
    const double gb = double(kLanes) * kFrames * 8 / 1e9;
    printf("{\"proc\": \"%s\", \"nb\": %d, \"run\": %d, \"ms\": [%.4f, %.4f, %.4f, %.4f], \"worst_frac\": %.3f, \"best_frac\": %.3f}\n", name, NB, int(RUN),
           t0, t1, t2, t3, gb / (worst * 1e-3) / 8000, gb / (std::min(std::min(t0, t1), std::min(t2, t3)) * 1e-3) / 8000);
    fflush(stdout);
}

template <class P>
void sweep(const char *name, const typename P::Params &prm, const Bufs &b)
{
    {
        using In = typename P::In;
        using Out = typename P::Out;
        const float t0 = run_regwin<P>(prm, b.st, reinterpret_cast<const In *>(b.x), reinterpret_cast<Out *>(b.yadj));
        const float t1 = run_regwin<P>(prm, b.st, reinterpret_cast<const In *>(b.x), reinterpret_cast<Out *>(b.yown));
        const float t2 = run_regwin<P>(prm, b.st, reinterpret_cast<const In *>(b.x), reinterpret_cast<Out *>(b.y48));
        const float t3 = run_regwin<P>(prm, b.st, reinterpret_cast<const In *>(b.x), reinterpret_cast<Out *>(b.x));
        const double gb = double(kLanes) * kFrames * 8 / 1e9;
        const float worst = std::max(std::max(t0, t1), std::max(t2, t3)), best = std::min(std::min(t0, t1), std::min(t2, t3));
        printf("{\"proc\": \"%s\", \"nb\": 0, \"run\": -1, \"kernel\": \"register window\", \"ms\": [%.4f, %.4f, %.4f, %.4f], \"worst_frac\": %.3f, \"best_frac\": %.3f}\n", name,
               t0, t1, t2, t3, gb / (worst * 1e-3) / 8000, gb / (best * 1e-3) / 8000);
    }
    combo<P, 4, false>(name, prm, b);
    combo<P, 5, false>(name, prm, b);
    combo<P, 6, false>(name, prm, b);
    combo<P, 7, false>(name, prm, b);
    combo<P, 8, false>(name, prm, b);
    combo<P, 4, true>(name, prm, b);
    combo<P, 5, true>(name, prm, b);
    combo<P, 6, true>(name, prm, b);
    combo<P, 7, true>(name, prm, b);
    combo<P, 8, true>(name, prm, b);
}

template <class SecP, int N>
bq::ChainParams<SecP, N> params_n()
{
    bq::ChainParams<SecP, N> p{};
    for (int k = 0; k < N; k++) {
        if constexpr (std::is_same<SecP, bq::SecI32>::value) {
            p.sec[k] = {{1 << 20, 1 << 21, 1 << 20, 1 << 30, -(1 << 29)}, 30, 3, -(1 << 30), 1 << 30};
        } else {
            p.sec[k] = {{0.001f, 0.002f, 0.001f, 1.9f, -0.91f}, 0.01f, -10.f, 10.f};
        }
    }
    return p;
}
template <class Sec>
bq::ChainParams<typename Sec::Sec, 1> params()
{
    return params_n<typename Sec::Sec, 1>();
}

int main(int argc, char **argv)
{
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    const size_t n = kLanes * kFrames * 4;
    Bufs b;
    CK(hipMalloc(&b.x, 2 * n + (1 << 20)));
    b.yadj = b.x + n;
    b.y48 = b.x + n + 49152;
    CK(hipMalloc(&b.yown, n));
    CK(hipMalloc(&b.st, kLanes * 256));
    CK(hipMemset(b.x, 1, n));
    CK(hipMemset(b.st, 0, kLanes * 256));
    int k = 0;
#define SWEEP(SEC) if (only < 0 || only == k) sweep<bq::Chain<bq::SEC, 1>>(#SEC, params<bq::SEC>(), b); k++;
    SWEEP(Df1I32<false>)
    SWEEP(Df1I32<true>)
    SWEEP(DitherI32<false>)
    SWEEP(DitherI32<true>)
    SWEEP(WideI32<false>)
    SWEEP(WideI32<true>)
    SWEEP(Df1F32<false>)
    SWEEP(Df1F32<true>)
    SWEEP(Df2tF32<false>)
    SWEEP(Df2tF32<true>)
    SWEEP(NormalI32)
    SWEEP(NormalF32)
#define SWEEPN(SEC, N) if (only < 0 || only == k) sweep<bq::Chain<bq::SEC, N>>(#SEC " x" #N, params_n<bq::SEC::Sec, N>(), b); k++;
    SWEEPN(Df1I32<false>, 2)
    SWEEPN(Df1F32<false>, 2)
    SWEEPN(Df1F32<false>, 4)
    SWEEPN(Df2tF32<false>, 2)
    SWEEPN(Df2tF32<false>, 4)
#define SWEEPC(T, SECP, N) if (only < 0 || only == k) sweep<bq::CascadeDf1<T, N>>("CascadeDf1<" #T "> x" #N, params_n<bq::SECP, N>(), b); k++;
    SWEEPC(int32_t, SecI32, 2)
    SWEEPC(float, SecF32, 2)
    SWEEPC(float, SecF32, 4)
    SWEEPN(Df1I32<false>, 3)
    SWEEPN(Df1I32<false>, 4)
    SWEEPN(Df1I32<true>, 2)
    SWEEPN(WideI32<false>, 2)
    SWEEPC(int32_t, SecI32, 3)
    SWEEPC(int32_t, SecI32, 4)
    SWEEPC(int32_t, SecI32, 8)
    SWEEPN(NormalI32, 2)
    SWEEPN(DitherI32<false>, 2)
    SWEEPN(Df1F32<false>, 3)
    SWEEPN(NormalF32, 2)
    SWEEPN(NormalF32, 4)
    return 0;
}
