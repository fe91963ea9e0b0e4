// tune_fm_small.hip — FrameMajor at lane counts that do not fill the chip: stream_frame_major_staged<P, LW> (LW = 64 / 32 / 16
// lanes per wave) against the register-window kernel in single-wave workgroups (what the launcher used there) and the
// LDS-DMA kernel; every staged run is first compared bit for bit with the register-window kernel.  One JSON line per case.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fwrapv -fno-slp-vectorize -Iinclude -Iidsp_amd/csrc \
//         tools/tune_fm_small.hip -o build/tune_fm_small
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "biquad_sections.h"

namespace idsp {
char *last_error_buf() { static thread_local char b[512]; return b; }
int fail(int code, const char *, ...) { return code; }
void note_kernel(const char *, const char *) {}
}  // namespace idsp
using namespace idsp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape { size_t lanes, frames, pitch; };
struct Bufs { char *x, *y, *yref; uint32_t *st, *stref; size_t cap; };

template <class F>
float timeit(F &&launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int i = 0; i < 30; i++) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 10) ts.push_back(ms);
    }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

template <class P, int LW>
void one(const char *name, const typename P::Params &prm, const Bufs &b, const Shape &sh, float tref, float tlds, bool inplace)
{
    using In = typename P::In;
    using Out = typename P::Out;
    const size_t bytes = kFmStagedTile + P::LDS_WORDS * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_frame_major_staged<P, LW>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    const unsigned grid = unsigned((sh.lanes + LW - 1) / LW);
    const size_t n = sh.frames * sh.pitch * sizeof(In);
    char *yy = b.y;
    auto launch = [&]() {
        hipLaunchKernelGGL((stream_frame_major_staged<P, LW>), dim3(grid), dim3(kWave), bytes, 0, prm, b.st, reinterpret_cast<const In *>(inplace ? yy : b.x),
                           reinterpret_cast<Out *>(yy), sh.lanes, sh.frames, sh.pitch, sh.pitch, sh.lanes);
    };
    CK(hipMemset(b.st, 0, sh.lanes * 256));
    if (inplace) CK(hipMemcpy(yy, b.x, n, hipMemcpyDeviceToDevice)); else CK(hipMemset(yy, 0xEE, n));
    launch();
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> got(n / 4), want(n / 4), sg(sh.lanes * 64), sw(sh.lanes * 64);
    CK(hipMemcpy(got.data(), yy, n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(want.data(), b.yref, n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sg.data(), b.st, sh.lanes * 256, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sw.data(), b.stref, sh.lanes * 256, hipMemcpyDeviceToHost));
    bool ok = sg == sw;
    for (size_t f = 0; f < sh.frames && ok; f++)
        ok = memcmp(&got[f * sh.pitch * sizeof(In) / 4], &want[f * sh.pitch * sizeof(In) / 4], sh.lanes * sizeof(In)) == 0;
    const float t = timeit(launch);
    const double gb = double(sh.lanes) * sh.frames * (sizeof(In) + sizeof(Out)) / 1e9;
    printf("{\"proc\": \"%s\", \"lanes\": %zu, \"frames\": %zu, \"pitch\": %zu, \"lw\": %d, \"inplace\": %d, \"ok\": %s, \"ms\": %.4f, \"frac\": %.3f, \"regwin_ms\": %.4f, \"lds_dma_ms\": %.4f}\n",
           name, sh.lanes, sh.frames, sh.pitch, LW, int(inplace), ok ? "true" : "false", t, gb / (t * 1e-3) / 8000, tref, tlds);
    fflush(stdout);
}

template <class P>
void sweep(const char *name, const typename P::Params &prm, const Bufs &b, const std::vector<Shape> &shapes)
{
    using In = typename P::In;
    using Out = typename P::Out;
    for (const Shape &sh : shapes) {
        if (sh.frames * sh.pitch * sizeof(In) > b.cap) continue;
        constexpr int U = MaxU<P>::value;
        const unsigned grid = unsigned((sh.lanes + kWave - 1) / kWave);
        auto ref = [&]() {
            hipLaunchKernelGGL((stream_frame_major<P, U>), dim3(grid), dim3(kWave), 0, 0, prm, b.stref, reinterpret_cast<const In *>(b.x),
                               reinterpret_cast<Out *>(b.yref), sh.lanes, sh.frames, sh.pitch, sh.pitch, 0);
        };
        const float tref = timeit(ref);
        float tlds = 0;
        if constexpr (sizeof(In) == 4) {
            if (sh.lanes % kFmBlock == 0 && sh.pitch % 4 == 0) {
                constexpr int NB = LdsRingOf<P>::value;
                constexpr bool RUN = LdsRunOf<P>::value;
                constexpr size_t bytes = (size_t(NB) * kLdsT * kFmBlock + 2 * kLdsT * kFmBlock + P::LDS_WORDS) * 4;
                CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_frame_major_lds<P, NB, 1, RUN>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
                auto lds = [&]() {
                    hipLaunchKernelGGL((stream_frame_major_lds<P, NB, 1, RUN>), dim3(unsigned(sh.lanes / kFmBlock)), dim3(kFmBlock), bytes, 0, prm, b.stref,
                                       reinterpret_cast<const In *>(b.x), reinterpret_cast<Out *>(b.yref), sh.lanes, sh.frames, sh.pitch, sh.pitch, sh.lanes);
                };
                tlds = timeit(lds);
            }
        }
        CK(hipMemset(b.stref, 0, sh.lanes * 256));
        CK(hipMemset(b.yref, 0xEE, sh.frames * sh.pitch * sizeof(In)));
        ref();
        CK(hipDeviceSynchronize());
        one<P, 64>(name, prm, b, sh, tref, tlds, false);
        one<P, 64>(name, prm, b, sh, tref, tlds, true);
        one<P, 32>(name, prm, b, sh, tref, tlds, false);
        one<P, 32>(name, prm, b, sh, tref, tlds, true);
        one<P, 16>(name, prm, b, sh, tref, tlds, false);
        one<P, 16>(name, prm, b, sh, tref, tlds, true);
    }
}

template <class SecP, int N>
bq::ChainParams<SecP, N> params_n()
{
    bq::ChainParams<SecP, N> p{};
    for (int k = 0; k < N; k++) {
        if constexpr (std::is_same<SecP, bq::SecI32>::value) {
            p.sec[k] = {{1 << 20, 1 << 21, 1 << 20, 1 << 30, -(1 << 29)}, 30, 3, -(1 << 30), 1 << 30};
        } else if constexpr (std::is_same<SecP, bq::SecF32>::value) {
            p.sec[k] = {{0.001f, 0.002f, 0.001f, 1.9f, -0.91f}, 0.01f, -10.f, 10.f};
        } else {
            p.sec[k] = {{0.001, 0.002, 0.001, 1.9, -0.91}, 0.01, -10., 10.};
        }
    }
    return p;
}

int main(int argc, char **argv)
{
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    Bufs b;
    b.cap = size_t(2) << 30;
    CK(hipMalloc(&b.x, b.cap)); CK(hipMalloc(&b.y, b.cap)); CK(hipMalloc(&b.yref, b.cap));
    CK(hipMalloc(&b.st, size_t(1) << 26)); CK(hipMalloc(&b.stref, size_t(1) << 26));
    {
        std::vector<uint32_t> h(size_t(64) << 20);
        std::mt19937 g(1);
        for (auto &v : h) v = g() >> 12;
        for (size_t o = 0; o < b.cap; o += h.size() * 4) CK(hipMemcpy(b.x + o, h.data(), std::min(h.size() * 4, b.cap - o), hipMemcpyHostToDevice));
    }
    const std::vector<Shape> shapes = {
        {1024, 65536, 1024}, {4096, 16384, 4096}, {8192, 8192, 8192}, {16384, 4096, 16384}, {32768, 4096, 32768}, {49152, 4096, 49152},
        {65536, 4096, 65536}, {16380, 1000, 16380}, {1000, 77, 1000}, {100, 131, 104}, {4, 300, 4}, {68, 1025, 72},
    };
    int k = 0;
#define SWEEPN(SEC, N) if (only < 0 || only == k) sweep<bq::Chain<bq::SEC, N>>(#SEC " x" #N, params_n<bq::SEC::Sec, N>(), b, shapes); k++;
    SWEEPN(Df1I32<false>, 1)
    SWEEPN(Df2tF32<false>, 1)
    SWEEPN(Df1I32<true>, 4)
    SWEEPN(Df1F64<false>, 1)
#define SWEEPC(T, SECP, N) if (only < 0 || only == k) sweep<bq::CascadeDf1<T, N>>("CascadeDf1<" #T "> x" #N, params_n<bq::SECP, N>(), b, shapes); k++;
    SWEEPC(int32_t, SecI32, 8)
    return 0;
}
