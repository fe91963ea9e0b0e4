// The stage-wave lock-in kernel (lockin_stages_kernel, idsp_amd/csrc/lockin_waves.h) against lockin_waves_kernel: outputs and
// written-back state compared word for word on hashed inputs, then both timed (30 launches after 10).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fwrapv -ffp-contract=off -fno-slp-vectorize -Iinclude -Iidsp_amd/csrc tools/exp_lockin_stages.hip -o build/exp_lockin_stages
#include "lockin_waves.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
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

template <class K, class... A>
float time_ms(K k, dim3 grid, dim3 block, A... a)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k, grid, block, 0, 0, a...);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < 30; i++) hipLaunchKernelGGL(k, grid, block, 0, 0, a...);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms / 30 < best ? ms / 30 : best;
    }
    return best;
}

template <int N, int MODE, int G, int R = 2, int WA = 4, int BA = 16>
int run(size_t lanes, size_t frames)
{
    using Out = typename LwOut<MODE>::type;
    LpParams p{};
    for (int i = 0; i < 4; i++) p.k[i][0] = 10000 + 777 * i, p.k[i][1] = -9500000 - 1313 * i;
    const size_t nst = 2 + 2 * 2 * N * 2;  // words per lane: acc, step, two arms of [Lowpass<N>; 2]
    uint32_t *st_a, *st_b;
    int32_t *x;
    Out *ya, *yb;
    CHK(hipMalloc(&st_a, nst * lanes * 4));
    CHK(hipMalloc(&st_b, nst * lanes * 4));
    CHK(hipMalloc(&x, lanes * frames * 4));
    CHK(hipMalloc(&ya, lanes * frames * sizeof(Out)));
    CHK(hipMalloc(&yb, lanes * frames * sizeof(Out)));
    std::vector<uint32_t> hs(nst * lanes);
    for (size_t i = 0; i < hs.size(); i++) hs[i] = uint32_t(i * 2654435761u) ^ uint32_t(i >> 3);
    std::vector<int32_t> hx(lanes * frames);
    for (size_t i = 0; i < hx.size(); i++) hx[i] = int32_t(uint32_t(i * 0x9E3779B1u) ^ uint32_t(i >> 7) * 40503u) >> 3;
    CHK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(st_a, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(st_b, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemset(ya, 0x55, lanes * frames * sizeof(Out)));
    CHK(hipMemset(yb, 0xAA, lanes * frames * sizeof(Out)));
    auto ka = lockin_waves_kernel<LpBank<N, 2>, WA, IN_FM_DMA, MODE, BA>;
    auto kb = lockin_stages_kernel<N, MODE, G, R>;
    const dim3 ga(unsigned(lanes / 64)), ba(WA * 64), gb(unsigned(lanes / (64 * G))), bb((4 + R) * G * 64);
    // two consecutive calls each (state carried over), then compare
    for (int c = 0; c < 2; c++) {
        hipLaunchKernelGGL(ka, ga, ba, 0, 0, p, st_a, x, ya, lanes, frames, static_cast<const int32_t *>(nullptr), 0u, frames);
        hipLaunchKernelGGL(kb, gb, bb, 0, 0, p, st_b, x, yb, lanes, frames);
    }
    CHK(hipDeviceSynchronize());
    CHK(hipGetLastError());
    std::vector<Out> ha(lanes * frames), hb(lanes * frames);
    std::vector<uint32_t> sa(nst * lanes), sb(nst * lanes);
    CHK(hipMemcpy(ha.data(), ya, ha.size() * sizeof(Out), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(hb.data(), yb, hb.size() * sizeof(Out), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(sa.data(), st_a, sa.size() * 4, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(sb.data(), st_b, sb.size() * 4, hipMemcpyDeviceToHost));
    const size_t bad_y = std::memcmp(ha.data(), hb.data(), ha.size() * sizeof(Out)) ? 1 : 0;
    size_t nbad = 0;
    if (bad_y)
        for (size_t i = 0; i < ha.size(); i++) nbad += std::memcmp(&ha[i], &hb[i], sizeof(Out)) != 0;
    size_t bad_s = 0;
    for (size_t i = 0; i < sa.size(); i++) bad_s += sa[i] != sb[i];
    const float ta = time_ms(ka, ga, ba, p, st_a, x, ya, lanes, frames, static_cast<const int32_t *>(nullptr), 0u, frames), tb = time_ms(kb, gb, bb, p, st_b, x, yb, lanes, frames);
    const double bytes = double(lanes) * double(frames) * (4.0 + sizeof(Out));
    std::printf("{\"N\": %d, \"mode\": %d, \"G\": %d, \"R\": %d, \"waves_form\": \"%dw B%d\", \"lanes\": %zu, \"frames\": %zu, \"y_mismatches\": %zu, \"state_mismatches\": %zu, \"ms_waves\": %.4f, \"ms_stages\": %.4f, "
                "\"frac_waves\": %.3f, \"frac_stages\": %.3f}\n",
                N, MODE, G, R, WA, BA, lanes, frames, nbad, bad_s, ta, tb, bytes / (ta * 1e-3) / 8e12, bytes / (tb * 1e-3) / 8e12);
    hipFree(st_a), hipFree(st_b), hipFree(x), hipFree(ya), hipFree(yb);
    return 0;
}

int main(int argc, char **)
{
    if (argc > 4) {  // one lane group per workgroup at two workgroups per CU: do they share the CU? (build with IDSP_LS_RING=4: 62 KiB of LDS)
        run<2, MODE_IQ, 1, 4, 4, 16>(16384, 4096);
        run<2, MODE_IQ, 1, 4, 4, 16>(32768, 4096);
        run<2, MODE_IQ, 2, 4, 4, 16>(32768, 4096);
        run<2, MODE_IQ, 1, 4, 4, 16>(65536, 4096);
        run<2, MODE_ARG, 1, 4, 4, 16>(32768, 4096);
        run<2, MODE_ARG, 2, 4, 4, 16>(32768, 4096);
        return 0;
    }
    if (argc > 3) {  // C4 and its neighbours on the 4-wave kernel only (build variants: priorities, mixer placement, roles)
        run<2, MODE_IQ, 2, 4, 4, 16>(32768, 4096);
        run<2, MODE_IQ, 2, 4, 4, 16>(65536, 4096);
        run<2, MODE_NORM_SQR, 2, 4, 4, 16>(32768, 4096);
        run<1, MODE_IQ, 2, 4, 4, 16>(32768, 4096);
        return 0;
    }
    if (argc > 2) {  // the 4-wave kernel with 8- and 16-frame batches against the stage kernel above two workgroups per CU
        run<2, MODE_IQ, 2, 4, 4, 8>(49152, 4096);
        run<2, MODE_IQ, 2, 4, 4, 16>(49152, 4096);
        run<2, MODE_IQ, 2, 4, 4, 8>(65536, 4096);
        run<2, MODE_IQ, 2, 4, 4, 16>(65536, 4096);
        run<2, MODE_IQ, 2, 4, 4, 8>(98304, 4096);
        run<2, MODE_IQ, 2, 4, 4, 16>(98304, 4096);
        run<2, MODE_IQ, 2, 4, 4, 8>(131072, 2048);
        run<2, MODE_IQ, 2, 4, 4, 16>(131072, 2048);
        run<2, MODE_NORM_SQR, 2, 4, 4, 8>(65536, 4096);
        run<2, MODE_NORM_SQR, 2, 4, 4, 16>(65536, 4096);
        run<2, MODE_ARG, 2, 4, 4, 16>(65536, 4096);
        run<2, MODE_ARG, 2, 4, 4, 16>(131072, 2048);
        run<2, MODE_ARG, 2, 4, 4, 16>(196608, 2048);
        return 0;
    }
    if (argc > 1) {  // dispatch survey
        run<2, MODE_IQ, 1, 4>(4096, 4096);
        run<2, MODE_IQ, 1, 4>(8192, 4096);
        run<2, MODE_IQ, 1, 4>(16384, 4096);
        run<2, MODE_IQ, 1, 4>(24576, 4096);
        run<2, MODE_IQ, 2, 4>(24576, 4096);
        run<2, MODE_IQ, 1, 4>(32768, 1024);
        run<2, MODE_IQ, 2, 4>(49152, 4096);
        run<2, MODE_NORM_SQR, 1, 4>(16384, 4096);
        run<2, MODE_ARG, 1, 4, 6, 8>(8192, 4096);
        run<2, MODE_ARG, 1, 4, 6, 8>(16384, 4096);
        run<2, MODE_ARG, 2, 4, 6, 8>(16384, 4096);
        run<2, MODE_ARG, 1, 4, 6, 8>(32768, 4096);
        run<2, MODE_ARG, 2, 4, 6, 8>(32768, 4096);
        run<2, MODE_ARG, 2, 4, 4, 8>(65536, 4096);
        run<2, MODE_ARG, 2, 4, 4, 8>(131072, 2048);
        run<1, MODE_ARG, 2, 4, 6, 8>(32768, 4096);
        run<1, MODE_IQ, 1, 4>(16384, 4096);
        return 0;
    }
    run<2, MODE_IQ, 2, 4>(32768, 4096);
    run<2, MODE_IQ, 2, 2>(32768, 4096);
    run<2, MODE_IQ, 1, 4>(16384, 4096);
    run<2, MODE_IQ, 2, 4>(65536, 4096);
    run<2, MODE_NORM_SQR, 2, 4>(32768, 4096);
    run<2, MODE_ARG, 2, 4>(32768, 4096);
    run<2, MODE_ARG, 2, 4>(65536, 4096);
    run<1, MODE_IQ, 2, 4>(32768, 4096);
    run<2, MODE_IQ, 2, 4>(8192, 16);
    run<2, MODE_IQ, 1, 4>(4096 + 64, 48);
    return 0;
}
