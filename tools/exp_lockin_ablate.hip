// What bounds the LaneMajor lock-in against the FrameMajor one?  Builds idsp_amd/csrc/lockin_waves.h as it ships and with its global
// stores and / or its input requests switched off by conditions that are never true at run time (IDSP_LW_ABL_NOSTORE,
// IDSP_LW_ABL_NOLOAD: the instruction stream stays), and times the C4 shape (32768 lanes x 4096 frames, [Lowpass<2>; 2], 4 waves,
// 16-frame batches) in both layouts.  tools/exp_lockin_ablate.sh builds the four variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fwrapv -ffp-contract=off -fno-slp-vectorize -Iinclude -Iidsp_amd/csrc [-DIDSP_LW_ABL_...] tools/exp_lockin_ablate.hip -o build/exp_lockin_ablate_<variant>
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

#ifndef VARIANT
#define VARIANT "product"
#endif

template <int IN, int B>
int run(size_t lanes, size_t frames, const char *name, unsigned sk_ticks = 0, unsigned sk_mod = 1, unsigned sk_shift = 0, int pattern = 0)
{
#ifdef IDSP_LW_ABL_SKEW
    {
        // workgroup b runs on XCD b % 8, CU slot (b >> 3) % 32 of it; b and b + 256 share a CU
        std::vector<unsigned> sk(4096);
        for (unsigned b = 0; b < 4096; b++) {
            const unsigned g = (b >> sk_shift) % sk_mod, second = (b >> 8) & 1;
            unsigned steps = g;
            if (pattern == 1) steps = second ? g : 0;                  // only the second workgroup of a CU waits
            if (pattern == 2) steps = second ? (g + sk_mod / 2) % sk_mod : g;  // the two of a CU half a cycle apart
            if (pattern == 3) steps = second ? g + 1 : 0;              // the second waits at least one step
            sk[b] = steps * sk_ticks;
        }
        CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_lw_skew), sk.data(), sk.size() * 4));
    }
#endif
    using Out = typename LwOut<MODE_IQ>::type;
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
    {
        std::vector<uint32_t> hx(lanes * frames);
        uint32_t s = 12345;
        for (auto &v : hx) s = s * 1664525u + 1013904223u, v = s >> 3;
        CHK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    }
    auto k = lockin_waves_kernel<LpBank<2, 2>, 4, IN, MODE_IQ, B>;
    const unsigned arg_skew = getenv("ABL_ARG_SKEW") ? unsigned(atoi(getenv("ABL_ARG_SKEW"))) : 0u;  // the product's stagger pattern
    hipFuncAttributes fa;
    CHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const dim3 grid(unsigned(lanes / 64)), block(4 * 64);
    // 250 ms of launches first: the clock settles
    for (int i = 0; i < 600; i++) hipLaunchKernelGGL(k, grid, block, 0, 0, p, st, x, y, lanes, frames, static_cast<const int32_t *>(nullptr), arg_skew, frames);
    CHK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 5; rep++) {
        CHK(hipEventRecord(e0));
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k, grid, block, 0, 0, p, st, x, y, lanes, frames, static_cast<const int32_t *>(nullptr), arg_skew, frames);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 20;
        best = ms < best ? ms : best, sum += ms;
    }
    std::printf("{\"variant\": \"%s\", \"form\": \"%s\", \"skew\": [%u, %u, %u, %d], \"lanes\": %zu, \"frames\": %zu, \"ms_mean\": %.4f, \"ms_min\": %.4f, \"frac\": %.3f, \"vgprs\": %d, \"scratch\": %zu}\n", VARIANT, name, sk_ticks, sk_mod, sk_shift, pattern, lanes, frames,
                sum / 5, best, double(lanes) * double(frames) * 12.0 / (sum / 5 * 1e-3) / 8e12, fa.numRegs, fa.localSizeBytes);
    (void)hipFree(st), (void)hipFree(x), (void)hipFree(y);
    return 0;
}

int main()
{
    run<IN_LM_DMA, 16>(32768, 4096, "LM dma");
    run<IN_FM_DMA, 16>(32768, 4096, "FM dma");
#ifdef IDSP_LW_ABL_SKEW
    // an interval of 16 frames is ~170 ticks
    if (const char *e = getenv("ABL_SKEW")) {  // "ticks,mod,shift,pattern;..."
        unsigned t, m, sh;
        int pat, n = 0;
        while (sscanf(e, "%u,%u,%u,%d%n", &t, &m, &sh, &pat, &n) == 4) {
            run<IN_LM_DMA, 16>(32768, 4096, "LM dma", t, m, sh, pat);
            e += n;
            if (*e == ';') e++;
        }
    }
#endif
    if (getenv("ABL_PITCH"))  // LaneMajor rows at pitches that are not powers of two
        for (size_t fr : {4112, 4128, 4160, 4224, 4352, 4608}) run<IN_LM_DMA, 16>(32768, fr, "LM dma");
    return 0;
}
