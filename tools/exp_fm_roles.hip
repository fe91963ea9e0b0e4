// exp_fm_roles.hip — round 5: the role-wave FrameMajor kernel (tools/fm_roles.h) against the shipped LDS-DMA kernel
// (stream_frame_major_lds), bit for bit first, then timed over several placements of the output buffer.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fwrapv -fno-slp-vectorize -Iinclude -Iidsp_amd/csrc -Itools \
//         tools/exp_fm_roles.hip -o build/exp_fm_roles
//   build/exp_fm_roles <i32|f32> <lanes> <frames> [iters]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "biquad_sections.h"
#include "fm_roles.h"  // tools/fm_roles.h: the role-wave experiment kernel (not part of the library)
#include "fm_sweep.h"

namespace idsp {
char *last_error_buf() { static thread_local char b[512]; return b; }
int fail(int code, const char *, ...) { return code; }
void note_kernel(const char *, const char *) {}
void note_kernel_also(const char *) {}
}  // namespace idsp

using namespace idsp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_kernel(uint32_t *p, size_t n, int is_float)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        uint32_t h = uint32_t(i) * 2654435761u ^ uint32_t(i >> 32) * 40503u;
        h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
        if (is_float)
            p[i] = __float_as_uint((float(int32_t(h)) * (1.0f / 2147483648.0f)));
        else
            p[i] = uint32_t(int32_t(h) >> 7);
    }
}
__global__ void diff_kernel(const uint32_t *a, const uint32_t *b, size_t n, unsigned long long *cnt)
{
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(cnt, c);
}

static int g_iters = 10;

template <class F>
static float time_ms(F &&launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int i = 0; i < g_iters + 3; i++) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 3) ts.push_back(ms);
    }
    CK(hipGetLastError());
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

static unsigned lib_grid(size_t lanes)
{
    const size_t wgs = (lanes + kFmBlock - 1) / kFmBlock;
    if (wgs <= kLdsGridCap) return unsigned(wgs);
    const size_t rounds = (wgs + 255) / 256;
    return unsigned((wgs + rounds - 1) / rounds);
}

template <class P>
struct Ctx {
    typename P::Params prm;
    uint32_t *st;
    const typename P::In *x;
    size_t lanes, frames;
};

template <class P>
static void launch_old(const Ctx<P> &c, typename P::Out *y)
{
    constexpr int NB = 7;
    constexpr size_t bytes = (size_t(NB) * kLdsT * kFmBlock + 2 * kLdsT * kFmBlock) * 4;
    static bool once = false;
    if (!once) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_frame_major_lds<P, NB, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
        once = true;
    }
    hipLaunchKernelGGL((stream_frame_major_lds<P, NB, 1, false>), dim3(lib_grid(c.lanes)), dim3(kFmBlock), bytes, 0, c.prm, c.st, c.x, y, c.lanes, c.frames, c.lanes,
                       c.lanes, c.lanes);
}

template <class P, int NB, int LPT, bool ILV>
static void launch_lpt(const Ctx<P> &c, typename P::Out *y)
{
    constexpr size_t ts = LPT > kLdsT ? LPT : kLdsT;
    constexpr size_t bytes = (size_t(NB) * ts * kFmBlock + 2 * ts * kFmBlock) * 4;
    static bool once = false;
    if (!once) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_frame_major_lds<P, NB, LPT, false, false, ILV>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
        once = true;
    }
    hipLaunchKernelGGL((stream_frame_major_lds<P, NB, LPT, false, false, ILV>), dim3(unsigned(c.lanes / kFmBlock / LPT)), dim3(kFmBlock), bytes, 0, c.prm, c.st, c.x, y,
                       c.lanes, c.frames, c.lanes, c.lanes, c.lanes);
}

static SweepGeom g_geom;
template <class P, int LPT, int NB, int FORM, int SLP = 0, int TSEG = 8>
static void launch_sweep_lpt(const Ctx<P> &c, typename P::Out *y)
{
    if constexpr (LPT == 1) {
        if (g_geom.fps > 1) {
            constexpr size_t bytes = (size_t(NB) * TSEG * kFmBlock + 2 * size_t(TSEG) * kFmBlock) * 4;
            static bool once1 = false;
            if (!once1) {
                CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_frame_major_sweep<P, 1, NB, FORM, SLP, TSEG, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
                once1 = true;
            }
            hipLaunchKernelGGL((stream_frame_major_sweep<P, 1, NB, FORM, SLP, TSEG, true>), dim3(g_geom.grid), dim3(kFmBlock), bytes, 0, c.prm, c.st, c.x, y, c.lanes, c.frames,
                               c.lanes, c.lanes, c.lanes, g_geom.bw, g_geom.rounds, g_geom.round_lanes, getenv("EXP_XCDC") ? 1u : 0u, g_geom.fps);
            return;
        }
    }
    constexpr size_t bytes = (size_t(NB) * TSEG * kFmBlock + 2 * size_t(TSEG) * kFmBlock) * 4;
    static bool once = false;
    if (!once) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_frame_major_sweep<P, LPT, NB, FORM, SLP, TSEG>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
        once = true;
    }
    hipLaunchKernelGGL((stream_frame_major_sweep<P, LPT, NB, FORM, SLP, TSEG>), dim3(g_geom.grid), dim3(kFmBlock), bytes, 0, c.prm, c.st, c.x, y, c.lanes, c.frames, c.lanes, c.lanes,
                       c.lanes, g_geom.bw, g_geom.rounds, g_geom.round_lanes, getenv("EXP_XCDC") ? 1u : 0u, g_geom.fps);
}
template <class P, int NB, int FORM, int SLP = 0, int TSEG = 8>
static void launch_sweep(const Ctx<P> &c, typename P::Out *y)
{
    switch (g_geom.lpt) {
        case 1: return launch_sweep_lpt<P, 1, NB, FORM, SLP, TSEG>(c, y);
        case 2: return launch_sweep_lpt<P, 2, NB, FORM, SLP, TSEG>(c, y);
        case 4: return launch_sweep_lpt<P, 4, NB, FORM, SLP, TSEG>(c, y);
        case 8: return launch_sweep_lpt<P, 8, NB, FORM, SLP, TSEG>(c, y);
        default: return launch_sweep_lpt<P, 16, NB, FORM, SLP, TSEG>(c, y);
    }
}

template <class P, int NB, int NLW, int NSW>
static void launch_roles(const Ctx<P> &c, typename P::Out *y, unsigned grid, int order)
{
    constexpr size_t bytes = roles_lds_bytes<P>(NB);
    static bool once = false;
    if (!once) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_frame_major_roles<P, NB, NLW, NSW>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
        once = true;
    }
    hipLaunchKernelGGL((stream_frame_major_roles<P, NB, NLW, NSW>), dim3(grid), dim3((kRolesCompute + NLW + NSW) * kWave), bytes, 0, c.prm, c.st, c.x, y, c.lanes,
                       c.frames, c.lanes, c.lanes, c.lanes, order);
}

template <class P>
static int run(const char *name, size_t lanes, size_t frames)
{
    using T = typename P::In;
    using T_ = T;
    const size_t n = lanes * frames, pad = size_t(96) << 20;
    char *buf;
    CK(hipMalloc(&buf, 3 * n * 4 + 2 * pad));
    T *x = reinterpret_cast<T *>(buf);
    T *yref = reinterpret_cast<T *>(buf + n * 4);
    char *ytest0 = buf + 2 * n * 4 + pad / 2;
    constexpr int SW = sizeof(P) / 4;
    uint32_t *st, *st_ref;
    unsigned long long *cnt;
    CK(hipMalloc(&st, lanes * SW * 4));
    CK(hipMalloc(&st_ref, lanes * SW * 4));
    CK(hipMalloc(&cnt, 8));
    fill_kernel<<<4096, 256>>>(reinterpret_cast<uint32_t *>(x), n, std::is_same<T, float>::value);
    CK(hipDeviceSynchronize());

    Ctx<P> c{};
    if constexpr (std::is_same<T, float>::value)
        c.prm.sec[0] = {{0.0009446918f, 0.0018893836f, 0.0009446918f, 1.9111970f, -0.9149758f}, 0.f, -1e30f, 1e30f};
    else
        c.prm.sec[0] = {{1014049, 2028098, 1014049, 2052110851, -982425224}, 30, 0, INT32_MIN, INT32_MAX};
    c.st = st, c.x = x, c.lanes = lanes, c.frames = frames;

    // reference output and state from zero state
    CK(hipMemset(st, 0, lanes * SW * 4));
    launch_old<P>(c, yref);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(st_ref, st, lanes * SW * 4, hipMemcpyDeviceToDevice));

    auto check = [&](auto &&launch) -> unsigned long long {
        CK(hipMemset(st, 0, lanes * SW * 4));
        CK(hipMemset(ytest0, 0xA5, n * 4));
        CK(hipMemset(cnt, 0, 8));
        launch(reinterpret_cast<T *>(ytest0));
        CK(hipDeviceSynchronize());
        diff_kernel<<<4096, 256>>>(reinterpret_cast<const uint32_t *>(yref), reinterpret_cast<const uint32_t *>(ytest0), n, cnt);
        diff_kernel<<<256, 256>>>(st_ref, st, lanes * SW, cnt);
        unsigned long long h = 0;
        CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
        return h;
    };
    const size_t deltas[] = {0, 4096, 65536, 262144, 1 << 20, 2 << 20, (6 << 20) + 8192, 25165824, 33554432 + 65536};
    const char *only = getenv("EXP_ONLY");  // comma-free substring filter on the variant name
    auto sweep = [&](const char *variant, unsigned grid, auto &&launch) {
        if (only) {  // '|'-separated substrings
            bool hit = false;
            std::string o(only);
            for (size_t a = 0; a <= o.size();) {
                const size_t b = o.find('|', a) == std::string::npos ? o.size() : o.find('|', a);
                if (b > a && strstr(variant, o.substr(a, b - a).c_str())) hit = true;
                a = b + 1;
            }
            if (!hit) return;
        }
        const unsigned long long bad = check(launch);
        std::vector<double> fr;
        for (size_t d : deltas) {
            const float ms = time_ms([&] { launch(reinterpret_cast<T *>(ytest0 + d)); });
            fr.push_back(double(n) * 8 / (ms * 1e-3) / 8e12);
        }
        // in place: y == x (x is refilled afterwards)
        const float msi = time_ms([&] { launch(const_cast<T *>(x)); });
        const double fi = double(n) * 8 / (msi * 1e-3) / 8e12;
        fill_kernel<<<4096, 256>>>(reinterpret_cast<uint32_t *>(x), n, std::is_same<T, float>::value);
        CK(hipDeviceSynchronize());
        printf("{\"proc\": \"%s\", \"lanes\": %zu, \"frames\": %zu, \"variant\": \"%s\", \"grid\": %u, \"mismatch\": %llu, \"frac\": [", name, lanes, frames, variant, grid, bad);
        double lo = 1, hi = 0, sum = 0;
        for (size_t i = 0; i < fr.size(); i++) {
            printf("%s%.3f", i ? ", " : "", fr[i]);
            lo = std::min(lo, fr[i]), hi = std::max(hi, fr[i]), sum += fr[i];
        }
        printf("], \"inplace\": %.3f, \"worst\": %.3f, \"best\": %.3f, \"mean\": %.3f, \"ms_mean\": %.4f}\n", fi, lo, hi, sum / fr.size(),
               double(n) * 8 / (sum / fr.size() * 8e12) * 1e3);
        fflush(stdout);
    };

    const unsigned g1 = lib_grid(lanes);
    const size_t wgs = (lanes + kFmBlock - 1) / kFmBlock;
    sweep("old lds nb7", g1, [&](T *y) { launch_old<P>(c, y); });
    {
        const int max_lpt = getenv("EXP_MAX_LPT") ? atoi(getenv("EXP_MAX_LPT")) : 16;
        const unsigned max_grid = getenv("EXP_MAX_GRID") ? unsigned(atoi(getenv("EXP_MAX_GRID"))) : 256u;
        if (sweep_geometry(lanes, max_lpt, g_geom, max_grid)) {
            if (getenv("EXP_BW")) g_geom.bw = unsigned(atoi(getenv("EXP_BW")));
            if (getenv("EXP_GRID")) g_geom.grid = unsigned(atoi(getenv("EXP_GRID")));
            if (g_geom.lpt == 1 && g_geom.bw <= 128 && !getenv("EXP_NO_FPS")) g_geom.fps = 256u / g_geom.bw;
            char nm[96];
#define SW(F, N, S, T) snprintf(nm, sizeof nm, "sweep f" #F " nb" #N " slp" #S " ts" #T " lpt%d bw%u fps%u", g_geom.lpt, g_geom.bw, g_geom.fps); sweep(nm, g_geom.grid, [&](T_ *y) { launch_sweep<P, N, F, S, T>(c, y); });
            if (getenv("EXP_SET") && !strcmp(getenv("EXP_SET"), "c5")) {
                SW(0, 7, 0, 8) SW(0, 8, 0, 8) SW(0, 6, 0, 8) SW(3, 7, 4, 8) SW(3, 7, 3, 8) SW(3, 7, 2, 8) SW(3, 8, 4, 8) SW(3, 6, 4, 8) SW(3, 7, 6, 8) SW(2, 7, 2, 8) SW(2, 7, 4, 8) SW(1, 7, 0, 8)
                SW(0, 7, 2, 8) SW(0, 7, 1, 8) SW(1, 7, 2, 8)
            } else
            if (getenv("EXP_SET") && !strcmp(getenv("EXP_SET"), "small")) {
                SW(0, 7, 0, 8) SW(1, 7, 0, 8) SW(3, 7, 0, 8) SW(2, 7, 0, 8)
            } else
            if (getenv("EXP_SET") && !strcmp(getenv("EXP_SET"), "ship")) {
                SW(0, 7, 0, 8) SW(3, 7, 0, 8) SW(3, 7, 4, 8)
            } else
            if (getenv("EXP_SET") && !strcmp(getenv("EXP_SET"), "inplace")) {
                SW(0, 5, 0, 8) SW(0, 6, 0, 8) SW(0, 7, 0, 8) SW(0, 8, 0, 8) SW(0, 9, 0, 8) SW(0, 7, 2, 8) SW(0, 7, 4, 8) SW(0, 6, 2, 8) SW(0, 8, 2, 8)
                SW(3, 6, 0, 8) SW(3, 7, 0, 8) SW(3, 7, 2, 8) SW(3, 7, 4, 8) SW(3, 6, 4, 8) SW(3, 8, 4, 8) SW(2, 7, 2, 8) SW(2, 6, 2, 8) SW(2, 8, 2, 8)
            } else {
                SW(0, 7, 0, 8) SW(1, 7, 0, 8) SW(0, 4, 0, 16) SW(1, 4, 0, 16) SW(1, 3, 0, 16) SW(3, 4, 0, 16) SW(1, 5, 0, 16)
            }
#undef SW
        }
    }
    auto lpt_sweeps = [&](auto lpt_tag) {
        constexpr int L = decltype(lpt_tag)::value;
        if (wgs % L || wgs / L < 128 || wgs / L > 1024) return;
        char nm[64];
        snprintf(nm, sizeof nm, "lds lpt%d adjacent nb7", L);
        sweep(nm, unsigned(wgs / L), [&](T *y) { launch_lpt<P, 7, L, false>(c, y); });
        snprintf(nm, sizeof nm, "lds lpt%d interleaved nb7", L);
        sweep(nm, unsigned(wgs / L), [&](T *y) { launch_lpt<P, 7, L, true>(c, y); });
        snprintf(nm, sizeof nm, "lds lpt%d interleaved nb5", L);
        sweep(nm, unsigned(wgs / L), [&](T *y) { launch_lpt<P, 5, L, true>(c, y); });
        snprintf(nm, sizeof nm, "lds lpt%d interleaved nb4", L);
        sweep(nm, unsigned(wgs / L), [&](T *y) { launch_lpt<P, 4, L, true>(c, y); });
    };
    lpt_sweeps(std::integral_constant<int, 2>{});
    lpt_sweeps(std::integral_constant<int, 4>{});
    lpt_sweeps(std::integral_constant<int, 8>{});
    lpt_sweeps(std::integral_constant<int, 16>{});
    std::vector<unsigned> grids{g1};
    if (wgs >= 512) grids.push_back(unsigned(std::min<size_t>(wgs, 512)));
    for (unsigned g : grids) {
        sweep("roles nb7 L2 S2", g, [&](T *y) { launch_roles<P, 7, 2, 2>(c, y, g, 3); });
        sweep("roles nb7 L1 S1", g, [&](T *y) { launch_roles<P, 7, 1, 1>(c, y, g, 3); });
        sweep("roles nb7 L4 S0 (coupled)", g, [&](T *y) { launch_roles<P, 7, 4, 0>(c, y, g, 3); });
        sweep("roles nb7 L1 S2", g, [&](T *y) { launch_roles<P, 7, 1, 2>(c, y, g, 3); });
        sweep("roles nb7 L2 S4", g, [&](T *y) { launch_roles<P, 7, 2, 4>(c, y, g, 3); });
        sweep("roles nb5 L2 S2", g, [&](T *y) { launch_roles<P, 5, 2, 2>(c, y, g, 3); });
        sweep("roles nb7 L2 S2 order0", g, [&](T *y) { launch_roles<P, 7, 2, 2>(c, y, g, 0); });
    }
    sweep("roles nb9 L2 S2", g1, [&](T *y) { launch_roles<P, 9, 2, 2>(c, y, g1, 3); });
    sweep("roles nb12 L2 S2", g1, [&](T *y) { launch_roles<P, 12, 2, 2>(c, y, g1, 3); });
    CK(hipFree(buf));
    CK(hipFree(st));
    CK(hipFree(st_ref));
    CK(hipFree(cnt));
    return 0;
}

// a section that does nothing (y = x): the kernels' skeleton without arithmetic
struct CopySec {
    using T = int32_t;
    using Sec = idsp::bq::SecI32;
    static constexpr int W = 1;
    static constexpr int COST = 1;
    static constexpr bool kClamp = false;
    static constexpr int LDS_MAX_N = 1;
    static __device__ __forceinline__ int32_t step(const idsp::bq::SecI32 &, uint32_t (&s)[W], int32_t x0)
    {
        s[0] = uint32_t(x0);
        return x0;
    }
};

int main(int argc, char **argv)
{
    const char *proc = argc > 1 ? argv[1] : "i32";
    const size_t lanes = argc > 2 ? atoll(argv[2]) : 65536, frames = argc > 3 ? atoll(argv[3]) : 4096;
    if (argc > 4) g_iters = atoi(argv[4]);
    if (!strcmp(proc, "f32")) return run<bq::Chain<bq::Df2tF32<false>, 1>>("f32_df2t", lanes, frames);
    if (!strcmp(proc, "copy")) return run<bq::Chain<CopySec, 1>>("copy", lanes, frames);
    return run<bq::Chain<bq::Df1I32<false>, 1>>("i32_df1", lanes, frames);
}
