// exp_lds.hip — experiment: FRAME_MAJOR i32 DF1 biquad with LDS-DMA staged tiles
// (global_load_lds_dwordx4 -> LDS -> one lane per thread -> LDS -> dwordx4 stores)
// versus the register-window kernel.  Development tool, not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp_lds.hip -o build/exp_lds
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                 \
    do {                                                      \
        hipError_t e = (x);                                   \
        if (e != hipSuccess) {                                \
            printf("%s: %s\n", #x, hipGetErrorString(e));     \
            exit(1);                                          \
        }                                                     \
    } while (0)

struct Sec {
    int32_t ba[5];
    int32_t frac;
};
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int32_t step1(const Sec &c, int32_t (&s)[4], int32_t x0)
{
    int64_t acc = int64_t(c.ba[0]) * x0;
    acc += int64_t(c.ba[1]) * s[0];
    acc += int64_t(c.ba[2]) * s[1];
    acc += int64_t(c.ba[3]) * s[2];
    acc += int64_t(c.ba[4]) * s[3];
    int32_t y0 = int32_t(__builtin_amdgcn_alignbit(uint32_t(uint64_t(acc) >> 32), uint32_t(acc), uint32_t(c.frac)));
    s[1] = s[0], s[0] = x0, s[3] = s[2], s[2] = y0;
    return y0;
}

// reference: plain register-window kernel (U=16, one lane per thread)
template <int U>
__global__ __launch_bounds__(256) void k_ref(const Sec c, const int32_t *x, int32_t *y, size_t lanes, size_t frames)
{
    const size_t lane = size_t(blockIdx.x) * 256 + threadIdx.x;
    int32_t s[4] = {};
    const int32_t *xp = x + lane;
    int32_t *yp = y + lane;
    int32_t ring[U];
#pragma unroll
    for (int u = 0; u < U; u++) ring[u] = xp[size_t(u) * lanes];
    size_t f = 0;
    for (; f + 2 * U <= frames; f += U) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            int32_t v = ring[u];
            ring[u] = xp[(f + U + u) * lanes];
            yp[(f + u) * lanes] = step1(c, s, v);
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
        if (f + u < frames) yp[(f + u) * lanes] = step1(c, s, ring[u]);
}

// 16-byte-per-lane LDS-DMA load: LDS destination = wave-uniform byte address + lane*16
template <bool NT>
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_dst)
                     : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// T frames per tile, NB input tiles in the ring; 256 lanes per block
template <int T, int NB, bool LNT = false, bool SNT = false>
__global__ __launch_bounds__(256) void k_lds(const Sec c, const int32_t *x, int32_t *y, size_t lanes, size_t frames)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *tin = smem;                  // [NB][T][256]
    uint32_t *tout = smem + NB * T * 256;  // [2][T][256]
    const int tid = threadIdx.x, lid = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t lane0 = size_t(blockIdx.x) * 256;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)smem;
    constexpr int RPW = T / 4;  // rows per wave and tile
    const size_t ntiles = frames / T;

    auto issue = [&](size_t tile) {
        const int slot = int(tile % NB);
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int r = wave + 4 * j;
            const int32_t *g = x + (tile * T + r) * lanes + lane0 + lid * 4;
            glds16<LNT>(g, lds_base + uint32_t(((slot * T + r) * 256) * 4));
        }
    };
    auto store = [&](size_t tile) {
        const uint32_t *o = tout + (tile & 1) * T * 256;
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int r = wave + 4 * j;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(o + r * 256 + lid * 4);
            if constexpr (SNT)
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(y + (tile * T + r) * lanes + lane0 + lid * 4));
            else
                *reinterpret_cast<u32x4 *>(y + (tile * T + r) * lanes + lane0 + lid * 4) = v;
        }
    };
    int32_t s[4] = {};
    auto compute = [&](size_t tile) {
        const uint32_t *in = tin + (tile % NB) * T * 256;
        uint32_t *o = tout + (tile & 1) * T * 256;
#pragma unroll
        for (int r = 0; r < T; r++) o[r * 256 + tid] = uint32_t(step1(c, s, int32_t(in[r * 256 + tid])));
    };

    for (size_t t = 0; t < NB && t < ntiles; t++) issue(t);
    size_t i = 0;
    // start-up and drain iterations wait for everything; steady state keeps NB-1 tiles of loads
    // (+ the interleaved stores) in flight: ops younger than tile i's loads = RPW + (NB-1)*2*RPW
    for (; i < ntiles; i++) {
        const bool steady = i >= NB && i + NB < ntiles;
        if (steady)
            wait_vm<RPW + (NB - 1) * 2 * RPW>();
        else
            wait_vm<0>();
        bar();
        compute(i);
        bar();
        if (i + NB < ntiles) issue(i + NB);
        store(i);
    }
}

template <class F>
float timeit(F launch, int it = 10)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int i = 0; i < it; i++) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    return best;
}

template <int T, int NB, bool LNT = false, bool SNT = false>
void run_lds(const char *name, const Sec &c, const int32_t *x, int32_t *y, const int32_t *yref, size_t lanes, size_t frames)
{
    const size_t bytes = size_t(NB + 2) * T * 256 * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds<T, NB, LNT, SNT>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    CK(hipMemset(y, 0xff, lanes * frames * 4));
    auto launch = [&] { hipLaunchKernelGGL((k_lds<T, NB, LNT, SNT>), dim3(unsigned(lanes / 256)), dim3(256), bytes, 0, c, x, y, lanes, frames); };
    launch();
    CK(hipDeviceSynchronize());
    // correctness against the reference kernel's output
    std::vector<int32_t> a(1 << 20), b(1 << 20);
    size_t bad = 0;
    for (size_t off : {size_t(0), lanes * frames / 2, lanes * frames - (size_t(1) << 20)}) {
        CK(hipMemcpy(a.data(), y + off, a.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), yref + off, b.size() * 4, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < a.size(); k++) bad += a[k] != b[k];
    }
    const float ms = timeit(launch);
    printf("%-22s lanes %-8zu min %.4f ms  %.0f GB/s  mismatches %zu  lds %zu KiB\n", name, lanes, ms, 8.0 * lanes * frames / ms / 1e6, bad,
           bytes >> 10);
}

int main()
{
    const size_t n = size_t(65536) * 4096;
    int32_t *x, *y, *yref;
    CK(hipMalloc(&x, n * 4));
    CK(hipMalloc(&y, n * 4));
    CK(hipMalloc(&yref, n * 4));
    std::vector<int32_t> h(1 << 20);
    for (auto &v : h) v = (rand() % (1 << 25)) - (1 << 24);
    for (size_t o = 0; o < n; o += h.size()) CK(hipMemcpy(x + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    Sec c{{1055, 2110, 1055, 2052218165, -982680842}, 30};
    for (size_t lanes : {size_t(65536), size_t(131072)}) {
        const size_t frames = n / lanes;
        auto ref = [&] { hipLaunchKernelGGL((k_ref<16>), dim3(unsigned(lanes / 256)), dim3(256), 0, 0, c, x, yref, lanes, frames); };
        const float ms = timeit(ref);
        printf("%-22s lanes %-8zu min %.4f ms  %.0f GB/s\n", "ref U16", lanes, ms, 8.0 * n / ms / 1e6);
        for (int rep = 0; rep < 2; rep++) {
            run_lds<8, 8>("lds T8 NB8", c, x, y, yref, lanes, frames);
            run_lds<8, 8, true, false>("lds T8 NB8 ldNT", c, x, y, yref, lanes, frames);
            run_lds<8, 8, false, true>("lds T8 NB8 stNT", c, x, y, yref, lanes, frames);
            run_lds<8, 8, true, true>("lds T8 NB8 ld+stNT", c, x, y, yref, lanes, frames);
            run_lds<16, 4, true, true>("lds T16 NB4 ld+stNT", c, x, y, yref, lanes, frames);
            run_lds<8, 12, true, true>("lds T8 NB12 ld+stNT", c, x, y, yref, lanes, frames);
            run_lds<4, 16, true, true>("lds T4 NB16 ld+stNT", c, x, y, yref, lanes, frames);
        }
    }
    return 0;
}
