// exp_c2.hip — kernel-shape experiments for the FRAME_MAJOR i32 DF1 biquad at
// 65536 lanes x 4096 frames (development tool, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/exp_c2.hip -o /tmp/exp_c2 && /tmp/exp_c2
// Variants: U = register-window depth, LPT = adjacent lanes per thread (load
// width 4*LPT bytes), NT = nontemporal loads/stores, BS = block size.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("%s: %s\n", #x, hipGetErrorString(e));                          \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

struct Sec {
    int32_t ba[5];
    int32_t frac;
};

__device__ __forceinline__ int32_t step1(const Sec &c, int32_t (&s)[4], int32_t x0)
{
    int64_t acc = int64_t(c.ba[0]) * x0;
    acc += int64_t(c.ba[1]) * s[0];
    acc += int64_t(c.ba[2]) * s[1];
    acc += int64_t(c.ba[3]) * s[2];
    acc += int64_t(c.ba[4]) * s[3];
    int32_t y0 = int32_t(__builtin_amdgcn_alignbit(uint32_t(uint64_t(acc) >> 32), uint32_t(acc), uint32_t(c.frac)));
    s[1] = s[0];
    s[0] = x0;
    s[3] = s[2];
    s[2] = y0;
    return y0;
}

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
template <int LPT>
struct Vec;
template <>
struct Vec<1> {
    using T = int32_t;
};
template <>
struct Vec<2> {
    using T = v2i;
};
template <>
struct Vec<4> {
    using T = v4i;
};

template <int LPT, bool NT>
__device__ __forceinline__ typename Vec<LPT>::T ld(const typename Vec<LPT>::T *p)
{
    if constexpr (NT)
        return __builtin_nontemporal_load(p);
    else
        return *p;
}
template <int LPT, bool NT>
__device__ __forceinline__ void st(typename Vec<LPT>::T *p, typename Vec<LPT>::T v)
{
    if constexpr (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

template <int U, int LPT, bool NT, int BS>
__global__ __launch_bounds__(BS) void k(const Sec c, const int32_t *x, int32_t *y, size_t lanes, size_t frames)
{
    using V = typename Vec<LPT>::T;
    const size_t t = size_t(blockIdx.x) * BS + threadIdx.x;
    const size_t lane = t * LPT;
    if (lane >= lanes) return;
    int32_t s[LPT][4] = {};
    const size_t stride = lanes / LPT;  // in V units
    const V *xp = reinterpret_cast<const V *>(x) + t;
    V *yp = reinterpret_cast<V *>(y) + t;
    V ring[U];
#pragma unroll
    for (int u = 0; u < U; u++) ring[u] = ld<LPT, NT>(xp + size_t(u) * stride);
    size_t f = 0;
    for (; f + 2 * U <= frames; f += U) {
        const V *xn = xp + (f + U) * stride;
        V *yn = yp + f * stride;
#pragma unroll
        for (int u = 0; u < U; u++) {
            V v = ring[u];
            ring[u] = ld<LPT, NT>(xn + size_t(u) * stride);
            V o;
            if constexpr (LPT == 1) {
                o = step1(c, s[0], v);
            } else if constexpr (LPT == 2) {
                o.x = step1(c, s[0], v.x);
                o.y = step1(c, s[1], v.y);
            } else {
                o.x = step1(c, s[0], v.x);
                o.y = step1(c, s[1], v.y);
                o.z = step1(c, s[2], v.z);
                o.w = step1(c, s[3], v.w);
            }
            st<LPT, NT>(yn + size_t(u) * stride, o);
        }
    }
    // (tail omitted: frames is a multiple of U in this experiment; drain the last window)
    V *yn = yp + f * stride;
#pragma unroll
    for (int u = 0; u < U; u++) {
        V v = ring[u];
        V o;
        if constexpr (LPT == 1) {
            o = step1(c, s[0], v);
        } else if constexpr (LPT == 2) {
            o.x = step1(c, s[0], v.x);
            o.y = step1(c, s[1], v.y);
        } else {
            o.x = step1(c, s[0], v.x);
            o.y = step1(c, s[1], v.y);
            o.z = step1(c, s[2], v.z);
            o.w = step1(c, s[3], v.w);
        }
        if (f + u < frames) st<LPT, NT>(yn + size_t(u) * stride, o);
    }
}

// plain float4 copy for the achievable-bandwidth reference
__global__ __launch_bounds__(256) void copy4(const int4 *a, int4 *b, size_t n)
{
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) b[i] = a[i];
}

template <int U, int LPT, bool NT, int BS>
void run(const char *name, const Sec &c, const int32_t *x, int32_t *y, size_t lanes, size_t frames)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const unsigned grid = unsigned((lanes / LPT + BS - 1) / BS);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<U, LPT, NT, BS>), dim3(grid), dim3(BS), 0, 0, c, x, y, lanes, frames);
    CK(hipDeviceSynchronize());
    float best = 1e9, tot = 0;
    const int it = 10;
    for (int i = 0; i < it; i++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<U, LPT, NT, BS>), dim3(grid), dim3(BS), 0, 0, c, x, y, lanes, frames);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
        tot += ms;
    }
    const double bytes = 8.0 * lanes * frames;
    printf("%-28s avg %.4f ms  min %.4f ms  %.0f GB/s (min %.0f)\n", name, tot / it, best, bytes / (tot / it) / 1e6, bytes / best / 1e6);
}

int main()
{
    const size_t lanes = 65536, frames = 4096 - (4096 % 48);  // 4080: multiple of 16, 24, 48... keep all variants tail-free
    const size_t n = lanes * 4096;
    int32_t *x, *y;
    CK(hipMalloc(&x, n * 4));
    CK(hipMalloc(&y, n * 4));
    std::vector<int32_t> h(1 << 20);
    for (auto &v : h) v = (rand() % (1 << 25)) - (1 << 24);
    for (size_t o = 0; o < n; o += h.size()) CK(hipMemcpy(x + o, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    Sec c{{1055, 2110, 1055, 2052218165, -982680842}, 30};
    {
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        for (int g : {2048, 8192, 65536}) {
            float best = 1e9;
            for (int i = 0; i < 8; i++) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(copy4, dim3(g), dim3(256), 0, 0, (const int4 *)x, (int4 *)y, n / 4);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                best = ms < best ? ms : best;
            }
            printf("copy4 grid %-6d             min %.4f ms  %.0f GB/s\n", g, best, 8.0 * n / best / 1e6);
        }
    }
    const size_t fr48 = frames;
    printf("-- 65536 lanes\n");
    run<12, 1, false, 256>("U12 LPT1 BS256", c, x, y, lanes, fr48);
    run<16, 1, false, 256>("U16 LPT1 BS256", c, x, y, lanes, fr48);
    run<16, 1, true, 256>("U16 LPT1 BS256 NT", c, x, y, lanes, fr48);
    run<16, 1, false, 128>("U16 LPT1 BS128", c, x, y, lanes, fr48);
    run<24, 2, false, 128>("U24 LPT2 BS128", c, x, y, lanes, fr48);
    run<16, 4, false, 64>("U16 LPT4 BS64", c, x, y, lanes, fr48);
    run<24, 4, false, 64>("U24 LPT4 BS64", c, x, y, lanes, fr48);
    run<24, 4, true, 64>("U24 LPT4 BS64 NT", c, x, y, lanes, fr48);
    run<48, 4, false, 64>("U48 LPT4 BS64", c, x, y, lanes, fr48);
    for (size_t L : {size_t(131072), size_t(262144), size_t(1048576)}) {
        const size_t F = (n / L) - ((n / L) % 48);
        printf("-- %zu lanes x %zu frames\n", L, F);
        run<8, 1, false, 256>("U8  LPT1 BS256", c, x, y, L, F);
        run<16, 1, false, 256>("U16 LPT1 BS256", c, x, y, L, F);
        run<8, 2, false, 128>("U8  LPT2 BS128", c, x, y, L, F);
        run<16, 2, false, 128>("U16 LPT2 BS128", c, x, y, L, F);
        run<24, 2, false, 128>("U24 LPT2 BS128", c, x, y, L, F);
        run<8, 4, false, 64>("U8  LPT4 BS64", c, x, y, L, F);
        run<12, 4, false, 64>("U12 LPT4 BS64", c, x, y, L, F);
        run<16, 4, false, 64>("U16 LPT4 BS64", c, x, y, L, F);
        run<24, 4, false, 64>("U24 LPT4 BS64", c, x, y, L, F);
        run<8, 4, false, 256>("U8  LPT4 BS256", c, x, y, L, F);
        run<16, 4, false, 256>("U16 LPT4 BS256", c, x, y, L, F);
    }
    return 0;
}
