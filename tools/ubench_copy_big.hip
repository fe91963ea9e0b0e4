// ubench_copy_big.hip — round 5: what a plain 16-byte-per-lane nontemporal copy reaches at the C2 (1 GiB -> 1 GiB) and the C5
// (16 GiB -> 16 GiB) footprints: the ceiling the FrameMajor kernels are read against at each size.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_copy_big.hip -o build/ubench_copy_big && build/ubench_copy_big
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// grid-stride: consecutive workgroups copy consecutive 4 KiB pieces, the whole grid sweeps the buffer front to back
template <int U>
__global__ __launch_bounds__(256) void k_copy(const u32x4 *a, u32x4 *b, size_t n)
{
    const size_t stride = size_t(gridDim.x) * 256;
    size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(a + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) __builtin_nontemporal_store(v[u], b + i + u * stride);
    }
    for (; i < n; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}
// chunked: workgroup w owns one contiguous chunk of n / grid elements
template <int U>
__global__ __launch_bounds__(256) void k_copy_chunk(const u32x4 *a, u32x4 *b, size_t n)
{
    const size_t per = n / gridDim.x, lo = per * blockIdx.x, hi = lo + per;
    for (size_t i = lo + threadIdx.x; i + (U - 1) * 256 < hi; i += U * 256) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = __builtin_nontemporal_load(a + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; u++) __builtin_nontemporal_store(v[u], b + i + u * 256);
    }
}

template <class F>
static float med_ms(F &&f, int it)
{
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    std::vector<float> ts;
    for (int i = 0; i < it + 2; i++) {
        hipEventRecord(a);
        f();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (i >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main()
{
    for (size_t gib : {size_t(1), size_t(4), size_t(16)}) {
        const size_t bytes = gib << 30, n = bytes / 16;
        char *buf;
        CK(hipMalloc(&buf, 2 * bytes + (size_t(64) << 20)));
        CK(hipMemset(buf, 1, 2 * bytes));
        const u32x4 *a = reinterpret_cast<const u32x4 *>(buf);
        for (size_t off : {size_t(0), size_t(4096), size_t(2) << 20, (size_t(6) << 20) + 8192}) {
            u32x4 *b = reinterpret_cast<u32x4 *>(buf + bytes + off);
            for (unsigned grid : {1024u, 2048u, 8192u}) {
                const float m1 = med_ms([&] { k_copy<4><<<grid, 256>>>(a, b, n); }, 7);
                const float m2 = med_ms([&] { k_copy<8><<<grid, 256>>>(a, b, n); }, 7);
                const float m3 = med_ms([&] { k_copy_chunk<8><<<grid, 256>>>(a, b, n); }, 7);
                printf("{\"GiB\": %zu, \"y_off\": %zu, \"grid\": %u, \"gridstride_u4_TBs\": %.3f, \"gridstride_u8_TBs\": %.3f, \"chunk_u8_TBs\": %.3f}\n", gib, off, grid,
                       2.0 * bytes / (m1 * 1e-3) / 1e12, 2.0 * bytes / (m2 * 1e-3) / 1e12, 2.0 * bytes / (m3 * 1e-3) / 1e12);
                fflush(stdout);
            }
        }
        CK(hipFree(buf));
    }
    return 0;
}
