// HBM ceilings on the box: read-only, write-only and copy streams with 16-byte-per-lane accesses,
// default and nontemporal, so that roofline fractions can be read against what the memory system
// actually delivers for each traffic mix.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm.hip -o build/ubench_hbm && build/ubench_hbm
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHK(x)                                                  \
    do {                                                        \
        hipError_t e_ = (x);                                    \
        if (e_ != hipSuccess) {                                 \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                           \
        }                                                       \
    } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void k_read(const u32x4 *a, u32x4 *sink, size_t n)
{
    u32x4 acc = {0, 0, 0, 0};
    const size_t stride = size_t(gridDim.x) * 256;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) acc ^= NT ? __builtin_nontemporal_load(a + i) : a[i];
    if (acc.x == 0x12345u) sink[0] = acc;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_write(u32x4 *a, size_t n)
{
    const u32x4 v = {1, 2, 3, uint32_t(threadIdx.x)};
    const size_t stride = size_t(gridDim.x) * 256;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
        if (NT)
            __builtin_nontemporal_store(v, a + i);
        else
            a[i] = v;
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const u32x4 *a, u32x4 *b, size_t n)
{
    const size_t stride = size_t(gridDim.x) * 256;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += stride) {
        const u32x4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT)
            __builtin_nontemporal_store(v, b + i);
        else
            b[i] = v;
    }
}
// 16 reads : 1 write (the /16 decimator mix) and 1 : 16 (the x16 interpolator mix)
template <bool NT>
__global__ __launch_bounds__(256) void k_r16w1(const u32x4 *a, u32x4 *b, size_t n)
{
    const size_t stride = size_t(gridDim.x) * 256;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n / 16; i += stride) {
        u32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 16; k++) acc ^= NT ? __builtin_nontemporal_load(a + i + size_t(k) * (n / 16)) : a[i + size_t(k) * (n / 16)];
        b[i] = acc;
    }
}

template <class F>
int timeit(const char *name, double bytes, F launch)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    launch();
    CHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 7; r++) {
        CHK(hipEventRecord(e0));
        launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::printf("%-28s %8.3f ms  %7.1f GB/s\n", name, best, bytes / (best * 1e-3) / 1e9);
    return 0;
}

int main()
{
    const size_t bytes = size_t(2) << 30;  // 2 GiB per buffer
    const size_t n = bytes / 16;
    u32x4 *a, *b;
    CHK(hipMalloc(&a, bytes));
    CHK(hipMalloc(&b, bytes));
    CHK(hipMemset(a, 1, bytes));
    CHK(hipMemset(b, 2, bytes));
    for (int blocks : {1024, 2048, 4096, 16384}) {
        std::printf("grid %d x 256\n", blocks);
        timeit("read", double(bytes), [&] { hipLaunchKernelGGL(k_read<false>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
        timeit("read nt", double(bytes), [&] { hipLaunchKernelGGL(k_read<true>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
        timeit("write", double(bytes), [&] { hipLaunchKernelGGL(k_write<false>, dim3(blocks), dim3(256), 0, 0, a, n); });
        timeit("write nt", double(bytes), [&] { hipLaunchKernelGGL(k_write<true>, dim3(blocks), dim3(256), 0, 0, a, n); });
        timeit("copy (r+w bytes)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy<false>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
        timeit("copy nt (r+w bytes)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy<true>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
        timeit("16 reads : 1 write", bytes * 17.0 / 16.0, [&] { hipLaunchKernelGGL(k_r16w1<false>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
        timeit("16 reads : 1 write nt", bytes * 17.0 / 16.0, [&] { hipLaunchKernelGGL(k_r16w1<true>, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    }
    return 0;
}
