// Access-shape microbenchmark: 16-byte pieces at a 64-byte lane stride (each thread streams its own
// 64-byte chunk, the Cic / FRAME_MAJOR `[T; 16]` shape) vs wave-contiguous 1 KiB per instruction,
// at one and four waves per CU, with U chunk rows in flight per thread.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// rows of `lanes` 64-byte chunks; thread = lane; STRIDED: piece k of own chunk; else piece (k*64+lane_in_wave) of the wave's 4 KiB
template <bool STRIDED, int U>
__global__ __launch_bounds__(64) void k_read(const u32x4 *a, u32x4 *sink, size_t lanes, size_t rows)
{
    const size_t wave0 = size_t(blockIdx.x) * 64;  // first lane of the wave
    const int lid = threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t r = 0; r + U <= rows; r += U) {
        u32x4 v[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32x4 *row = a + ((r + u) * lanes + wave0) * 4;  // wave's 4 KiB = 256 vectors
#pragma unroll
            for (int k = 0; k < 4; k++) v[u][k] = STRIDED ? row[lid * 4 + k] : row[k * 64 + lid];
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int k = 0; k < 4; k++) acc ^= v[u][k];
    }
    if (acc.x == 0x1234567u) sink[0] = acc;
}
template <bool STRIDED>
__global__ __launch_bounds__(64) void k_write(u32x4 *a, size_t lanes, size_t rows)
{
    const size_t wave0 = size_t(blockIdx.x) * 64;
    const int lid = threadIdx.x;
    const u32x4 v = {1, 2, 3, uint32_t(lid)};
    for (size_t r = 0; r < rows; r++) {
        u32x4 *row = a + (r * lanes + wave0) * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (STRIDED) row[lid * 4 + k] = v; else row[k * 64 + lid] = v;
        }
    }
}

// same stores, but every EVERY rows the wave also waits for a 4-byte load (vmcnt is in-order on gfx9:
// the wait drains every older store too) — the shape of an interpolator that reads one input per chunk
template <bool STRIDED, int EVERY>
__global__ __launch_bounds__(64) void k_write_drain(u32x4 *a, const uint32_t *in, size_t lanes, size_t rows)
{
    const size_t wave0 = size_t(blockIdx.x) * 64;
    const int lid = threadIdx.x;
    u32x4 v = {1, 2, 3, uint32_t(lid)};
    for (size_t r = 0; r < rows; r++) {
        if (r % EVERY == 0) {
            const uint32_t t = __builtin_nontemporal_load(in + (r * 64 + lid) % 4096);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            v.x += t;
        }
        u32x4 *row = a + (r * lanes + wave0) * 4;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (STRIDED) row[lid * 4 + k] = v; else row[k * 64 + lid] = v;
        }
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
    for (int r = 0; r < 5; r++) {
        CHK(hipEventRecord(e0));
        launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::printf("  %-34s %8.3f ms  %7.1f GB/s\n", name, best, bytes / (best * 1e-3) / 1e9);
    return 0;
}

int main()
{
    const size_t bytes = size_t(4) << 30;
    u32x4 *a, *b;
    CHK(hipMalloc(&a, bytes));
    CHK(hipMalloc(&b, 65536));
    CHK(hipMemset(a, 1, bytes));
    for (size_t lanes : {size_t(16384), size_t(65536)}) {
        const size_t rows = bytes / (lanes * 64);
        const unsigned waves = unsigned(lanes / 64);
        std::printf("lanes %zu (%u waves), rows %zu\n", lanes, waves, rows);
        timeit("read strided U=1", double(bytes), [&] { hipLaunchKernelGGL((k_read<true, 1>), dim3(waves), dim3(64), 0, 0, a, b, lanes, rows); });
        timeit("read strided U=4", double(bytes), [&] { hipLaunchKernelGGL((k_read<true, 4>), dim3(waves), dim3(64), 0, 0, a, b, lanes, rows); });
        timeit("read strided U=16", double(bytes), [&] { hipLaunchKernelGGL((k_read<true, 16>), dim3(waves), dim3(64), 0, 0, a, b, lanes, rows); });
        timeit("read contiguous U=1", double(bytes), [&] { hipLaunchKernelGGL((k_read<false, 1>), dim3(waves), dim3(64), 0, 0, a, b, lanes, rows); });
        timeit("read contiguous U=4", double(bytes), [&] { hipLaunchKernelGGL((k_read<false, 4>), dim3(waves), dim3(64), 0, 0, a, b, lanes, rows); });
        timeit("read contiguous U=16", double(bytes), [&] { hipLaunchKernelGGL((k_read<false, 16>), dim3(waves), dim3(64), 0, 0, a, b, lanes, rows); });
        timeit("write strided", double(bytes), [&] { hipLaunchKernelGGL((k_write<true>), dim3(waves), dim3(64), 0, 0, a, lanes, rows); });
        timeit("write contiguous", double(bytes), [&] { hipLaunchKernelGGL((k_write<false>), dim3(waves), dim3(64), 0, 0, a, lanes, rows); });
        timeit("write strided, drain every 8", double(bytes), [&] { hipLaunchKernelGGL((k_write_drain<true, 8>), dim3(waves), dim3(64), 0, 0, a, (const uint32_t *)b, lanes, rows); });
        timeit("write strided, drain every 32", double(bytes), [&] { hipLaunchKernelGGL((k_write_drain<true, 32>), dim3(waves), dim3(64), 0, 0, a, (const uint32_t *)b, lanes, rows); });
        timeit("write contiguous, drain every 8", double(bytes), [&] { hipLaunchKernelGGL((k_write_drain<false, 8>), dim3(waves), dim3(64), 0, 0, a, (const uint32_t *)b, lanes, rows); });
        timeit("write contiguous, drain every 32", double(bytes), [&] { hipLaunchKernelGGL((k_write_drain<false, 32>), dim3(waves), dim3(64), 0, 0, a, (const uint32_t *)b, lanes, rows); });
    }
    return 0;
}
