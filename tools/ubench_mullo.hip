// v_mul_lo_u32 against v_mad_u64_u32 on gfx950: the Wdf adaptor (normal_wdf.hip) is five 32-bit multiplies by small run-time integers, a
// v_mul_hi_i32 and three adds; is a 32-bit multiply-add cheaper as the low word of a 64-bit mad?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fwrapv tools/ubench_mullo.hip -o build/ubench_mullo
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t mad_lo(uint32_t a, uint32_t b, uint32_t c)  // a * b + c through v_mad_u64_u32 (low word)
{
    uint64_t d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(uint64_t(c)));
    return uint32_t(d);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, int n, uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3)
{
    uint32_t x = threadIdx.x * 2654435761u, z = blockIdx.x * 40503u + 7, acc = 0;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            uint32_t o0, o1;
            if constexpr (MODE == 0) {
                o0 = acc + x * k0 + z * k1;
                o1 = acc + x * k2 + z * k3;
            } else {
                o0 = mad_lo(x, k0, mad_lo(z, k1, acc));
                o1 = mad_lo(x, k2, mad_lo(z, k3, acc));
            }
            z = o0, x = o1, acc += 1;
        }
    }
    if (x + z == 0x1234567u) out[threadIdx.x] = x;
}

template <int MODE>
void run(const char *name, uint32_t *out, int cus, double ghz)
{
    for (int wps : {1, 2, 4}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        const int n = 4096;
        hipLaunchKernelGGL(k<MODE>, dim3(cus * wps), dim3(256), 0, 0, out, n, 1u, 2u, 0xffffffffu, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<MODE>, dim3(cus * wps), dim3(256), 0, 0, out, n, 1u, 2u, 0xffffffffu, 1u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms / 5 * 1e-3 * ghz * 1e9 / (double(n) * 8 * wps);
        std::printf("  %-34s %d waves/SIMD: %6.1f cycles per [o0, o1] pair and SIMD (4 multiplies + 4 adds)\n", name, wps, cyc);
    }
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    uint32_t *out;
    hipMalloc(&out, 4096);
    run<0>("v_mul_lo_u32 + adds", out, p.multiProcessorCount, p.clockRate * 1e-6);
    run<1>("v_mad_u64_u32 (low word)", out, p.multiProcessorCount, p.clockRate * 1e-6);
    return 0;
}
