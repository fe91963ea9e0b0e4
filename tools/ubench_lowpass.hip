// Issue rate of the lock-in's lowpass recurrence (dds_dev.h lowpass_step<2>: v_sub_i32 clamp, 2 x v_mad_i64_i32, 4 x v_lshl_add_u64 per
// step) with 1 ... 8 waves per SIMD and nothing else going on: no LDS, no barrier, no memory.  One workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fwrapv -Iinclude -Iidsp_amd/csrc tools/ubench_lowpass.hip -o build/ubench_lowpass
#include "dds_dev.h"

#include <cstdio>

using namespace idsp;

template <int CHAINS>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int steps, int k0, int k1)
{
    int64_t s[CHAINS][2];
    int32_t x[CHAINS];
    const int32_t kk[2] = {k0, k1};
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s[c][0] = threadIdx.x * 77 + c, s[c][1] = blockIdx.x * 131 + c, x[c] = int32_t(threadIdx.x * 2654435761u) >> (3 + c);
    for (int i = 0; i < steps; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) x[c] = lowpass_step<2>(kk, s[c], x[c] + 12345);
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc += uint32_t(x[c]) + uint32_t(s[c][0]);
    if (acc == 0x12345u) out[threadIdx.x] = acc;
}

template <int CHAINS>
void run(uint32_t *out, int cus, double ghz)
{
    for (int wps : {1, 2, 3, 4, 8}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        const int steps = 2048;
        hipLaunchKernelGGL(k<CHAINS>, dim3(cus), dim3(256 * wps > 1024 ? 1024 : 256 * wps), 0, 0, out, steps, 10000, -9500000);
        hipDeviceSynchronize();
        const int blocks = wps == 8 ? 2 * cus : cus;
        hipEventRecord(e0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256 * wps > 1024 ? 1024 : 256 * wps), 0, 0, out, steps, 10000, -9500000);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms / 5 * 1e-3 * ghz * 1e9;
        const double nstep = double(steps) * 16 * CHAINS;  // lowpass steps per wave
        // 8 VALU instructions per step (7 + the add that perturbs x)
        std::printf("  chains/wave %d, %d waves/SIMD: %6.1f cycles per step and wave, %5.2f cycles per VALU instruction and SIMD\n", CHAINS, wps, cyc / nstep, cyc / (nstep * wps * 8));
    }
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    std::printf("%s: %d CUs, %.2f GHz nominal\n", p.gcnArchName, p.multiProcessorCount, ghz);
    uint32_t *out;
    hipMalloc(&out, 4096 * 4);
    run<1>(out, p.multiProcessorCount, ghz);
    run<2>(out, p.multiProcessorCount, ghz);
    return 0;
}
