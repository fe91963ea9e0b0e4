// The i32 DF1 biquad step (src/iir/biquad.rs:366-383) as ONE wave runs it, in shader cycles per step: five
// `v_mad_i64_i32` on one dependent chain + `v_alignbit_b32`, with 16 / 32 / 64 active lanes, 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_df1_step.hip -o build/ubench_df1_step && build/ubench_df1_step
// Why: round 5 explained the 0.145 ms floor of 8192 .. 16384 lanes x 4096 frames as "a
// v_mad_i64_i32 occupies its SIMD for 16 cycles, 5 per step = 80 cycles", while r01_ubench_valu.txt has 9.85 cycles per
// instruction at one wave per SIMD.  This measures the step itself, in the three forms the library has:
//   chain   Df1I32::step — the reference's left-to-right sum: 5 dependent MADs
//   tile    Df1I32::tile — feed-forward products of 8 samples first (independent), a1 y1 / a2 y2 and the shift on the chain
//   f32     DF2T f32 step (biquad.rs:418-428) for comparison: 5 mul + 4 add, chain = 1 mul + 1 add
// Samples come from registers (16 per block, rotated), so no memory is in the loop.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHK(x)                                                      \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));     \
            return 1;                                               \
        }                                                           \
    } while (0)

constexpr int kBlocks = 512;  // blocks of 8 steps
constexpr int kR = 8;

__device__ __forceinline__ int64_t mulw(int32_t c, int32_t v) { return int64_t(c) * int64_t(v); }
__device__ __forceinline__ int64_t wadd(int64_t a, int64_t b) { return int64_t(uint64_t(a) + uint64_t(b)); }
__device__ __forceinline__ int32_t shr_lo(int64_t acc, int f)
{
    return int32_t(__builtin_amdgcn_alignbit(uint32_t(uint64_t(acc) >> 32), uint32_t(uint64_t(acc)), uint32_t(f)));
}

struct Sec {
    int32_t ba[5];
    int32_t frac;
};

template <int FORM>
__global__ void k_df1(int32_t *out, long long *cyc, Sec c, int active)
{
    const int lid = threadIdx.x % 64;
    int32_t x[kR];
    for (int r = 0; r < kR; r++) x[r] = int32_t((lid * 2654435761u + r * 40503u) >> 8) - (1 << 23);
    int32_t x1 = 0, x2 = 0, y1 = 0, y2 = 0;
    float fs0 = 0.f, fs1 = 0.f;
    const float fb0 = 0.01f, fb1 = 0.02f, fb2 = 0.01f, fa1 = 1.9f, fa2 = -0.91f;
    long long t0 = 0, t1 = 0;
    if (lid < active) {
        t0 = clock64();
        for (int b = 0; b < kBlocks; b++) {
            if constexpr (FORM == 0) {
#pragma unroll
                for (int r = 0; r < kR; r++) {
                    int64_t acc = mulw(c.ba[0], x[r]);
                    acc = wadd(acc, mulw(c.ba[1], x1));
                    acc = wadd(acc, mulw(c.ba[2], x2));
                    acc = wadd(acc, mulw(c.ba[3], y1));
                    acc = wadd(acc, mulw(c.ba[4], y2));
                    const int32_t y0 = shr_lo(acc, c.frac);
                    x2 = x1, x1 = x[r], y2 = y1, y1 = y0;
                    x[r] ^= y0 & 0xff;  // keeps the block from being hoisted; one VALU
                }
            } else if constexpr (FORM == 1) {
                int64_t acc[kR];
#pragma unroll
                for (int r = 0; r < kR; r++) acc[r] = mulw(c.ba[0], x[r]);
#pragma unroll
                for (int r = 0; r < kR; r++) acc[r] = wadd(acc[r], mulw(c.ba[1], r >= 1 ? x[r - 1] : x1));
#pragma unroll
                for (int r = 0; r < kR; r++) acc[r] = wadd(acc[r], mulw(c.ba[2], r >= 2 ? x[r - 2] : (r == 1 ? x1 : x2)));
                x2 = x[kR - 2], x1 = x[kR - 1];
#pragma unroll
                for (int r = 0; r < kR; r++) {
                    acc[r] = wadd(acc[r], mulw(c.ba[4], y2));
                    acc[r] = wadd(acc[r], mulw(c.ba[3], y1));
                    const int32_t y0 = shr_lo(acc[r], c.frac);
                    y2 = y1, y1 = y0;
                    x[r] ^= y0 & 0xff;
                }
            } else {
#pragma unroll
                for (int r = 0; r < kR; r++) {
                    const float xf = __int_as_float(0x3f800000 | (x[r] & 0x7fffff));
                    const float y0 = fs0 + fb0 * xf;
                    fs0 = (fs1 + fb1 * xf) + fa1 * y0;
                    fs1 = fb2 * xf + fa2 * y0;
                    x[r] ^= __float_as_int(y0) & 0xff;
                }
            }
        }
        t1 = clock64();
    }
    if (lid == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (lid < active) out[(blockIdx.x * blockDim.x + threadIdx.x) % 4096] = y1 + x[0] + __float_as_int(fs0);
}

template <int FORM>
int run(const char *name, int32_t *out, long long *cyc, int cus)
{
    Sec c{{17563, 35126, 17563, 2050000000 / 2, -490000000 / 2}, 30};
    for (int active : {16, 32, 64}) {
        std::printf("%-6s EXEC %2d lanes:", name, active);
        for (int wps : {1, 2, 4}) {
            hipLaunchKernelGGL((k_df1<FORM>), dim3(cus * wps), dim3(256), 0, 0, out, cyc, c, active);
            CHK(hipDeviceSynchronize());
            std::vector<long long> v(size_t(cus) * wps * 4);
            CHK(hipMemcpy(v.data(), cyc, v.size() * 8, hipMemcpyDeviceToHost));
            double s = 0;
            for (long long q : v) s += double(q);
            s /= double(v.size());
            std::printf("  %dw/SIMD: %6.1f cyc/step/wave (%5.1f per step and SIMD)", wps, s / (kBlocks * kR), s / (kBlocks * kR) / wps);
        }
        std::printf("\n");
    }
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    std::printf("%s: %d CUs, clock64() ticks; 4096 steps at N cycles = N x 4096 / 2.4 GHz: 60 -> 0.102 ms, 80 -> 0.137 ms\n", p.gcnArchName, cus);
    int32_t *out;
    long long *cyc;
    CHK(hipMalloc(&out, 4096 * 4));
    CHK(hipMalloc(&cyc, size_t(cus) * 16 * 8));
    run<0>("chain", out, cyc, cus);
    run<1>("tile", out, cyc, cus);
    run<2>("f32", out, cyc, cus);
    return 0;
}
