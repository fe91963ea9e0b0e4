// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction and SIMD at 1, 2 and 4
// waves per SIMD, for the instructions the integer filter paths are made of.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o build/ubench_valu && build/ubench_valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdint>
#include <vector>

#define CHK(x)                                                                    \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 1;                                                             \
        }                                                                         \
    } while (0)

constexpr int kIters = 2048;
constexpr int kChains = 8;

// one asm statement = one instruction on chain c; 8 independent chains hide the dependent latency
#define BODY32(INS)                                                     \
    uint32_t a[kChains];                                                \
    for (int c = 0; c < kChains; c++) a[c] = seed + c;                  \
    uint32_t b = seed * 3 + 1;                                          \
    for (int i = 0; i < kIters; i++) {                                  \
        _Pragma("unroll") for (int c = 0; c < kChains; c++) asm volatile(INS : "+v"(a[c]) : "v"(b)); \
    }                                                                   \
    uint32_t r = 0;                                                     \
    for (int c = 0; c < kChains; c++) r ^= a[c];                        \
    if (r == 0x12345678u) out[threadIdx.x] = r;

__global__ void k_add(uint32_t *out, uint32_t seed) { BODY32("v_add_u32 %0, %0, %1") }
__global__ void k_mullo(uint32_t *out, uint32_t seed) { BODY32("v_mul_lo_u32 %0, %0, %1") }
__global__ void k_mulhi(uint32_t *out, uint32_t seed) { BODY32("v_mul_hi_i32 %0, %0, %1") }
__global__ void k_fma(uint32_t *out, uint32_t seed) { BODY32("v_fma_f32 %0, %0, %1, %1") }
__global__ void k_mul24(uint32_t *out, uint32_t seed) { BODY32("v_mad_u32_u24 %0, %0, %1, %1") }
__global__ void k_lshladd(uint32_t *out, uint32_t seed) { BODY32("v_lshl_add_u32 %0, %0, 3, %1") }
__global__ void k_alignbit(uint32_t *out, uint32_t seed) { BODY32("v_alignbit_b32 %0, %0, %1, 7") }
__global__ void k_cndmask(uint32_t *out, uint32_t seed) { BODY32("v_cndmask_b32 %0, %0, %1, vcc") }
__global__ void k_addc(uint32_t *out, uint32_t seed) { BODY32("v_addc_co_u32 %0, vcc, %0, %1, vcc") }
__global__ void k_mov_dpp(uint32_t *out, uint32_t seed) { BODY32("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }

__global__ void k_mad64(uint32_t *out, uint32_t seed)
{
    uint64_t a[kChains];
    for (int c = 0; c < kChains; c++) a[c] = seed + c;
    uint32_t b = seed * 3 + 1, d = seed + 7;
    for (int i = 0; i < kIters; i++) {
#pragma unroll
        for (int c = 0; c < kChains; c++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a[c]) : "v"(b), "v"(d) : "vcc");
    }
    uint64_t r = 0;
    for (int c = 0; c < kChains; c++) r ^= a[c];
    if (r == 0x12345678u) out[threadIdx.x] = uint32_t(r);
}
__global__ void k_fma64(uint32_t *out, uint32_t seed)
{
    double a[kChains];
    for (int c = 0; c < kChains; c++) a[c] = seed + c;
    double b = seed * 3 + 1;
    for (int i = 0; i < kIters; i++) {
#pragma unroll
        for (int c = 0; c < kChains; c++) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b));
    }
    double r = 0;
    for (int c = 0; c < kChains; c++) r += a[c];
    if (r == 0.12345) out[threadIdx.x] = 1;
}
__global__ void k_pkfma(uint32_t *out, uint32_t seed)
{
    uint64_t a[kChains];
    for (int c = 0; c < kChains; c++) a[c] = seed + c;
    uint64_t b = seed * 3 + 1;
    for (int i = 0; i < kIters; i++) {
#pragma unroll
        for (int c = 0; c < kChains; c++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b));
    }
    uint64_t r = 0;
    for (int c = 0; c < kChains; c++) r ^= a[c];
    if (r == 0x12345678u) out[threadIdx.x] = uint32_t(r);
}

// ---- dependent-chain latency: CH independent chains per wave (1 = every instruction waits for the previous one)
template <int CH>
__global__ void k_lat_add64(uint32_t *out, uint32_t seed)
{
    uint64_t a[CH];
    for (int c = 0; c < CH; c++) a[c] = seed + c;
    uint64_t b = seed * 3 + 1;
    for (int i = 0; i < kIters * kChains / CH; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[c]) : "v"(b));
    }
    uint64_t r = 0;
    for (int c = 0; c < CH; c++) r ^= a[c];
    if (r == 0x12345678u) out[threadIdx.x] = uint32_t(r);
}
template <int CH>
__global__ void k_lat_addc_pair(uint32_t *out, uint32_t seed)  // 64-bit add as v_add_co_u32 + v_addc_co_u32 (2 instructions)
{
    uint32_t lo[CH], hi[CH];
    for (int c = 0; c < CH; c++) lo[c] = seed + c, hi[c] = seed;
    uint32_t b = seed * 3 + 1, d = seed + 5;
    for (int i = 0; i < kIters * kChains / CH; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++)
            asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo[c]), "+v"(hi[c]) : "v"(b), "v"(d) : "vcc");
    }
    uint32_t r = 0;
    for (int c = 0; c < CH; c++) r ^= lo[c] ^ hi[c];
    if (r == 0x12345678u) out[threadIdx.x] = r;
}
template <int CH>
__global__ void k_lat_mad64(uint32_t *out, uint32_t seed)  // accumulate chain: C operand depends on the previous result
{
    uint64_t a[CH];
    for (int c = 0; c < CH; c++) a[c] = seed + c;
    uint32_t b = seed * 3 + 1, d = seed + 7;
    for (int i = 0; i < kIters * kChains / CH; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a[c]) : "v"(b), "v"(d) : "vcc");
    }
    uint64_t r = 0;
    for (int c = 0; c < CH; c++) r ^= a[c];
    if (r == 0x12345678u) out[threadIdx.x] = uint32_t(r);
}
template <int CH>
__global__ void k_lat_mad64_mul(uint32_t *out, uint32_t seed)  // multiplicand chain: the high word of the result feeds the next multiply
{
    uint64_t a[CH];
    for (int c = 0; c < CH; c++) a[c] = seed + c;
    uint32_t d = seed + 7;
    uint64_t z = seed;
    for (int i = 0; i < kIters * kChains / CH; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t h = uint32_t(a[c] >> 32);
            asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(a[c]) : "v"(h), "v"(d), "v"(z) : "vcc");
        }
    }
    uint64_t r = 0;
    for (int c = 0; c < CH; c++) r ^= a[c];
    if (r == 0x12345678u) out[threadIdx.x] = uint32_t(r);
}
template <int CH>
__global__ void k_lat_add32(uint32_t *out, uint32_t seed)
{
    uint32_t a[CH];
    for (int c = 0; c < CH; c++) a[c] = seed + c;
    uint32_t b = seed * 3 + 1;
    for (int i = 0; i < kIters * kChains / CH; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
    }
    uint32_t r = 0;
    for (int c = 0; c < CH; c++) r ^= a[c];
    if (r == 0x12345678u) out[threadIdx.x] = r;
}

template <class K>
int run(const char *name, K kern, uint32_t *out, int cus, double ghz)
{
    std::printf("%-16s", name);
    for (int wps : {1, 2, 4}) {
        // blocks of 256 threads = 4 waves = one per SIMD; wps blocks per CU
        const int blocks = cus * wps;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double cyc = ms / 5 * 1e-3 * ghz * 1e9;           // cycles for the launch
        const double instr_per_simd = double(kIters) * kChains * wps;
        std::printf("  %dw/SIMD: %5.2f cyc/instr/SIMD", wps, cyc / instr_per_simd);
    }
    std::printf("\n");
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate * 1e-6;
    std::printf("%s: %d CUs, %.2f GHz nominal\n", p.gcnArchName, cus, ghz);
    uint32_t *out;
    CHK(hipMalloc(&out, 4096));
    run("v_add_u32", k_add, out, cus, ghz);
    run("v_fma_f32", k_fma, out, cus, ghz);
    run("v_pk_fma_f32", k_pkfma, out, cus, ghz);
    run("v_fma_f64", k_fma64, out, cus, ghz);
    run("v_mad_u32_u24", k_mul24, out, cus, ghz);
    run("v_lshl_add_u32", k_lshladd, out, cus, ghz);
    run("v_alignbit_b32", k_alignbit, out, cus, ghz);
    run("v_cndmask_b32", k_cndmask, out, cus, ghz);
    run("v_addc_co_u32", k_addc, out, cus, ghz);
    run("v_mov_b32_dpp", k_mov_dpp, out, cus, ghz);
    run("v_mul_lo_u32", k_mullo, out, cus, ghz);
    run("v_mul_hi_i32", k_mulhi, out, cus, ghz);
    run("v_mad_i64_i32", k_mad64, out, cus, ghz);
    std::printf("dependent chains per wave (same instruction totals; 1 chain = pure latency)\n");
    run("add_u32 x1", k_lat_add32<1>, out, cus, ghz);
    run("add_u32 x2", k_lat_add32<2>, out, cus, ghz);
    run("lshl_add_u64 x1", k_lat_add64<1>, out, cus, ghz);
    run("lshl_add_u64 x2", k_lat_add64<2>, out, cus, ghz);
    run("lshl_add_u64 x4", k_lat_add64<4>, out, cus, ghz);
    run("lshl_add_u64 x8", k_lat_add64<8>, out, cus, ghz);
    run("add_co+addc x1", k_lat_addc_pair<1>, out, cus, ghz);
    run("add_co+addc x2", k_lat_addc_pair<2>, out, cus, ghz);
    run("add_co+addc x8", k_lat_addc_pair<8>, out, cus, ghz);
    run("mad_i64 acc x1", k_lat_mad64<1>, out, cus, ghz);
    run("mad_i64 acc x2", k_lat_mad64<2>, out, cus, ghz);
    run("mad_i64 acc x4", k_lat_mad64<4>, out, cus, ghz);
    run("mad_i64 mul x1", k_lat_mad64_mul<1>, out, cus, ghz);
    run("mad_i64 mul x2", k_lat_mad64_mul<2>, out, cus, ghz);
    return 0;
}
