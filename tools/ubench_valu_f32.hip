// f32 VALU issue cost on gfx950 in shader cycles (clock64) per wave64 instruction and SIMD, at 1 / 2 / 4 waves per SIMD:
// the instructions the half-band FIR stages are made of (hbf_blk.h: scalar symmetric sums, packed tap multiplies, packed
// accumulation) and their mix.  Round 6: the C3 kernels turned out VALU-issue-bound in cycles and power-bound in clock.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu_f32.hip -o build/ubench_valu_f32 && build/ubench_valu_f32
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

constexpr int kIters = 1024, kCh = 8;

#define KERNEL(NAME, DECL, INIT, BODY, NINSTR)                                             \
    __global__ void NAME(uint32_t *out, long long *cyc, uint32_t seed)                     \
    {                                                                                      \
        DECL a[kCh];                                                                       \
        for (int c = 0; c < kCh; c++) a[c] = INIT + c;                                     \
        DECL b = INIT * 3 + 1;                                                             \
        const long long t0 = clock64();                                                    \
        for (int i = 0; i < kIters; i++) {                                                 \
            _Pragma("unroll") for (int c = 0; c < kCh; c++) { BODY; }                      \
        }                                                                                  \
        const long long t1 = clock64();                                                    \
        uint64_t r = 0;                                                                    \
        for (int c = 0; c < kCh; c++) r ^= uint64_t(a[c]);                                 \
        if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0; \
        if (r == 0x12345678u) out[threadIdx.x] = uint32_t(r);                              \
    }                                                                                      \
    constexpr int NAME##_n = NINSTR;

KERNEL(k_add, uint32_t, seed, asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b)), 1)
KERNEL(k_mul, uint32_t, seed, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b)), 1)
KERNEL(k_fma, uint32_t, seed, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b)), 1)
KERNEL(k_add_dpp, uint32_t, seed, asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[c]) : "v"(b)), 1)
KERNEL(k_pkadd, uint64_t, uint64_t(seed), asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b)), 1)
KERNEL(k_pkmul, uint64_t, uint64_t(seed), asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b)), 1)
KERNEL(k_pkfma, uint64_t, uint64_t(seed), asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b)), 1)
KERNEL(k_pkadd_sel, uint64_t, uint64_t(seed), asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a[c]) : "v"(b)), 1)
// the FIR pattern per pair of outputs and tap: two scalar sums, one packed multiply, one packed add
__global__ void k_mix(uint32_t *out, long long *cyc, uint32_t seed)
{
    uint64_t a[kCh], b = uint64_t(seed) * 3 + 1;
    uint32_t s0[kCh], s1[kCh], d = seed + 7;
    for (int c = 0; c < kCh; c++) a[c] = seed + c, s0[c] = seed + 2 * c, s1[c] = seed + 3 * c;
    const long long t0 = clock64();
    for (int i = 0; i < kIters; i++) {
#pragma unroll
        for (int c = 0; c < kCh; c++)
            asm volatile("v_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_pk_mul_f32 %0, %0, %3\n\tv_pk_add_f32 %0, %0, %3"
                         : "+v"(a[c]), "+v"(s0[c]), "+v"(s1[c])
                         : "v"(b), "v"(d));
    }
    const long long t1 = clock64();
    uint64_t r = 0;
    for (int c = 0; c < kCh; c++) r ^= a[c] ^ s0[c] ^ s1[c];
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
    if (r == 0x12345678u) out[threadIdx.x] = uint32_t(r);
}
constexpr int k_mix_n = 4;

static long long *g_cyc;
static uint32_t *g_out;

template <class K>
int run(const char *name, K kern, int ninstr, int cus)
{
    std::printf("%-28s", name);
    for (int wps : {1, 2, 4}) {
        const int blocks = cus * wps;  // 256 threads = one wave per SIMD
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, g_out, g_cyc, 1u);
        CHK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, g_out, g_cyc, 1u);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> c(size_t(blocks) * 4);
        CHK(hipMemcpy(c.data(), g_cyc, c.size() * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (long long v : c) s += double(v);
        s /= double(c.size());
        const double per = double(kIters) * kCh * ninstr;
        std::printf("  %dw/SIMD: %5.2f cyc/instr/SIMD (launch %.1f us)", wps, s / per / wps, ms / 5 * 1e3);
    }
    std::printf("\n");
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    std::printf("%s: %d CUs; cycles = clock64() ticks of the wave's loop / (instructions x waves per SIMD)\n", p.gcnArchName, cus);
    CHK(hipMalloc(&g_out, 4096));
    CHK(hipMalloc(&g_cyc, size_t(cus) * 16 * 8));
#define RUN(K, LABEL) run(LABEL, K, K##_n, cus)
    RUN(k_add, "v_add_f32");
    RUN(k_mul, "v_mul_f32");
    RUN(k_fma, "v_fma_f32");
    RUN(k_add_dpp, "v_add_f32 dpp row_shr:1");
    RUN(k_pkadd, "v_pk_add_f32");
    RUN(k_pkmul, "v_pk_mul_f32");
    RUN(k_pkfma, "v_pk_fma_f32");
    RUN(k_pkadd_sel, "v_pk_add_f32 op_sel");
    RUN(k_mix, "2 add + pk_mul + pk_add");
    return 0;
}
