// LDS read forms on gfx950: which widths work at which alignment, what each costs, and whether LDS and VALU
// instructions of the waves of one SIMD issue side by side or one after the other.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds_issue.hip -o build/ubench_lds_issue && build/ubench_lds_issue
// Questions (round 6, C3 half-band decimator: 81 LDS + 200 VALU instructions per wave and round, kernel cycles ==
// 4 x their sum):
//  1. `ds_read_b64` / `ds_read_b128` at 4- and 8-byte alignment: right data?  how many cycles?  (hipcc splits such
//     loads into `ds_read2_b32`; the hardware runs in unaligned access mode under ROCm.)
//  2. the overlapping-window patterns of the FIR stages: thread t reads 2 / 4 words starting at word t (or 2 t).
//  3. N packed-f32 multiplies + N b128 reads per iteration against each alone, 1 / 2 / 4 waves per SIMD.
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

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

constexpr int kIters = 512;
constexpr int kPer = 15;          // reads per iteration (one lgkmcnt(0) per iteration)
constexpr int kRegion = 2048;     // words of LDS per wave

enum Form { B32, B64, B128, R2B32, R2B64 };
template <int F> struct W;
template <> struct W<B32> { using T = uint32_t; static constexpr int n = 1; };
template <> struct W<B64> { using T = u2; static constexpr int n = 2; };
template <> struct W<B128> { using T = u4; static constexpr int n = 4; };
template <> struct W<R2B32> { using T = u2; static constexpr int n = 2; };
template <> struct W<R2B64> { using T = u4; static constexpr int n = 4; };

// OFF: instruction offset in units of 256 bytes (the read2 forms count their offsets in elements)
template <int F, int OFF = 0>
__device__ __forceinline__ typename W<F>::T rd(uint32_t addr)
{
    typename W<F>::T v;
    if constexpr (F == B32) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF * 256));
    if constexpr (F == B64) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF * 256));
    if constexpr (F == B128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF * 256));
    if constexpr (F == R2B32) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "i"(OFF * 16), "i"(OFF * 16 + 1));
    if constexpr (F == R2B64) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "i"(OFF * 8), "i"(OFF * 8 + 1));
    return v;
}
template <int F, int I, int N>
__device__ __forceinline__ void rd_all(uint32_t addr, typename W<F>::T &v)
{
    if constexpr (I < N) {
        v = rd<F, I>(addr);
        rd_all<F, I + 1, N>(addr, v);
    }
}

// thread t of a wave reads W words at word  stride * t + mis  of the wave's region; the first read's words go to
// `probe` (checked on the host against the word index that was stored there), cycles of the loop to `cyc`
template <int F>
__global__ void k_read(uint32_t *probe, long long *cyc, int stride, int mis)
{
    extern __shared__ uint32_t lds[];
    const int lid = threadIdx.x % 64, w = threadIdx.x / 64;
    uint32_t *reg = lds + w * kRegion;
    for (int i = lid; i < kRegion; i += 64) reg[i] = uint32_t(i);
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)reg;
    const uint32_t a0 = base + 4u * uint32_t(stride * lid + mis);
    typename W<F>::T v = rd<F>(a0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (blockIdx.x == 0 && w == 0)
        for (int j = 0; j < W<F>::n; j++) probe[lid * 4 + j] = reinterpret_cast<uint32_t *>(&v)[j];
    const long long t0 = clock64();
    typename W<F>::T acc = v;
    for (int i = 0; i < kIters; i++) {
        rd_all<F, 0, kPer>(a0 + 1024u * uint32_t(i & 3), v);  // instruction offsets 0, 256, ... bytes: no address arithmetic between the reads
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    acc ^= v;
    const long long t1 = clock64();
    if (lid == 0) cyc[blockIdx.x * (blockDim.x / 64) + w] = t1 - t0;
    if (reinterpret_cast<uint32_t *>(&acc)[0] == 0xdeadbeefu) probe[0] = 1;
}

// NV packed multiplies and NL b128 reads per iteration, interleaved
template <int NV, int NL>
__global__ void k_mix(uint32_t *probe, long long *cyc)
{
    extern __shared__ uint32_t lds[];
    const int lid = threadIdx.x % 64, w = threadIdx.x / 64;
    uint32_t *reg = lds + w * kRegion;
    for (int i = lid; i < kRegion; i += 64) reg[i] = uint32_t(i);
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)reg;
    const uint32_t a0 = base + 16u * uint32_t(lid);
    uint64_t a[8];
    for (int c = 0; c < 8; c++) a[c] = 0x3f8000013f800001ull + c;
    const uint64_t b = 0x3f8000003f800000ull;
    u4 v{0, 0, 0, 0};
    constexpr int N = NV > NL ? NV : NL;
    const long long t0 = clock64();
    for (int i = 0; i < kIters; i++) {
#pragma unroll
        for (int r = 0; r < N; r++) {
            if (r < NV) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[r % 8]) : "v"(b));
            if (r < NL) asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a0 + 1024u * uint32_t(r % 4)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64();
    uint64_t r = v.x;
    for (int c = 0; c < 8; c++) r ^= a[c];
    if (lid == 0) cyc[blockIdx.x * (blockDim.x / 64) + w] = t1 - t0;
    if (r == 0xdeadbeefu) probe[0] = 1;
}

static uint32_t *g_probe;
static long long *g_cyc;

template <int F>
int run_read(const char *name, int stride, int mis, int cus)
{
    std::printf("%-12s stride %d word%s + %d: ", name, stride, stride == 1 ? " " : "s", mis);
    bool ok = true;
    for (int wpc : {4, 16}) {  // waves per CU
        const int blocks = cus, threads = wpc * 64;
        CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_read<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * kRegion * 4));
        hipLaunchKernelGGL((k_read<F>), dim3(blocks), dim3(threads), size_t(wpc) * kRegion * 4, 0, g_probe, g_cyc, stride, mis);
        CHK(hipDeviceSynchronize());
        std::vector<long long> c(size_t(blocks) * wpc);
        CHK(hipMemcpy(c.data(), g_cyc, c.size() * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (long long v : c) s += double(v);
        s /= double(c.size());
        uint32_t p[256];
        CHK(hipMemcpy(p, g_probe, sizeof p, hipMemcpyDeviceToHost));
        for (int t = 0; t < 64; t++)
            for (int j = 0; j < W<F>::n; j++) ok = ok && p[t * 4 + j] == uint32_t(stride * t + mis + j);
        // cycles the CU's LDS spends per instruction = wave's loop cycles / (instructions per wave x waves per CU)
        std::printf(" %2d waves/CU: %6.2f cyc/instr/CU (%6.1f per wave)", wpc, s / (double(kIters) * kPer * wpc), s / (double(kIters) * kPer));
    }
    std::printf("  data %s\n", ok ? "ok" : "WRONG");
    return 0;
}
template <int NV, int NL>
int run_mix(int cus)
{
    std::printf("pk_mul x%-2d + ds_read_b128 x%-2d per iteration:", NV, NL);
    for (int wpc : {4, 8, 16}) {
        CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mix<NV, NL>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * kRegion * 4));
        hipLaunchKernelGGL((k_mix<NV, NL>), dim3(cus), dim3(wpc * 64), size_t(wpc) * kRegion * 4, 0, g_probe, g_cyc);
        CHK(hipDeviceSynchronize());
        std::vector<long long> c(size_t(cus) * wpc);
        CHK(hipMemcpy(c.data(), g_cyc, c.size() * 8, hipMemcpyDeviceToHost));
        double s = 0;
        for (long long v : c) s += double(v);
        s /= double(c.size());
        std::printf("  %2d waves/CU: %7.1f cyc/iteration/wave", wpc, s / kIters);
    }
    std::printf("\n");
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    std::printf("%s: %d CUs; clock64() ticks\n", p.gcnArchName, cus);
    CHK(hipMalloc(&g_probe, 4096));
    CHK(hipMalloc(&g_cyc, size_t(cus) * 16 * 8));
    run_read<B32>("ds_read_b32", 1, 0, cus);
    run_read<B64>("ds_read_b64", 2, 0, cus);
    run_read<B64>("ds_read_b64", 2, 1, cus);
    run_read<B64>("ds_read_b64", 1, 0, cus);
    run_read<R2B32>("ds_read2_b32", 2, 0, cus);
    run_read<R2B32>("ds_read2_b32", 1, 0, cus);
    run_read<R2B32>("ds_read2_b32", 2, 1, cus);
    run_read<B128>("ds_read_b128", 4, 0, cus);
    run_read<B128>("ds_read_b128", 4, 1, cus);
    run_read<B128>("ds_read_b128", 4, 2, cus);
    run_read<B128>("ds_read_b128", 2, 0, cus);
    run_read<B128>("ds_read_b128", 1, 0, cus);
    run_read<B128>("ds_read_b128", 1, 3, cus);
    run_read<R2B64>("ds_read2_b64", 4, 0, cus);
    run_read<R2B64>("ds_read2_b64", 2, 0, cus);
    run_mix<16, 0>(cus);
    run_mix<0, 16>(cus);
    run_mix<16, 16>(cus);
    run_mix<16, 8>(cus);
    run_mix<16, 4>(cus);
    run_mix<32, 8>(cus);
    return 0;
}
