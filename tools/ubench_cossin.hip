// cossin() formulations on gfx950: issue cost per evaluation and bit-exactness against the shipped one
// (idsp_amd/csrc/dds_dev.h: cossin_dev, a restatement of src/cossin.rs:14-67).  The C4 lock-in is VALU-issue
// bound and cossin is the largest single item of its per-sample instruction budget (profiles/NOTES.md §3, §5).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fwrapv -Iinclude -Iidsp_amd/csrc tools/ubench_cossin.hip -o build/ubench_cossin
//   build/ubench_cossin            (prints cycles per evaluation and SIMD at 1 / 2 / 4 waves per SIMD)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#include "dds_dev.h"

using namespace idsp;

#define CHK(x)                                                  \
    do {                                                        \
        hipError_t e_ = (x);                                    \
        if (e_ != hipSuccess) {                                 \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                           \
        }                                                       \
    } while (0)

constexpr int kIters = 1024, kChains = 4;

// ---- V0: the shipped formulation, 512-byte table ------------------------------------------------------------------
struct V0 {
    static constexpr int kLdsWords = 1 << kCossinDepth;
    static __device__ void fill(uint32_t *sh, int tid, int n) { fill_cossin(sh, tid, n); }
    const uint32_t *lut;
    __device__ void init(const uint32_t *sh, int) { lut = sh; }
    __device__ __forceinline__ Cplx eval(uint32_t phase) const { return cossin_dev(int32_t(phase), lut); }
};

// ---- V1: branch-free unmap (masks, v_bfi swap, xor/sub negation), same 512-byte table -----------------------------
struct V1 {
    static constexpr int kLdsWords = 1 << kCossinDepth;
    static __device__ void fill(uint32_t *sh, int tid, int n) { fill_cossin(sh, tid, n); }
    const uint32_t *lut;
    __device__ void init(const uint32_t *sh, int) { lut = sh; }
    __device__ __forceinline__ Cplx eval(uint32_t x) const
    {
        const uint32_t m29 = uint32_t(int32_t(x << 2) >> 31), m30 = uint32_t(int32_t(x << 1) >> 31), m31 = uint32_t(int32_t(x) >> 31);
        const uint32_t xx = x ^ m29;
        const uint32_t lookup = lut[(xx >> 22) & 0x7fu];
        const int32_t p = int32_t((xx >> 7) & 0x7fffu) - 16384;
        const int32_t dphi = (p * 51471) >> 16;
        int32_t c = int32_t(lookup & 0xffffu) + (1 << 16);
        int32_t s = int32_t(lookup >> 16);
        const int32_t dcos = (s * dphi) >> 7, dsin = (c * dphi) >> 8;
        c = int32_t(uint32_t(c) << 14) - dcos;
        s = int32_t(uint32_t(s) << 15) + dsin;
        const uint32_t msw = m29 ^ m30, mc = m30 ^ m31;
        uint32_t re = (msw & uint32_t(s)) | (~msw & uint32_t(c));
        uint32_t im = (msw & uint32_t(c)) | (~msw & uint32_t(s));
        re = (re ^ mc) - mc;
        im = (im ^ m31) - m31;
        return Cplx{int32_t(re), int32_t(im)};
    }
};

// ---- V2: pre-shifted 16-byte entries {c << 14, s << 15, c << 8, s << 9}, 16 copies (a private bank quartet per lane of
//          every ds_read_b128 lane group), both interpolation products as one v_mul_hi_i32 each --------------------------
struct V2 {
    static constexpr int kCopies = 16;
    static constexpr int kLdsWords = (1 << kCossinDepth) * kCopies * 4;  // 32 KiB
    static __device__ void fill(uint32_t *sh, int tid, int n)
    {
        for (int i = tid; i < (1 << kCossinDepth) * kCopies; i += n) {
            const uint32_t lookup = d_cossin_table[i / kCopies];
            const uint32_t c = (lookup & 0xffffu) + (1u << 16), s = lookup >> 16;
            uint32_t *e = sh + i * 4;
            e[0] = c << 14, e[1] = s << 15, e[2] = c << 8, e[3] = s << 9;
        }
    }
    uint32_t base;  // LDS byte address of this lane's copy
    const uint32_t *sh_;
    __device__ void init(const uint32_t *sh, int lid) { sh_ = sh, base = uint32_t(lid % kCopies) * 16u; }
    __device__ __forceinline__ Cplx eval(uint32_t x) const
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint32_t m29 = uint32_t(int32_t(x << 2) >> 31), m30 = uint32_t(int32_t(x << 1) >> 31), m31 = uint32_t(int32_t(x) >> 31);
        const uint32_t xx = x ^ m29;
        const uint32_t off = ((xx >> 14) & 0x7f00u) | base;  // entry (xx >> 22) & 127, 256 bytes per entry group
        const u32x4 e = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(sh_) + off);
        const int32_t t = int32_t((xx >> 7) & 0x7fffu) * 51471 - 16384 * 51471;
        const int32_t d16 = int32_t(uint32_t(t) & 0xffff0000u);  // dphi << 16
        const int32_t dcos = __mulhi(int32_t(e.w), d16), dsin = __mulhi(int32_t(e.z), d16);
        const uint32_t c = e.x - uint32_t(dcos), s = e.y + uint32_t(dsin);
        const uint32_t msw = m29 ^ m30, mc = m30 ^ m31;
        uint32_t re = (msw & s) | (~msw & c);
        uint32_t im = (msw & c) | (~msw & s);
        re = (re ^ mc) - mc;
        im = (im ^ m31) - m31;
        return Cplx{int32_t(re), int32_t(im)};
    }
};

// ---- V3: V2's table and products, unmap by compare + v_cndmask (what does a select cost?) ----------------------------
struct V3 : V2 {
    __device__ __forceinline__ Cplx eval(uint32_t x) const
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint32_t xx = (x & (1u << 29)) ? ~x : x;
        const uint32_t off = ((xx >> 14) & 0x7f00u) | base;
        const u32x4 e = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(sh_) + off);
        const int32_t t = int32_t((xx >> 7) & 0x7fffu) * 51471 - 16384 * 51471;
        const int32_t d16 = int32_t(uint32_t(t) & 0xffff0000u);
        const int32_t dcos = __mulhi(int32_t(e.w), d16), dsin = __mulhi(int32_t(e.z), d16);
        int32_t c = int32_t(e.x - uint32_t(dcos)), s = int32_t(e.y + uint32_t(dsin));
        const uint32_t oct = x ^ (x >> 1);
        if (oct & (1u << 29)) {
            const int32_t tt = c;
            c = s, s = tt;
        }
        if (oct & (1u << 30)) c = int32_t(0u - uint32_t(c));
        if (oct & (1u << 31)) s = int32_t(0u - uint32_t(s));
        return Cplx{c, s};
    }
};

// ---- V4: 8-byte entries {c << 14, s << 15}, 32 copies, products against dphi << 10 -----------------------------------
struct V4 {
    static constexpr int kCopies = 32;
    static constexpr int kLdsWords = (1 << kCossinDepth) * kCopies * 2;  // 32 KiB
    static __device__ void fill(uint32_t *sh, int tid, int n)
    {
        for (int i = tid; i < (1 << kCossinDepth) * kCopies; i += n) {
            const uint32_t lookup = d_cossin_table[i / kCopies];
            const uint32_t c = (lookup & 0xffffu) + (1u << 16), s = lookup >> 16;
            sh[2 * i] = c << 14, sh[2 * i + 1] = s << 15;
        }
    }
    uint32_t base;
    const uint32_t *sh_;
    __device__ void init(const uint32_t *sh, int lid) { sh_ = sh, base = uint32_t(lid % kCopies) * 8u; }
    __device__ __forceinline__ Cplx eval(uint32_t x) const
    {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const uint32_t m29 = uint32_t(int32_t(x << 2) >> 31), m30 = uint32_t(int32_t(x << 1) >> 31), m31 = uint32_t(int32_t(x) >> 31);
        const uint32_t xx = x ^ m29;
        const uint32_t off = ((xx >> 14) & 0x7f00u) | base;
        const u32x2 e = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const char *>(sh_) + off);
        const int32_t t = int32_t((xx >> 7) & 0x7fffu) * 51471 - 16384 * 51471;
        const int32_t d10 = int32_t(uint32_t(t >> 6) & 0xfffffc00u);  // dphi << 10
        const int32_t dcos = __mulhi(int32_t(e.y), d10), dsin = __mulhi(int32_t(e.x), d10);
        const uint32_t c = e.x - uint32_t(dcos), s = e.y + uint32_t(dsin);
        const uint32_t msw = m29 ^ m30, mc = m30 ^ m31;
        uint32_t re = (msw & s) | (~msw & c);
        uint32_t im = (msw & c) | (~msw & s);
        re = (re ^ mc) - mc;
        im = (im ^ m31) - m31;
        return Cplx{int32_t(re), int32_t(im)};
    }
};

// ---- V5: V2 with the octant swap as one v_bitop3_b32 select per component (no compare / v_cndmask) --------------------
__device__ __forceinline__ uint32_t sel3(uint32_t m, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA); }  // m ? a : b, bitwise
struct V5 : V2 {
    __device__ __forceinline__ Cplx eval(uint32_t x) const
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint32_t m29 = uint32_t(__builtin_amdgcn_sbfe(int32_t(x), 29, 1)), m30 = uint32_t(__builtin_amdgcn_sbfe(int32_t(x), 30, 1)), m31 = uint32_t(int32_t(x) >> 31);
        const uint32_t xx = x ^ m29;
        const uint32_t off = ((xx >> 14) & 0x7f00u) | base;
        const u32x4 e = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(sh_) + off);
        const int32_t t = int32_t(__builtin_amdgcn_ubfe(xx, 7, 15)) * 51471 - 16384 * 51471;
        const int32_t d16 = int32_t(uint32_t(t) & 0xffff0000u);
        const int32_t dcos = __mulhi(int32_t(e.w), d16), dsin = __mulhi(int32_t(e.z), d16);
        const uint32_t c = e.x - uint32_t(dcos), s = e.y + uint32_t(dsin);
        const uint32_t msw = m29 ^ m30, mc = m30 ^ m31;
        uint32_t re = sel3(msw, s, c), im = sel3(msw, c, s);
        re = (re ^ mc) - mc;
        im = (im ^ m31) - m31;
        return Cplx{int32_t(re), int32_t(im)};
    }
};

// ---- V6: V5 with the masks taken from ONE shifted copy of the phase (x << 2 puts bit 29 in the sign) and the negations
//          as xor + subtract of 0 / 1 bits --------------------------------------------------------------------------------
struct V6 : V2 {
    __device__ __forceinline__ Cplx eval(uint32_t x) const
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint32_t o = x ^ (x >> 1);  // octant bits 29 (swap), 30 (negate cos), 31 (negate sin)
        const uint32_t m29 = uint32_t(__builtin_amdgcn_sbfe(int32_t(x), 29, 1));
        const uint32_t xx = x ^ m29;
        const uint32_t off = ((xx >> 14) & 0x7f00u) | base;
        const u32x4 e = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(sh_) + off);
        const int32_t t = int32_t(__builtin_amdgcn_ubfe(xx, 7, 15)) * 51471 - 16384 * 51471;
        const int32_t d16 = int32_t(uint32_t(t) & 0xffff0000u);
        const int32_t dcos = __mulhi(int32_t(e.w), d16), dsin = __mulhi(int32_t(e.z), d16);
        const uint32_t c = e.x - uint32_t(dcos), s = e.y + uint32_t(dsin);
        const uint32_t msw = uint32_t(__builtin_amdgcn_sbfe(int32_t(o), 29, 1)), mc = uint32_t(__builtin_amdgcn_sbfe(int32_t(o), 30, 1)), ms = uint32_t(int32_t(o) >> 31);
        uint32_t re = sel3(msw, s, c), im = sel3(msw, c, s);
        re = (re ^ mc) - mc;
        im = (im ^ ms) - ms;
        return Cplx{int32_t(re), int32_t(im)};
    }
};

// ---- V7: the octant logic in the table (dds_dev.h cossin_circle): 1024 x 16-byte entries indexed by the top ten phase
//          bits, each output component the high word of one v_mad_i64_i32 ----------------------------------------------
struct V7 {
    static constexpr int kLdsWords = kCosCircleWords;
    static __device__ void fill(uint32_t *sh, int tid, int n) { fill_cossin_circle(sh, tid, n); }
    const uint32_t *tab;
    __device__ void init(const uint32_t *sh, int) { tab = sh; }
    __device__ __forceinline__ Cplx eval(uint32_t phase) const { return cossin_circle(phase, tab); }
};

template <class V>
__global__ __launch_bounds__(256) void k_time(uint32_t *out, uint32_t seed)
{
    extern __shared__ uint32_t sh[];
    V::fill(sh, threadIdx.x, 256);
    __syncthreads();
    V v;
    v.init(sh, threadIdx.x % 64);
    uint32_t ph[kChains], inc[kChains], acc = 0;
    for (int c = 0; c < kChains; c++) ph[c] = seed * 2654435761u + threadIdx.x * 40503u + c * 977u + blockIdx.x, inc[c] = 0x01234567u * (threadIdx.x + 1) + c;
    for (int i = 0; i < kIters; i++) {
#pragma unroll
        for (int c = 0; c < kChains; c++) {
            ph[c] += inc[c];
            const Cplx r = v.eval(ph[c]);
            acc += uint32_t(r.re) ^ uint32_t(r.im);
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <class V>
__global__ __launch_bounds__(256) void k_values(Cplx *out, uint32_t start, uint32_t stride, uint32_t n)
{
    extern __shared__ uint32_t sh[];
    V::fill(sh, threadIdx.x, 256);
    __syncthreads();
    V v;
    v.init(sh, threadIdx.x % 64);
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = v.eval(start + i * stride);
}

template <class V>
int run(const char *name, uint32_t *out, Cplx *va, Cplx *vb, int cus, double ghz)
{
    const size_t lds = size_t(V::kLdsWords) * 4;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_time<V>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_values<V>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    // exactness: 2^24 phases at an odd stride (every octant, every table entry, every interpolation offset class) + the edges
    const uint32_t n = 1u << 24;
    size_t bad = 0;
    for (uint32_t start : {0u, 0x1fffff00u, 0x3fffff00u, 0x5fffff00u, 0x7fffff00u, 0x9fffff00u, 0xbfffff00u, 0xdfffff00u, 0xffffff00u, 0x003fff00u,
                           0x203fff00u, 12345u, 777u, 99u}) {
        const bool sweep = start == 12345u || start == 777u || start == 99u;
        const uint32_t stride = start == 12345u ? 257u : start == 777u ? 255u : start == 99u ? 65521u : 1u, cnt = sweep ? n : 512u;
        hipLaunchKernelGGL((k_values<V0>), dim3(1024), dim3(256), size_t(V0::kLdsWords) * 4, 0, va, start, stride, cnt);
        hipLaunchKernelGGL((k_values<V>), dim3(1024), dim3(256), lds, 0, vb, start, stride, cnt);
        CHK(hipDeviceSynchronize());
        std::vector<Cplx> a(cnt), b(cnt);
        CHK(hipMemcpy(a.data(), va, cnt * sizeof(Cplx), hipMemcpyDeviceToHost));
        CHK(hipMemcpy(b.data(), vb, cnt * sizeof(Cplx), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < cnt; i++) bad += a[i].re != b[i].re || a[i].im != b[i].im;
    }
    std::printf("%-44s mismatches %zu;", name, bad);
    for (int wps : {1, 2, 4}) {
        if (lds * wps > 160 * 1024) {
            std::printf("  %dw/SIMD: (LDS)", wps);
            continue;
        }
        const int blocks = cus * wps;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0));
        CHK(hipEventCreate(&e1));
        hipLaunchKernelGGL((k_time<V>), dim3(blocks), dim3(256), lds, 0, out, 1u);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k_time<V>), dim3(blocks), dim3(256), lds, 0, out, 1u);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double cyc = ms / 5 * 1e-3 * ghz * 1e9;
        std::printf("  %dw/SIMD: %6.1f cyc/eval/SIMD", wps, cyc / (double(kIters) * kChains * wps));
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
    std::printf("%s: %d CUs, %.2f GHz nominal; cycles per cossin evaluation of one wave64 and SIMD (incl. the phase add and 2 consuming ops)\n",
                p.gcnArchName, cus, ghz);
    uint32_t *out;
    Cplx *va, *vb;
    CHK(hipMalloc(&out, 4096));
    CHK(hipMalloc(&va, sizeof(Cplx) << 24));
    CHK(hipMalloc(&vb, sizeof(Cplx) << 24));
    run<V0>("V0 shipped (branches, 512 B table)", out, va, vb, cus, ghz);
    run<V1>("V1 branch-free unmap, 512 B table", out, va, vb, cus, ghz);
    run<V2>("V2 16 B pre-shifted entries x16, mul_hi", out, va, vb, cus, ghz);
    run<V3>("V3 = V2 with compare/select unmap", out, va, vb, cus, ghz);
    run<V4>("V4 8 B pre-shifted entries x32, mul_hi", out, va, vb, cus, ghz);
    run<V5>("V5 = V2 + v_bitop3 selects, xor masks", out, va, vb, cus, ghz);
    run<V6>("V6 = V5 with masks from the octant word", out, va, vb, cus, ghz);
    run<V7>("V7 octant logic in a 16 KiB table, 2 mad64", out, va, vb, cus, ghz);
    return 0;
}
