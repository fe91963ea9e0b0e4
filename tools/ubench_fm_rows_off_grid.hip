// ubench_fm_rows_off_grid.hip — round 5: what FrameMajor rows that start off the 64-byte grid cost the LDS-DMA walk, and whether requesting
// and storing ALIGNED windows instead wins it back.  A copy with the sweep kernel's skeleton (one round of 256 workgroups, a ring of 7 tiles of
// 8 one-KiB segments, hand-counted vmcnt, outputs staged in LDS), 240 lanes per workgroup so that the 64-byte aligned 1 KiB window around a
// block's 960 bytes always holds them:
//   MODE 0  requests and 16-byte stores at the block's own addresses (what fm_sweep.h does on such rows: each lane's 16 bytes straddle)
//   MODE 1  requests of the aligned window (the block's words sit `o` words into the segment, o = row start mod 16), stores as MODE 0
//   MODE 2  aligned window requests AND aligned 16-byte stores; the up to 3 words at either end of the block by one 4-byte store
// over row pitches 61440 (on the grid), +1, +4, +8 lanes, plain and XCD-contiguous block order.
//   hipcc --offload-arch=gfx950 -O3 -Iidsp_amd/csrc tools/ubench_fm_rows_off_grid.hip -o build/ubench_fm_rows_off_grid && build/ubench_fm_rows_off_grid
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "lds_dma.h"

using namespace idsp;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#ifndef UB_BW
#define UB_BW 240  // -DUB_BW=256: full 1 KiB segments (MODE 0 only: the aligned window of MODE 1 / 2 needs 16 spare words)
#endif
constexpr int NB = 7, TS = 8, BW = UB_BW, SEG = 256;

template <int MODE>
__global__ __launch_bounds__(256) void k_rows(const uint32_t *x, uint32_t *y, const size_t pitch, const size_t frames, uint32_t *dummy, const unsigned xcdc)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *tin = smem;                   // [NB][TS][SEG]
    uint32_t *tout = smem + NB * TS * SEG;  // [2][TS][SEG]
    const int tid = threadIdx.x, lid = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)smem;
    const size_t G = gridDim.x;
    size_t w = blockIdx.x;
    if (xcdc) {
        const size_t q = G / 8, r = G % 8, j = blockIdx.x % 8;
        w = j * q + (j < r ? j : r) + blockIdx.x / 8;
    }
    const size_t first = w * BW;
    const size_t ntiles = frames / TS;
    // MODE 3 / 4 / 5 (full 256-lane blocks): the wave's request (3), its store (4) or both (5) as two instructions, lanes 0..59 and 60..63
    constexpr bool SPLIT_RQ = MODE == 3 || MODE == 5, SPLIT_ST = MODE == 4 || MODE == 5;
    constexpr int RPT = SPLIT_RQ ? 4 : 2;           // requests per wave and tile
    constexpr int SPT = MODE == 2 || SPLIT_ST ? 4 : 2;  // stores per wave and tile
    constexpr int K = (NB - 1) * (RPT + SPT), K0 = RPT * (NB - 1);

    auto issue = [&](size_t t) {
        if (t >= ntiles) t = ntiles - 1;  // static request count: the last tile once more
        const int slot = int(t % NB);
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int g = wave + 4 * j;
            const size_t S = (t * TS + g) * pitch + first;
            const size_t A = MODE == 0 || MODE >= 3 ? S : S & ~size_t(15);
            if constexpr (SPLIT_RQ) {
                if (lid < 60) glds16_s(x + A, uint32_t(lid * 16), lds_base + uint32_t(((slot * TS + g) * SEG) * 4));
                if (lid >= 60) glds16_s(x + A, uint32_t(lid * 16), lds_base + uint32_t(((slot * TS + g) * SEG) * 4));
            } else if ((MODE != 0 && MODE < 3) || lid < BW / 4) {
#ifdef UB_VADDR  // the request's address as a 64-bit VGPR pair per lane (round 3's kernel) instead of SGPR base + 32-bit lane offset (fm_sweep.h)
                glds16(x + A + lid * 4, lds_base + uint32_t(((slot * TS + g) * SEG) * 4));
#else
                glds16_s(x + A, uint32_t(lid * 16), lds_base + uint32_t(((slot * TS + g) * SEG) * 4));
#endif
            }
        }
    };
    for (size_t t = 0; t + 1 < NB; t++) issue(t);
    for (size_t i = 0; i < ntiles; i++) {
        issue(i + NB - 1);
        if (i < NB)
            wait_vmcnt<K0>();
        else
            wait_vmcnt<K>();
        __syncthreads();
        const int slot = int(i % NB);
        uint32_t *o = tout + (i & 1) * TS * SEG;
        if (tid < BW) {
#pragma unroll
            for (int g = 0; g < TS; g++) {
                const size_t S = (i * TS + g) * pitch + first;
                const unsigned off = MODE == 0 || MODE >= 3 ? 0u : unsigned(S & 15);
                const unsigned offo = MODE == 2 ? off : 0u;
                o[g * SEG + offo + tid] = tin[(slot * TS + g) * SEG + off + tid] + 1u;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int g = wave + 4 * j;
            const size_t S = (i * TS + g) * pitch + first;
            if constexpr (SPLIT_ST) {
                if (lid < 60) __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(o + g * SEG + lid * 4), reinterpret_cast<u32x4 *>(y + S + lid * 4));
                if (lid >= 60) __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(o + g * SEG + lid * 4), reinterpret_cast<u32x4 *>(y + S + lid * 4));
            } else if constexpr (MODE != 2) {
                if (lid < BW / 4) __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(o + g * SEG + lid * 4), reinterpret_cast<u32x4 *>(y + S + lid * 4));
            } else {
                const size_t A = S & ~size_t(15);
                const unsigned off = unsigned(S - A);
                const size_t c0 = A + size_t(lid) * 4;  // this lane's aligned piece
                if (c0 >= S && c0 + 4 <= S + BW) __builtin_nontemporal_store(*reinterpret_cast<const u32x4 *>(o + g * SEG + lid * 4), reinterpret_cast<u32x4 *>(y + c0));
                // the words before the first whole piece (lanes 0..2) and behind the last one (lanes 3..5); absent ones go to a dummy line
                if (lid < 6) {
                    const unsigned nl = (4u - (off & 3u)) & 3u, nr = (off + BW) & 3u;
                    const unsigned e = lid < 3 ? unsigned(lid) : unsigned(lid - 3);
                    const bool on = lid < 3 ? e < nl : e < nr;
                    const unsigned word = lid < 3 ? off + e : ((off + BW) & ~3u) + e;  // in the segment
                    uint32_t *dst = on ? y + A + word : dummy + size_t(blockIdx.x) * 16 + lid;
                    __builtin_nontemporal_store(o[g * SEG + (on ? word : 0u)], dst);
                }
            }
        }
    }
}

template <class F>
static float med_ms(F &&f, int it)
{
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    std::vector<float> ts;
    for (int i = 0; i < it + 3; i++) {
        hipEventRecord(a);
        f();
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (i >= 3) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

template <int MODE>
static int run(const uint32_t *x, uint32_t *y, uint32_t *dummy, size_t pitch, size_t frames, unsigned xcdc, std::vector<uint32_t> &hx, std::vector<uint32_t> &hy)
{
    const size_t lanes = 256 * size_t(BW), bytes_lds = size_t(NB + 2) * TS * SEG * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rows<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes_lds)));
    const size_t words = pitch * frames;
    CK(hipMemset(y, 0xAB, words * 4));
    auto go = [&] { hipLaunchKernelGGL((k_rows<MODE>), dim3(256), dim3(256), bytes_lds, 0, x, y, pitch, frames, dummy, xcdc); };
    go();
    CK(hipDeviceSynchronize());
    // check: rows 0, 1, 7, 8, 1000, last — copied lanes + 1, the words between the rows untouched
    size_t bad = 0;
    for (size_t f : {size_t(0), size_t(1), size_t(7), size_t(8), size_t(1001), frames - 1}) {
        CK(hipMemcpy(hy.data(), y + f * pitch, pitch * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hx.data(), x + f * pitch, pitch * 4, hipMemcpyDeviceToHost));
        for (size_t l = 0; l < pitch; l++) bad += hy[l] != (l < lanes ? hx[l] + 1u : 0xABABABABu);
    }
    const float ms = med_ms(go, 15);
    const double gb = 2.0 * 4.0 * double(lanes) * double(frames) / 1e9;
    printf("{\"mode\": %d, \"pitch\": %zu, \"xcdc\": %u, \"ms\": %.4f, \"frac_hbm_peak\": %.3f, \"mismatches\": %zu}\n", MODE, pitch, xcdc, ms, gb / (ms * 1e-3) / 8000.0, bad);
    fflush(stdout);
    return 0;
}

int main()
{
    const size_t frames = 4096, lanes = 256 * size_t(BW), maxpitch = lanes + 16;
    uint32_t *x, *y, *dummy;
    CK(hipMalloc(&x, maxpitch * frames * 4 + 8192));
    CK(hipMalloc(&y, maxpitch * frames * 4 + 8192));
    CK(hipMalloc(&dummy, 256 * 64));
    std::vector<uint32_t> hx(maxpitch * frames / 64 + 16), hy(maxpitch + 16);
    {
        std::vector<uint32_t> init(maxpitch * frames + 2048);
        uint32_t s = 12345;
        for (auto &v : init) v = (s = s * 1664525u + 1013904223u) >> 1;
        CK(hipMemcpy(x, init.data(), init.size() * 4, hipMemcpyHostToDevice));
    }
    hx.resize(maxpitch + 16);
    for (size_t pitch : {lanes, lanes + 1, lanes + 4, lanes + 8}) {
        for (unsigned xcdc : {0u, 1u}) {
            if (run<0>(x, y, dummy, pitch, frames, xcdc, hx, hy)) return 1;
            if (BW <= 240 && run<1>(x, y, dummy, pitch, frames, xcdc, hx, hy)) return 1;
            if (BW <= 240 && run<2>(x, y, dummy, pitch, frames, xcdc, hx, hy)) return 1;
            if (BW == 256 && run<3>(x, y, dummy, pitch, frames, xcdc, hx, hy)) return 1;
            if (BW == 256 && run<4>(x, y, dummy, pitch, frames, xcdc, hx, hy)) return 1;
            if (BW == 256 && run<5>(x, y, dummy, pitch, frames, xcdc, hx, hy)) return 1;
        }
    }
    return 0;
}
