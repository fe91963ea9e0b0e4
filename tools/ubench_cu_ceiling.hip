// What a CU can move, by access pattern and by the number of resident waves (round 6: the LaneMajor biquad runs at ~21 GB/s per CU with three
// waves or with four; is that the kernel or the chip?).  One wave per workgroup, 32 KiB tiles of 32 x `global_load_dwordx4` / `global_store_dwordx4`
// (nontemporal), the next tile's loads issued before this tile's stores — the skeleton of stream_lane_major_staged without LDS and arithmetic:
//   scattered  wave w owns 64 rows of `row_bytes` (16 KiB = 4096 frames x 4 B), rows 64 w .. 64 w + 63; a tile is a 512-byte run of each of them
//              (instruction j: rows 2 j, 2 j + 1) — what LANE_MAJOR is: 65536 concurrent 512-byte runs 16 KiB apart
//   dense      tile t of wave w is the 32 KiB at ((t G + w) x 32 KiB): the G waves together walk the buffer front to back — what FRAME_MAJOR is
// each as read-only (the sum of what was read is stored once at the end), write-only, and copy; grids of 768 / 896 / 1024 waves (3, 3.5, 4 per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_cu_ceiling.hip -o build/ubench_cu_ceiling && build/ubench_cu_ceiling
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHK(x)                                                  \
    do {                                                        \
        hipError_t e_ = (x);                                    \
        if (e_ != hipSuccess) {                                 \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_)); \
            return 1;                                           \
        }                                                       \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4 gvec;

constexpr int kNI = 32;           // instructions per tile
constexpr size_t kTile = 32768;   // bytes per tile and wave
constexpr size_t kRow = 16384;    // bytes per row (scattered)
enum { READ = 1, WRITE = 2 };

__device__ __forceinline__ unsigned long long wall_ticks() { return __builtin_amdgcn_s_memrealtime(); }  // 100 MHz

template <int MODE, int DENSE, int ST = 0>
__global__ __launch_bounds__(64) void k_move(const char *x, char *y, const int ntiles, uint32_t *sink, const unsigned slot)
{
    // slot != 0: loads are issued in even slots of the wall clock only, stores in odd ones — every wave of the chip in the same phase
    auto phase = [&](unsigned want) {
        if (slot)
            while (((unsigned(wall_ticks()) / slot) & 1u) != want) __builtin_amdgcn_s_sleep(1);
    };
    const int lid = threadIdx.x;
    const size_t w = blockIdx.x, G = gridDim.x;
    // byte offset of this thread's piece of instruction j in tile t
    auto off = [&](int t, int j) -> size_t {
        if (DENSE == 2) return ((size_t(t) * kNI + size_t(j)) * G + w) * 1024 + size_t(lid) * 16;
        if (DENSE) return (size_t(t) * G + w) * kTile + size_t(j) * 1024 + size_t(lid) * 16;
        return (w * 64 + size_t(2 * j + lid / 32)) * kRow + size_t(t) * 512 + size_t(lid % 32) * 16;
    };
    u32x4 a[kNI], b[kNI], acc = {0, 0, 0, 0};
    auto load = [&](u32x4 (&r)[kNI], int t) {
        if (MODE & READ) {
#pragma unroll
            for (int j = 0; j < kNI; j++) r[j] = __builtin_nontemporal_load((const gvec *)(x + off(t, j)));
        }
    };
    auto store = [&](u32x4 (&r)[kNI], int t) {
#pragma unroll
        for (int j = 0; j < kNI; j++) {
            if (MODE & WRITE) {
                const u32x4 v = (MODE & READ) ? r[j] : u32x4{uint32_t(t), uint32_t(j), uint32_t(lid), 7u};
                if (ST == 0) {
                    __builtin_nontemporal_store(v, (gvec *)(y + off(t, j)));
                } else if (ST == 1) {
                    *(gvec *)(y + off(t, j)) = v;  // plain
                } else {
                    // four 4-byte nontemporal stores: instruction q writes the wave's 256 contiguous bytes at q * 256 of the KiB
                    typedef __attribute__((address_space(1))) uint32_t gword;
                    char *kib = y + (off(t, j) - size_t(lid) * 16 + (DENSE ? 0 : 0));
                    if (DENSE) {
#pragma unroll
                        for (int q = 0; q < 4; q++) __builtin_nontemporal_store(v[q], (gword *)(kib + q * 256 + lid * 4));
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; q++) __builtin_nontemporal_store(v[q], (gword *)(y + off(t, j) + q * 4));
                    }
                }
            } else {
                acc ^= r[j];
            }
        }
    };
    load(a, 0);
    for (int t = 0; t < ntiles; t += 2) {
        phase(0);
        if (t + 1 < ntiles) load(b, t + 1);
        phase(1);
        store(a, t);
        phase(0);
        if (t + 2 < ntiles) load(a, t + 2);
        phase(1);
        if (t + 1 < ntiles) store(b, t + 1);
    }
    if (!(MODE & WRITE) && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = acc.x;
}

template <int MODE, int DENSE, int ST = 0>
int run(const char *name, const char *x, char *y, uint32_t *sink, int waves, int ntiles, unsigned slot = 0)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_move<MODE, DENSE, ST>), dim3(waves), dim3(64), 0, 0, x, y, ntiles, sink, slot);
    CHK(hipDeviceSynchronize());
    const int iters = 20;
    CHK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k_move<MODE, DENSE, ST>), dim3(waves), dim3(64), 0, 0, x, y, ntiles, sink, slot);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double bytes = double(waves) * ntiles * kTile * ((MODE & READ ? 1 : 0) + (MODE & WRITE ? 1 : 0));
    std::printf("{\"pattern\": \"%s\", \"store\": \"%s\", \"mode\": \"%s\", \"slot_us\": %.2f, \"waves\": %d, \"ms\": %.4f, \"TB/s\": %.3f, \"GB/s per CU\": %.1f}\n", DENSE == 2 ? "front" : DENSE ? "dense" : "scattered", ST == 0 ? "nt x4" : ST == 1 ? "plain x4" : "nt x1", name, slot * 0.01, waves, ms,
                bytes / ms / 1e9, bytes / ms / 1e6 / 256.0);
    return 0;
}

int main()
{
    const int ntiles = 32;
    const size_t bytes = size_t(1024) * ntiles * kTile;  // 1 GiB
    char *x, *y;
    uint32_t *sink;
    CHK(hipMalloc(&x, bytes));
    CHK(hipMalloc(&y, bytes));
    CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(x, 1, bytes));
    CHK(hipMemset(y, 0, bytes));
    for (int waves : {768, 896, 1024}) {
        run<READ, 0>("read", x, y, sink, waves, ntiles);
        run<WRITE, 0>("write", x, y, sink, waves, ntiles);
        run<READ | WRITE, 0>("copy", x, y, sink, waves, ntiles);
        run<READ, 1>("read", x, y, sink, waves, ntiles);
        run<WRITE, 1>("write", x, y, sink, waves, ntiles);
        run<READ | WRITE, 1>("copy", x, y, sink, waves, ntiles);
    }
    // store forms and the dense FRONT (1 KiB pieces of all waves side by side, as the FrameMajor sweep's row segments)
    run<WRITE, 1, 1>("write", x, y, sink, 1024, ntiles);
    run<WRITE, 1, 2>("write", x, y, sink, 1024, ntiles);
    run<READ | WRITE, 1, 1>("copy", x, y, sink, 1024, ntiles);
    run<READ, 2>("read", x, y, sink, 1024, ntiles);
    run<WRITE, 2>("write", x, y, sink, 1024, ntiles);
    run<WRITE, 2, 1>("write", x, y, sink, 1024, ntiles);
    run<WRITE, 2, 2>("write", x, y, sink, 1024, ntiles);
    run<READ | WRITE, 2>("copy", x, y, sink, 1024, ntiles);
    run<READ | WRITE, 2, 1>("copy", x, y, sink, 1024, ntiles);
    run<WRITE, 0, 1>("write", x, y, sink, 1024, ntiles);
    run<READ | WRITE, 0, 1>("copy", x, y, sink, 1024, ntiles);
    // chip-wide read / write phases by the wall clock
    for (unsigned slot : {25u, 50u, 100u, 200u, 400u, 800u}) {
        run<READ | WRITE, 0>("copy", x, y, sink, 1024, ntiles, slot);
        run<READ | WRITE, 1>("copy", x, y, sink, 1024, ntiles, slot);
    }
    return 0;
}
