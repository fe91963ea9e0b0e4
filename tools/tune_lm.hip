// tune_lm.hip — LaneMajor: the staged 16-byte-piece kernel (stream_lane_major_staged<P, LW, LB>: 512-byte runs per lane, 64 / 32 /
// 16 lanes per wave) against the 4-byte tile kernel (stream_lane_major<P>) per processor, lane count and row pitch; every
// combination is first compared bit for bit (outputs and states) with the tile kernel on random input, out of place and in
// place.  Also carries the LDS-DMA experiments the staged kernel came from (exp_lane_major_lds<P, NB, LB, NTL, NTS, PF>: run
// length LB, ring depth NB, nontemporal on / off, touch-prefetch PF), kept here as the record.  One JSON line per combination.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fwrapv -fno-slp-vectorize -Wno-pass-failed -Iinclude \
//         -Iidsp_amd/csrc tools/tune_lm.hip -o build/tune_lm
//   build/tune_lm [processor index, -1 = all] [1 = also 1 KiB runs at 32 / 16 lanes per wave, 2 = also the LDS-DMA experiments]
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "biquad_sections.h"

namespace idsp {
char *last_error_buf() { static thread_local char b[512]; return b; }
int fail(int code, const char *, ...) { return code; }
void note_kernel(const char *, const char *) {}
}  // namespace idsp

// The LDS-DMA twin of stream_lane_major_staged that was measured and not kept (profiles/NOTES.md section 3): ring of NB slots of 64 lanes x LB
// bytes filled by global_load_lds_dwordx4 (hand-counted vmcnt), optional touch-prefetch of PF-byte chunks (PF).
namespace idsp {
template <bool NT>
__device__ __forceinline__ void exp_glds16(const void *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    if constexpr (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
template <class P, int NB, int LB, bool NTL = true, bool NTS = true, int PF = 0>
__global__ __launch_bounds__(kWave) void exp_lane_major_lds(
    const typename P::Params prm, uint32_t *st, const typename P::In *x, typename P::Out *y,
    const size_t lanes, const size_t frames, const size_t xl, const size_t yl)
{
    using In = typename P::In;
    using Out = typename P::Out;
    static_assert(P::HAS_IN && P::IN_DIV == 1 && sizeof(In) == sizeof(Out) && (sizeof(In) == 4 || sizeof(In) == 8),
                  "LaneMajor LDS path: one input and one output of the same 4- or 8-byte size per lane and frame");
    static_assert(LB == 128 || LB == 256 || LB == 512 || LB == 1024, "bytes per lane and tile");
    constexpr int W = sizeof(In) / 4;        // words per sample
    constexpr int TF = LB / 4 / W;           // frames per tile
    constexpr int PCS = LB / 16;             // 16-byte pieces per lane and tile = DMA instructions per tile
    constexpr int G = kWave / PCS;           // lanes per DMA instruction
    constexpr int SWM = (PCS < 16 ? PCS : 16) - 1;  // swizzle mask
    constexpr int SPP = 4 / W;               // samples per piece
    constexpr int kSlot = kWave * LB;        // bytes per ring slot
    // EXPERIMENT (PF > 0): every tile also touches LB / 128 x 64 lines (one dword each, into a scratch LDS row) of the
    // chunk of PF bytes per lane two chunks ahead, so that whole PF-byte runs of every lane are requested together
    constexpr int TPT = PF ? LB / 128 : 0;   // touch instructions per tile
    constexpr int PFT = PF ? PF / LB : 1;    // tiles per chunk
    constexpr int LPC = PF ? PF / 128 : 1;   // lines per lane and chunk = touch instructions per chunk
    constexpr int kYoungAll = (NB - 1) * 2 * PCS + NB * TPT;
    constexpr int kYoung = kYoungAll < 63 ? kYoungAll : 63;  // the counter has 6 bits: waiting for more is always safe
    static_assert(NB >= 1, "ring");
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *ptab = smem + NB * kSlot / 4 + (PF ? 64 : 0);  // [P::LDS_WORDS]
    const int lid = threadIdx.x;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)smem;
    const uint32_t lds_scratch = lds_base + NB * kSlot;

    const size_t lane0 = size_t(blockIdx.x) * kWave;
    const size_t nrows = lanes - lane0 < size_t(kWave) ? lanes - lane0 : size_t(kWave);
    const bool active = size_t(lid) < nrows;

    P p;
    if constexpr (P::LDS_WORDS > 0) {
        P::fill_shared(ptab, lid, kWave);
        lds_wave_sync();
        p.set_shared(ptab);
    }
    if (active) p.load(prm, st, lanes, lane0 + lid);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // state loads landed, visibly to the compiler (see stream_frame_major_lds)

    // mover role of this thread: in instruction j, lane mq + j of the tile, piece mpc ^ (j & SWM) of its run
    const int mq = (lid / PCS) * PCS, mpc = lid % PCS;
    const uint32_t *xq = reinterpret_cast<const uint32_t *>(x) + (lane0 + mq) * xl * W;
    uint32_t *yq = reinterpret_cast<uint32_t *>(y) + (lane0 + mq) * yl * W;
    const size_t xrow = xl * W, yrow = yl * W;  // words between lanes
    // owner role: slot row of this thread's lane, and the byte offset of its piece k = own ^ (16 k)
    const uint32_t own = uint32_t((lid % PCS) * G + lid / PCS) * LB + uint32_t(lid & SWM) * 16;

    const size_t nfull = frames / TF;
    auto issue = [&](size_t v, int slot) __attribute__((always_inline)) {
        const uint32_t *src = xq + v * (TF * W);
#pragma unroll
        for (int j = 0; j < PCS; j++)
            if (size_t(mq + j) < nrows) exp_glds16<NTL>(src + j * xrow + ((mpc ^ (j & SWM)) * 4), lds_base + uint32_t(slot * kSlot + j * 1024));
    };
    auto compute = [&](int slot) __attribute__((always_inline)) {
        char *base = reinterpret_cast<char *>(smem) + slot * kSlot;
        auto piece = [&](int k) __attribute__((always_inline)) {
            u32x4 *q = reinterpret_cast<u32x4 *>(base + (own ^ uint32_t(k * 16)));
            u32x4 v = *q;
#pragma unroll
            for (int s = 0; s < SPP; s++) {
                uint32_t w[W];
#pragma unroll
                for (int h = 0; h < W; h++) w[h] = v[s * W + h];
                const Out o = step1(p, prm, words_to<In>(w));
                to_words<Out>(o, w);
#pragma unroll
                for (int h = 0; h < W; h++) v[s * W + h] = w[h];
            }
            *q = v;
        };
        if (active) {
            if constexpr (MaxU<P>::value < 24) {
                for (int k = 0; k < PCS; k++) piece(k);  // large body: keep the loop rolled
            } else {
#pragma unroll
                for (int k = 0; k < PCS; k++) piece(k);
            }
        }
    };
    auto store = [&](size_t v, int slot) __attribute__((always_inline)) {
        const char *base = reinterpret_cast<const char *>(smem) + slot * kSlot + lid * 16;
        uint32_t *dst = yq + v * (TF * W);
#pragma unroll
        for (int j = 0; j < PCS; j++) {
            const u32x4 v4 = *reinterpret_cast<const u32x4 *>(base + j * 1024);
            if (size_t(mq + j) < nrows) {
                u32x4 *d = reinterpret_cast<u32x4 *>(dst + j * yrow + ((mpc ^ (j & SWM)) * 4));
                if constexpr (NTS)
                    __builtin_nontemporal_store(v4, d);
                else
                    *d = v4;
            }
        }
    };

    const size_t row_words = frames * W;
    auto touch = [&](size_t v) __attribute__((always_inline)) {
        if constexpr (PF > 0) {
            const size_t chunk = v / PFT + 2;
            const int k0 = int(v % PFT) * TPT;
#pragma unroll
            for (int u = 0; u < TPT; u++) {
                const int k = k0 + u;
                size_t lane = size_t(k) * (kWave / LPC) + lid / LPC;
                lane = lane < nrows ? lane : nrows - 1;
                size_t w = chunk * (PF / 4) + size_t(lid % LPC) * 32;
                w = w < row_words ? w : row_words - 1;
                const uint32_t *src = reinterpret_cast<const uint32_t *>(x) + (lane0 + lane) * xrow + w;
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_scratch))
                             : "memory");
            }
        }
    };
    for (size_t t = 0; t < size_t(NB) && t < nfull; t++) issue(t, int(t));
    size_t i = 0;
    int slot = 0;
    auto tile = [&](auto steady) __attribute__((always_inline)) {
        if constexpr (decltype(steady)::value)
            wait_vmcnt<kYoung>();
        else
            wait_vmcnt<0>();
        compute(slot);
        lds_wave_sync();
        store(i, slot);
        lds_wave_sync();  // the slot has been read: re-arm it
        if (i + NB < nfull) issue(i + NB, slot);
        touch(i);
        slot = slot + 1 == NB ? 0 : slot + 1;
    };
    for (; i < nfull && i < size_t(NB); i++) tile(std::false_type{});
    if (nrows == size_t(kWave))  // whole wave: every past tile issued exactly PCS DMA rows and PCS stores
        for (; i + NB < nfull; i++) tile(std::true_type{});
    for (; i < nfull; i++) tile(std::false_type{});
    wait_vmcnt<0>();

    if (active) {  // frames % TF: this lane's own row, sample by sample
        const In *xr = x + (lane0 + lid) * xl;
        Out *yr = y + (lane0 + lid) * yl;
        for (size_t f = nfull * TF; f < frames; f++) yr[f] = step1(p, prm, xr[f]);
        p.store(prm, st, lanes, lane0 + lid);
    }
}

}  // namespace idsp

using namespace idsp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape {
    size_t lanes, frames, pitch;
};

template <class F>
float timeit(F &&launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int i = 0; i < 40; i++) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 20) ts.push_back(ms);
    }
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

struct Bufs {
    char *x, *y, *yref;
    uint32_t *st, *stref;
    size_t cap;
};

template <class P, int NB, int LB, bool NTL = true, bool NTS = true, int PF = 0>
void one(const char *name, const typename P::Params &prm, const Bufs &b, const Shape &sh, float tref)
{
    using In = typename P::In;
    using Out = typename P::Out;
    const size_t bytes = size_t(NB) * kWave * LB + P::LDS_WORDS * 4 + 256;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(exp_lane_major_lds<P, NB, LB, NTL, NTS, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    const unsigned grid = unsigned((sh.lanes + kWave - 1) / kWave);
    auto launch = [&]() {
        hipLaunchKernelGGL((exp_lane_major_lds<P, NB, LB, NTL, NTS, PF>), dim3(grid), dim3(kWave), bytes, 0, prm, b.st, reinterpret_cast<const In *>(b.x),
                           reinterpret_cast<Out *>(b.y), sh.lanes, sh.frames, sh.pitch, sh.pitch);
    };
    // parity with the tile kernel: same input, zero state
    const size_t n = sh.lanes * sh.pitch * sizeof(In);
    CK(hipMemset(b.st, 0, sh.lanes * 256));
    CK(hipMemset(b.y, 0xEE, n));
    launch();
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> got(n / 4), want(n / 4), sg(sh.lanes * 64), sw(sh.lanes * 64);
    CK(hipMemcpy(got.data(), b.y, n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(want.data(), b.yref, n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sg.data(), b.st, sh.lanes * 256, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sw.data(), b.stref, sh.lanes * 256, hipMemcpyDeviceToHost));
    const bool ok = got == want && sg == sw;
    const float t = timeit(launch);
    const double gb = double(sh.lanes) * sh.frames * (sizeof(In) + sizeof(Out)) / 1e9;
    printf("{\"proc\": \"%s\", \"lanes\": %zu, \"frames\": %zu, \"pitch\": %zu, \"nb\": %d, \"lb\": %d, \"nt\": \"%d%d pf%d\", \"ok\": %s, \"ms\": %.4f, \"frac\": %.3f, \"tile_ms\": %.4f, \"tile_frac\": %.3f}\n",
           name, sh.lanes, sh.frames, sh.pitch, NB, LB, int(NTL), int(NTS), PF, ok ? "true" : "false", t, gb / (t * 1e-3) / 8000, tref, gb / (tref * 1e-3) / 8000);
    fflush(stdout);
}

template <class P, int LW = 64, int LB = kLmRun>
void one_staged(const char *name, const typename P::Params &prm, const Bufs &b, const Shape &sh, float tref, bool inplace = false)
{
    using In = typename P::In;
    using Out = typename P::Out;
    const size_t bytes = lm_staged_lds_bytes<P, LW, LB>();
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stream_lane_major_staged<P, LW, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    const unsigned grid = unsigned((sh.lanes + LW - 1) / LW);
    const size_t n = sh.lanes * sh.pitch * sizeof(In);
    char *yy = b.y;
    auto launch = [&]() {
        hipLaunchKernelGGL((stream_lane_major_staged<P, LW, LB>), dim3(grid), dim3(kWave), bytes, 0, prm, b.st, reinterpret_cast<const In *>(inplace ? yy : b.x),
                           reinterpret_cast<Out *>(yy), sh.lanes, sh.frames, sh.pitch, sh.pitch, 0u, 8u);
    };
    CK(hipMemset(b.st, 0, sh.lanes * 256));
    if (inplace)
        CK(hipMemcpy(yy, b.x, n, hipMemcpyDeviceToDevice));
    else
        CK(hipMemset(yy, 0xEE, n));
    launch();
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> got(n / 4), want(n / 4), sg(sh.lanes * 64), sw(sh.lanes * 64);
    CK(hipMemcpy(got.data(), yy, n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(want.data(), b.yref, n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sg.data(), b.st, sh.lanes * 256, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sw.data(), b.stref, sh.lanes * 256, hipMemcpyDeviceToHost));
    bool ok = sg == sw;
    // the pitch gap (pitch > frames) is never written by either kernel: compare the frames only
    for (size_t l = 0; l < sh.lanes && ok; l++)
        ok = memcmp(&got[l * sh.pitch * sizeof(In) / 4], &want[l * sh.pitch * sizeof(In) / 4], sh.frames * sizeof(In)) == 0;
    const float t = timeit(launch);
    const double gb = double(sh.lanes) * sh.frames * (sizeof(In) + sizeof(Out)) / 1e9;
    printf("{\"proc\": \"%s\", \"lanes\": %zu, \"frames\": %zu, \"pitch\": %zu, \"nb\": %d, \"lb\": %d, \"nt\": \"staged%s\", \"ok\": %s, \"ms\": %.4f, \"frac\": %.3f, \"tile_ms\": %.4f, \"tile_frac\": %.3f}\n",
           name, sh.lanes, sh.frames, sh.pitch, LW, LB, inplace ? " in place" : "", ok ? "true" : "false", t, gb / (t * 1e-3) / 8000, tref, gb / (tref * 1e-3) / 8000);
    fflush(stdout);
}

static bool exp_too = false;
static int exp_level = 0;
template <class P>
void sweep(const char *name, const typename P::Params &prm, const Bufs &b, const std::vector<Shape> &shapes)
{
    using In = typename P::In;
    using Out = typename P::Out;
    for (const Shape &sh : shapes) {
        if (sh.lanes * sh.pitch * sizeof(In) > b.cap) continue;
        const unsigned grid = unsigned((sh.lanes + kWave - 1) / kWave);
        auto ref = [&]() {
            hipLaunchKernelGGL((stream_lane_major<P>), dim3(grid), dim3(kWave), 0, 0, prm, b.stref, reinterpret_cast<const In *>(b.x),
                               reinterpret_cast<Out *>(b.yref), sh.lanes, sh.frames, sh.pitch, sh.pitch);
        };
        const float tref = timeit(ref);
        CK(hipMemset(b.stref, 0, sh.lanes * 256));
        CK(hipMemset(b.yref, 0xEE, sh.lanes * sh.pitch * sizeof(In)));
        ref();
        CK(hipDeviceSynchronize());
        one_staged<P, 64>(name, prm, b, sh, tref);
        one_staged<P, 64>(name, prm, b, sh, tref, true);
        one_staged<P, 32>(name, prm, b, sh, tref);
        one_staged<P, 32>(name, prm, b, sh, tref, true);
        one_staged<P, 16>(name, prm, b, sh, tref);
        one_staged<P, 16>(name, prm, b, sh, tref, true);
        if (exp_too) {
            one_staged<P, 32, 1024>(name, prm, b, sh, tref);
            one_staged<P, 32, 1024>(name, prm, b, sh, tref, true);
            one_staged<P, 16, 1024>(name, prm, b, sh, tref);
        }
        if (exp_level >= 2 && sizeof(In) == 4) {
            one<P, 1, 512>(name, prm, b, sh, tref);
            one<P, 2, 256>(name, prm, b, sh, tref);
            one<P, 4, 128>(name, prm, b, sh, tref);
        }
    }
}

template <class SecP, int N>
bq::ChainParams<SecP, N> params_n()
{
    bq::ChainParams<SecP, N> p{};
    for (int k = 0; k < N; k++) {
        if constexpr (std::is_same<SecP, bq::SecI32>::value) {
            p.sec[k] = {{1 << 20, 1 << 21, 1 << 20, 1 << 30, -(1 << 29)}, 30, 3, -(1 << 30), 1 << 30};
        } else if constexpr (std::is_same<SecP, bq::SecF32>::value) {
            p.sec[k] = {{0.001f, 0.002f, 0.001f, 1.9f, -0.91f}, 0.01f, -10.f, 10.f};
        } else {
            p.sec[k] = {{0.001, 0.002, 0.001, 1.9, -0.91}, 0.01, -10., 10.};
        }
    }
    return p;
}

int main(int argc, char **argv)
{
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    exp_level = argc > 2 ? atoi(argv[2]) : 0;
    exp_too = exp_level != 0;
    Bufs b;
    b.cap = size_t(3) << 30;
    CK(hipMalloc(&b.x, b.cap));
    CK(hipMalloc(&b.y, b.cap));
    CK(hipMalloc(&b.yref, b.cap));
    CK(hipMalloc(&b.st, size_t(1) << 28));
    CK(hipMalloc(&b.stref, size_t(1) << 28));
    {
        // small random integers: valid as i32 samples, and as f32 / f64 bit patterns they are denormals (finite)
        std::vector<uint32_t> h(size_t(64) << 20);
        std::mt19937 g(1);
        for (auto &v : h) v = g() >> 12;
        for (size_t o = 0; o < b.cap; o += h.size() * 4) CK(hipMemcpy(b.x + o, h.data(), std::min(h.size() * 4, b.cap - o), hipMemcpyHostToDevice));
    }
    const std::vector<Shape> shapes = {
        {65536, 4096, 4096}, {65536, 4096, 4128}, {16384, 4096, 4096}, {32768, 4096, 4096}, {131072, 4096, 4096}, {4096, 16384, 16384},
        {65536, 4099, 4100}, {65500, 1000, 1000}, {1000, 77, 80}, {1000, 131, 132}, {777, 3, 4}, {64, 1025, 1028},
    };
    int k = 0;
#define SWEEPN(SEC, N) if (only < 0 || only == k) sweep<bq::Chain<bq::SEC, N>>(#SEC " x" #N, params_n<bq::SEC::Sec, N>(), b, shapes); k++;
    SWEEPN(Df1I32<false>, 1)
    SWEEPN(Df1I32<true>, 1)
    SWEEPN(WideI32<false>, 1)
    SWEEPN(Df2tF32<false>, 1)
    SWEEPN(Df1F32<true>, 1)
    SWEEPN(Df1I32<false>, 2)
    SWEEPN(Df1I32<false>, 4)
    SWEEPN(Df2tF32<false>, 4)
    SWEEPN(Df1F64<false>, 1)
    SWEEPN(Df2tF64<true>, 2)
#define SWEEPC(T, SECP, N) if (only < 0 || only == k) sweep<bq::CascadeDf1<T, N>>("CascadeDf1<" #T "> x" #N, params_n<bq::SECP, N>(), b, shapes); k++;
    SWEEPC(int32_t, SecI32, 8)
    SWEEPC(double, SecF64, 4)
    return 0;
}
