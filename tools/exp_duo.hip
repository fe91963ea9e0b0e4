// exp_duo.hip — round-3 experiment: a serial chain of i32 sections on one wave per 64 lanes (the register-window kernel) against
// the same chain split over two waves that hand the samples over through LDS (stream_frame_major_duo), at the C2 shape.
// Compares the outputs bit for bit first.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fwrapv -ffp-contract=off -fno-slp-vectorize -w -Iinclude -Iidsp_amd/csrc tools/exp_duo.hip -o build/exp_duo
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "biquad_sections.h"

namespace idsp {
char *last_error_buf() { static thread_local char b[512]; return b; }
int fail(int code, const char *, ...) { return code; }
void note_kernel(const char *, const char *) {}
}  // namespace idsp
using namespace idsp;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class F>
float timeit(F f)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::vector<float> ts;
    for (int i = 0; i < 30; i++) {
        CK(hipEventRecord(a));
        f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (i >= 15) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

template <class PW, class PA, class PB, int NA, class Sec>
void run(const char *name, size_t lanes, size_t frames, int words_a)
{
    typename PW::Params pw{};
    constexpr int N = sizeof(pw.sec) / sizeof(pw.sec[0]);
    for (int k = 0; k < N; k++) {
        pw.sec[k] = Sec{};
        const int32_t ba[5] = {1 << 26, 1 << 27, 1 << 26, (1 << 29) + 1000 * k, -(1 << 28)};
        std::memcpy(pw.sec[k].ba, ba, sizeof(ba));
        pw.sec[k].frac = 30, pw.sec[k].mn = INT32_MIN, pw.sec[k].mx = INT32_MAX;
    }
    typename PA::Params pa{};
    typename PB::Params pb{};
    for (int k = 0; k < NA; k++) pa.sec[k] = pw.sec[k];
    for (int k = NA; k < N; k++) pb.sec[k - NA] = pw.sec[k];
    int32_t *x, *y1, *y2;
    uint32_t *st1, *st2;
    const size_t n = lanes * frames, sw = 64;
    CK(hipMalloc(&x, n * 4));
    CK(hipMalloc(&y1, n * 4));
    CK(hipMalloc(&y2, n * 4));
    CK(hipMalloc(&st1, sw * lanes * 4));
    CK(hipMalloc(&st2, sw * lanes * 4));
    std::vector<int32_t> hx(n);
    for (size_t i = 0; i < n; i++) hx[i] = int32_t((i * 2654435761u) >> 7) - (1 << 24);
    CK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(st1, 0, sw * lanes * 4));
    CK(hipMemset(st2, 0, sw * lanes * 4));
    auto whole = [&]() { hipLaunchKernelGGL((stream_frame_major<PW, 24>), dim3(unsigned(lanes / 256)), dim3(256), 0, 0, pw, st1, x, y1, lanes, frames, lanes, lanes, 0); };
    auto duo = [&]() {
        hipLaunchKernelGGL((stream_frame_major_duo<PA, PB>), dim3(unsigned((lanes + 63) / 64)), dim3(128), 0, 0, pa, pb, st2, st2 + size_t(words_a) * lanes, x, y2, lanes, frames,
                           lanes, lanes);
    };
    whole();
    duo();
    CK(hipDeviceSynchronize());
    std::vector<int32_t> a(n), b(n);
    std::vector<uint32_t> sa(sw * lanes), sb(sw * lanes);
    CK(hipMemcpy(a.data(), y1, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), y2, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sa.data(), st1, sw * lanes * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sb.data(), st2, sw * lanes * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, sbad = 0;
    for (size_t i = 0; i < n; i++) bad += a[i] != b[i];
    for (size_t i = 0; i < sa.size(); i++) sbad += sa[i] != sb[i];
    const float tw = timeit(whole), td = timeit(duo);
    const double gb = double(n) * 8 / 1e9;
    printf("{\"case\": \"%s\", \"lanes\": %zu, \"frames\": %zu, \"output_mismatches\": %zu, \"state_mismatches\": %zu, \"one_wave_ms\": %.4f, \"two_waves_ms\": %.4f, \"one_wave_frac\": %.3f, "
           "\"two_waves_frac\": %.3f}\n",
           name, lanes, frames, bad, sbad, tw, td, gb / (tw * 1e-3) / 8000, gb / (td * 1e-3) / 8000);
    fflush(stdout);
    hipFree(x), hipFree(y1), hipFree(y2), hipFree(st1), hipFree(st2);
}

int main()
{
    using S = bq::Df1I32<false>;
    for (size_t lanes : {size_t(65536), size_t(32768), size_t(131072)}) {
        run<bq::Chain<S, 4>, bq::Chain<S, 2>, bq::Chain<S, 2>, 2, bq::SecI32>("Chain<Df1I32, 4> = 2 + 2", lanes, 4096, 2 * 4);
        run<bq::CascadeDf1<int32_t, 8>, bq::CascadeDf1<int32_t, 4>, bq::CascadeDf1<int32_t, 4>, 4, bq::SecI32>("CascadeDf1<i32, 8> = 4 + 4", lanes, 4096, 2 * 4);
        run<bq::CascadeDf1<int32_t, 4>, bq::CascadeDf1<int32_t, 2>, bq::CascadeDf1<int32_t, 2>, 2, bq::SecI32>("CascadeDf1<i32, 4> = 2 + 2", lanes, 4096, 2 * 2);
    }
    return 0;
}
