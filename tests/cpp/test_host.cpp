// test_host.cpp — the reference's own tests for the hot path, written against the C++ host
// layer (include/idsp_hip.hpp) the way they read in Rust.  Runs on the GPU box:
//   g++ -std=c++17 -Iinclude tests/cpp/test_host.cpp -Lidsp_amd/lib -lidsp_hip -o build/test_host
#include <cmath>
#include <cstdio>
#include <vector>

#include "idsp_hip.hpp"

using namespace idsp_hip;

static int failures = 0;
#define EXPECT(cond)                                                   \
    do {                                                               \
        if (!(cond)) {                                                 \
            std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            failures++;                                                \
        }                                                              \
    } while (0)

// coefficients::Filter::default().critical_frequency(f0).gain(g).{lowpass,highpass}() in f64
// (src/iir/coefficients.rs:259-335)
static std::array<double, 6> filter(double f0, double gain, bool highpass)
{
    const double w0 = 2.0 * M_PI * f0, fsin = std::sin(w0), fcos = std::cos(w0);
    const double alpha = 0.5 * fsin * std::sqrt(2.0);
    const double b = gain * 0.5 * (highpass ? 1.0 + fcos : 1.0 - fcos);
    return {b, (highpass ? -2.0 : 2.0) * b, b, 1.0 + alpha, -2.0 * fcos, 1.0 - alpha};
}

int main()
{
    // src/iir/coefficients.rs:289-300 and :316-326
    for (int hp = 0; hp < 2; hp++) {
        auto iir = Biquad<Q32<30>>::from_sos(filter(0.1, 1000.0, hp));
        DeviceBuffer<int32_t> xy(std::vector<int32_t>{3, -4, 5, 7, -3, 2});
        auto p = Split(iir, DirectForm1{}).lanes(1);
        p.inplace(xy);
        const std::vector<int32_t> want = hp ? std::vector<int32_t>{5, -9, 11, 12, -1, 17} : std::vector<int32_t>{5, 3, 9, 25, 42, 49};
        EXPECT(xy.to_host() == want);
    }
    // src/iir/biquad.rs:176-214: IDENTITY, proportional, HOLD
    {
        DeviceBuffer<float> x(std::vector<float>{3.0f}), y(1);
        Split(Biquad<float>::identity(), DirectForm1{}).lanes(1).block(x, y);
        EXPECT(y.to_host()[0] == 3.0f);
        Split(Biquad<float>::proportional(2.0f), DirectForm1{}).lanes(1).block(x, y);
        EXPECT(y.to_host()[0] == 6.0f);
        auto hold = Split(Biquad<float>::hold(), DirectForm1{}).lanes(1);
        const float two = 2.0f;
        uint32_t bits;
        std::memcpy(&bits, &two, 4);
        hold.state().upload({0u, 0u, bits, bits});  // state.set_y(2.0)
        DeviceBuffer<float> x7(std::vector<float>{7.0f});
        hold.block(x7, y);
        EXPECT(y.to_host()[0] == 2.0f);
        EXPECT(Biquad<float>::proportional(3.0f).forward_gain() == 3.0f);
    }
    // src/iir/biquad.rs:127-157: BiquadClamp u / min / max; :385-393, :409-417 DF2T identity
    {
        DeviceBuffer<float> x(std::vector<float>{0.0f}), y(1);
        BiquadClamp<float> i;
        i.u = 5.0f;
        Split(i, DirectForm1{}).lanes(1).block(x, y);
        EXPECT(y.to_host()[0] == 5.0f);
        i = BiquadClamp<float>();
        i.min = 5.0f;
        Split(i, DirectForm1{}).lanes(1).block(x, y);
        EXPECT(y.to_host()[0] == 5.0f);
        i = BiquadClamp<float>();
        i.max = -5.0f;
        Split(i, DirectForm1{}).lanes(1).block(x, y);
        EXPECT(y.to_host()[0] == -5.0f);
        DeviceBuffer<float> x3(std::vector<float>{3.0f});
        Split(BiquadClamp<float>(Biquad<float>::identity()), DirectForm2Transposed{}).lanes(1).block(x3, y);
        EXPECT(y.to_host()[0] == 3.0f);
        Split(Biquad<float>::identity(), DirectForm2Transposed{}).lanes(1).block(x3, y);
        EXPECT(y.to_host()[0] == 3.0f);
    }
    // dsp-process/src/lib.rs:146-154 (LaneMajor lanes) and :136-144 (FrameMajor fallback), offset stage
    {
        BiquadClamp<float> off(Biquad<float>::identity());
        off.u = 3.0f;
        auto p = Split(off, DirectForm1{}).lanes(2);
        DeviceBuffer<float> x(std::vector<float>{1, 2, 3, 10, 20, 30}), y(6);
        p.process_view(View<float, LaneMajor>::from_flat(x, 2), ViewMut<float, LaneMajor>::from_flat(y, 2));
        EXPECT((y.to_host() == std::vector<float>{4, 5, 6, 13, 23, 33}));
        auto q = Split(off, DirectForm1{}).lanes(2);
        DeviceBuffer<float> xf(std::vector<float>{1, 2, 3, 4}), yf(4);
        q.process_view(View<float, FrameMajor>::from_flat(xf, 2), ViewMut<float, FrameMajor>::from_flat(yf, 2));
        EXPECT((yf.to_host() == std::vector<float>{4, 5, 6, 7}));
        bool threw = false;
        try {
            DeviceBuffer<float> bad(5);
            View<float, LaneMajor>::from_flat(bad, 2);  // view.rs:182 assert_eq!
        } catch (const Error &) {
            threw = true;
        }
        EXPECT(threw);
    }
    // src/iir/biquad.rs:493-510: DirectForm1Dither doctest
    {
        auto p = Split(Biquad<Q32<30>>::identity(), DirectForm1Dither{}).lanes(1);
        p.state().upload({1u, 2u, 3u, 4u, 5u});
        DeviceBuffer<int32_t> x(std::vector<int32_t>{6}), y(1);
        p.block(x, y);
        EXPECT(y.to_host()[0] == 6);
        EXPECT((p.state().to_host() == std::vector<uint32_t>{6u, 1u, 6u, 3u, 5u}));
    }
    // src/hbf.rs:577-595 response length of the /16 cascade; DC gain 2 per stage (hbf.rs:548-555)
    {
        HbfDecCascade h(4, 3);
        EXPECT(h.response_length() == 57);
        EXPECT(h.state().len() == 118 * 3);
        const size_t frames = 200;
        DeviceBuffer<float> x(std::vector<float>(3 * frames * 16, 1.0f)), y(3 * frames);
        h.process_view(View<float, LaneMajor>::from_flat(x, 3, 16), ViewMut<float, LaneMajor>::from_flat(y, 3));
        EXPECT(std::fabs(y.to_host().back() - 16.0f) < 1e-4f);
        HbfIntCascade g(4, 2);
        EXPECT(g.response_length() == 922);
    }
    // src/cossin.rs:14-67 at phase 0 and the py.rs:10-28 shape
    {
        DeviceBuffer<int32_t> p(std::vector<int32_t>{0, 1 << 30}), out(4);
        cossin(p, out);
        auto o = out.to_host();
        EXPECT(o[0] == 2147454703 && o[1] == -1898);
        EXPECT(std::abs(o[2]) < (1 << 17) && o[3] > 2147400000);
    }
    // src/atan2.rs:177-183 (zero axes are exact), rows are [x, y]
    {
        DeviceBuffer<int32_t> xy(std::vector<int32_t>{1, 0, 0, 1, INT32_MAX, 0, 0, INT32_MAX}), out(4);
        atan2(xy, out);
        auto o = out.to_host();
        EXPECT(o[0] == 0 && o[1] == 0x3fffffff && o[2] == 0 && o[3] == 0x3fffffff);
    }
    // Lockin<[Lowpass<2>; 2]> + Accu: a tone at the LO frequency demodulates to DC (examples/ddc_lockin.rs shape)
    {
        const size_t n = 8192, lanes = 2;
        const double f = 0.173, phi = 0.37, amp = double(1 << 28);
        const int32_t step = int32_t(uint32_t(std::llround(f * 4294967296.0)));
        std::vector<int32_t> xs(n * lanes);
        uint32_t ph = 0;
        for (size_t i = 0; i < n; i++) {
            ph += uint32_t(step);
            const int32_t v = int32_t(std::lround(amp * std::cos(2.0 * M_PI * double(ph) / 4294967296.0 - phi)));
            xs[i * lanes] = xs[i * lanes + 1] = v;
        }
        const double k = M_PI * 2147483648.0 * 0.004;
        const std::array<int32_t, 2> lp{int32_t(k * k / 4294967296.0), -int32_t(k * std::sqrt(2.0))};
        Lockin<2, 2> li({lp, lp}, {0, 0}, {step, step});
        DeviceBuffer<int32_t> x(xs), y(2 * n * lanes);
        li.process_view(View<int32_t, FrameMajor>::from_flat(x, lanes), ViewMut<int32_t, FrameMajor>::from_flat(y, lanes, 2));
        auto o = y.to_host();
        double si = 0, sq = 0;
        for (size_t i = 6144; i < n; i++) si += o[(i * lanes) * 2], sq += o[(i * lanes) * 2 + 1];
        si /= (n - 6144) * amp, sq /= (n - 6144) * amp;
        EXPECT(std::fabs(si - 0.25 * std::cos(phi)) < 3e-3 && std::fabs(sq - 0.25 * std::sin(phi)) < 3e-3);
        // the same pass with the polar read-out fused in: arg == atan2 of the Complex<i32> output, norm_sqr == re^2 + im^2
        Lockin<2, 2> la({lp, lp}, {0, 0}, {step, step}), lq({lp, lp}, {0, 0}, {step, step});
        DeviceBuffer<int32_t> ya(n * lanes), want(n * lanes);
        DeviceBuffer<int64_t> yp(n * lanes);
        la.process_view_arg(View<int32_t, FrameMajor>::from_flat(x, lanes), ViewMut<int32_t, FrameMajor>::from_flat(ya, lanes));
        lq.process_view_norm_sqr(View<int32_t, FrameMajor>::from_flat(x, lanes), ViewMut<int64_t, FrameMajor>::from_flat(yp, lanes));
        atan2(y, want);
        EXPECT(ya.to_host() == want.to_host());
        auto p = yp.to_host();
        bool ok = true;
        for (size_t i = 0; i < n * lanes; i++) ok = ok && p[i] == int64_t(o[2 * i]) * o[2 * i] + int64_t(o[2 * i + 1]) * o[2 * i + 1];
        EXPECT(ok);
    }
    // `Biquad<f64>` DF1 vs DF2T (shape of src/iir/biquad.rs:672-682) and a same-rate EvenSymmetric FIR
    {
        auto b = Biquad<double>::from_sos({0.7, -0.4, 0.1, 1.0, -0.2, 0.05});
        DeviceBuffer<double> x(std::vector<double>{-1.0, 0.25, 0.75, -0.5, 0.125, 0.0, 0.5, -0.25}), y1(8), y2(8);
        Split(b, DirectForm1{}).lanes(1).block(x, y1);
        Split(b, DirectForm2Transposed{}).lanes(1).block(x, y2);
        auto a = y1.to_host(), c = y2.to_host();
        for (int i = 0; i < 8; i++) EXPECT(std::fabs(a[i] - c[i]) < 1e-12);
        FirSym fir(IDSP_FIR_EVEN_SYMMETRIC, {0.25f, 0.5f}, 1);
        DeviceBuffer<float> xi(std::vector<float>{1, 0, 0, 0, 0, 0}), yi(6);
        fir.process_view(View<float, LaneMajor>::from_flat(xi, 1), ViewMut<float, LaneMajor>::from_flat(yi, 1));
        EXPECT((yi.to_host() == std::vector<float>{0.25f, 0.5f, 0.5f, 0.25f, 0.0f, 0.0f}));
    }
    // `ByLane<[Biquad<Q32<30>>; 3]>` (dsp-process/src/compose.rs:363-390): lane i == filter i run alone
    {
        std::vector<Biquad<Q32<30>>> bank;
        for (double f0 : {0.02, 0.1, 0.3}) bank.push_back(Filter().critical_frequency(f0).build_biquad<Q32<30>>(IDSP_LOWPASS));
        std::vector<int32_t> xs(3 * 50);
        for (size_t i = 0; i < xs.size(); i++) xs[i] = int32_t((i * 2654435761u) >> 8) - (1 << 23);
        DeviceBuffer<int32_t> x(xs), y(xs.size());
        ByLane<Biquad<Q32<30>>, DirectForm1> bl(bank);
        bl.block(x, y);
        auto yh = y.to_host();
        for (size_t l = 0; l < 3; l++) {
            std::vector<int32_t> xl(50);
            for (size_t f = 0; f < 50; f++) xl[f] = xs[f * 3 + l];
            DeviceBuffer<int32_t> xd(xl), yd(50);
            Split(bank[l], DirectForm1{}).lanes(1).block(xd, yd);
            auto one = yd.to_host();
            for (size_t f = 0; f < 50; f++) EXPECT(one[f] == yh[f * 3 + l]);
        }
    }
    // src/cic.rs:223-240 (rate 0 is the identity) and :286-306 (Cic<i64, 3, 3>: gain_log2 6, gain 27)
    {
        CicDecimator<int64_t> id(3, 0, 1);
        DeviceBuffer<int64_t> x(std::vector<int64_t>{5, -7, 1ll << 40, -(1ll << 50)}), y(4);
        id.process_view(View<int64_t, LaneMajor>::from_flat(x, 1), ViewMut<int64_t, LaneMajor>::from_flat(y, 1));
        EXPECT(y.to_host() == x.to_host());
        CicDecimator<int64_t> unit(3, 0, 1, 3);
        EXPECT(unit.gain_log2() == 6 && unit.gain() == 27);
        // a settled rate-4 cubic interpolator holds x * gain (cic.rs:242-262): R = 4, N = 3 -> gain 64
        CicInterpolator<int32_t> up(3, 3, 1);
        DeviceBuffer<int32_t> xi(std::vector<int32_t>(8, 10)), yi(32);
        up.process_view(View<int32_t, LaneMajor>::from_flat(xi, 1), ViewMut<int32_t, LaneMajor>::from_flat(yi, 1, 4));
        EXPECT(up.gain() == 64 && up.response_length() == 9);
        auto o = yi.to_host();
        for (size_t i = up.response_length(); i < o.size(); i++) EXPECT(o[i] == 640);
    }
    // `Wdf<1, 0x1>::default()` is a unit delay; the bench's 0xad section quantises (tests/embedded/src/bin/biquad.rs:126)
    {
        idsp_wdf d{1, 0x1, {0}};
        WdfLanes w({d}, 1);
        DeviceBuffer<int32_t> x(std::vector<int32_t>{1, 2, 3, 4}), y(4);
        w.process_view(View<int32_t, LaneMajor>::from_flat(x, 1), ViewMut<int32_t, LaneMajor>::from_flat(y, 1));
        EXPECT((y.to_host() == std::vector<int32_t>{0, 1, 2, 3}));
        auto q = WdfLanes::quantize(0xad, {-0.9, 0.9});
        EXPECT(q.n == 2 && q.a[0] < 0 && q.a[1] < 0);
        // Normal::from (normal.rs:62-76): |p|^2 = a2 / a0
        auto nb = NormalLanes<double>::from_ba({0.2, 0.4, 0.2, 1.0, -1.2, 0.52});
        EXPECT(std::fabs(nb.ba[3] * nb.ba[3] + nb.ba[4] * nb.ba[4] - 0.52) < 1e-12);
    }
    // contract violations are reported as errors, never aborts
    {
        bool threw = false;
        try {
            DeviceBuffer<int32_t> x(8), y(4);
            Split(Biquad<Q32<30>>::identity(), DirectForm1{}).lanes(4).block(x, y);
        } catch (const Error &e) {
            threw = e.code == IDSP_EINVAL;
        }
        EXPECT(threw);
    }
    std::printf(failures ? "%d FAILURES\n" : "all host-layer tests passed\n", failures);
    return failures ? 1 : 0;
}
