// Coefficient front-end of the C++ host layer (include/idsp_hip.hpp: Filter, pid::Builder, Pid,
// Units, build_config) written like the reference's doctests/tests.  Host code only: runs
// without a GPU.
#include <cmath>
#include <cstdio>

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

int main()
{
    // src/iir/coefficients.rs:289-300: Filter::default().critical_frequency(0.1).gain(1000.0).lowpass().into()
    {
        auto iir = Filter().critical_frequency(0.1).gain(1000.0).build_biquad<Q32<30>>(IDSP_LOWPASS);
        const std::array<int32_t, 5> want{2147483647, 2147483647, 2147483647, 1227265970, -443242341};
        EXPECT(iir.ba == want);
        auto hp = Filter().critical_frequency(0.1).gain(1000.0).try_build_biquad<float>(IDSP_HIGHPASS);
        EXPECT(hp.ba[0] == hp.ba[2] && hp.ba[1] == -2.0f * hp.ba[0]);
    }
    // src/iir/pid.rs:574-590
    {
        auto b = pid::Builder().gain(pid::I, 1e-3).gain(pid::P, 1.0).gain(pid::D, 1e2).limit(pid::I, 1e3).limit(pid::D, 1e1)
                     .build<float>(1.0);
        const float want[5] = {9.181909f, -18.272726f, 9.090908f, 1.9090908f, -0.9090908f};
        for (int i = 0; i < 5; i++) EXPECT(std::fabs(b.ba[i] / want[i] - 1.0f) < 2.0f * 1.1920929e-07f);
        // pid.rs:251-255
        auto p = pid::Builder().gain(pid::P, 3.0).order(pid::OrderP).build<float>(1.0);
        EXPECT(p.ba == Biquad<float>::proportional(3.0f).ba);
        // pid.rs:592-603 (Builder<f32> -> Biquad<Q32<29>>)
        auto q = pid::Builder(true).ki(1e-5).kp(1e-2).kd(1e0).limit_i(1e1).limit_d(1e-1).build<Q32<29>>(1.0);
        EXPECT(q.ba[0] > 0 && q.ba[1] < 0);
    }
    // validation errors carry the reference's `iir::Error` text (src/iir/error.rs:18-28)
    {
        bool ok = false;
        try {
            Filter().critical_frequency(0.6).try_build(IDSP_LOWPASS);
        } catch (const Error &e) {
            ok = e.code == IDSP_EOUTOFRANGE && std::string(e.what()).find("parameter `frequency` is out of range") != std::string::npos;
        }
        EXPECT(ok);
        ok = false;
        try {
            pid::Builder().ki(1.0).limit_i(-1.0).try_build<double>(1.0);
        } catch (const Error &e) {
            ok = e.code == IDSP_ESIGN;
        }
        EXPECT(ok);
        // src/iir/config.rs:188-200
        idsp_ba_config ba{{0, 0, 0, 1, 0, 0}, 0.0, 1.0, 0.0, 1};
        ok = false;
        try {
            build_config<float>(ba, Units(), true);
        } catch (const Error &e) {
            ok = e.code == IDSP_EINVERTED;
        }
        EXPECT(ok);
        EXPECT(build_config<float>(ba, Units(), false).min == 1.0f);  // unchecked `build`
    }
    // Pid + Units -> BiquadClamp (src/iir/pid.rs:533-567): limits scale by 1/units.y, u = -setpoint/x * (b0+b1+b2)
    {
        auto c = Pid().kp(-2.0).ki(-10.0).limit_i(-50.0).setpoint(0.5).output_limits(-1.0, 1.0).try_build<double>(Units(1e-3, 2.0, 4.0));
        EXPECT(c.min == -0.25 && c.max == 0.25);
        EXPECT(c.u == -0.25 * c.coeff.forward_gain());
        auto ci = Pid().order(pid::OrderP).kp(1.0).setpoint(100.0).build<Q32<28>>(Units());
        EXPECT(ci.coeff.ba[0] == (1 << 28) && ci.u == -100);
    }
    std::printf(failures ? "%d FAILURES\n" : "all coefficient front-end tests passed\n", failures);
    return failures ? 1 : 0;
}
