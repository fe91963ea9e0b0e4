// coefficients.hip — host-side coefficient front-end of the C ABI (include/idsp_hip.h):
// `iir::coefficients::Filter` (src/iir/coefficients.rs), `iir::pid::{Builder, Pid}`
// (src/iir/pid.rs) and `iir::config::BiquadConfig::{build, try_build}` (src/iir/config.rs).
//
// Everything is a template over the builder float type T (f32 or f64) and keeps the
// reference's association and rounding points: each binary operation below is one
// operation of the reference expression, evaluated in T (-ffp-contract=off).  The
// transcendental functions are the platform libm ones in T, as for Rust's std.
// No device work; nothing here is on the per-sample path.
#include <cmath>
#include <cstring>
#include <limits>

#include "common.h"

namespace idsp {
namespace {

// ---------------------------------------------------------------- numeric casts
// Rust `as i32` from a float: truncate, saturate, NaN -> 0
template <class T>
int32_t as_i32(T v)
{
    if (std::isnan(v)) return 0;
    if (v >= T(2147483648.0)) return INT32_MAX;
    if (v <= T(-2147483648.0)) return INT32_MIN;
    return int32_t(v);
}
// float -> Q32<F>: `(v * 2^F).round() as i32` evaluated in T (dsp-fixedpoint/src/num_traits_impl.rs:32-46)
template <class T>
int32_t to_q(T v, int frac)
{
    return as_i32(std::round(v * T(double(int64_t(1) << frac))));
}

// Coefficient type C and sample type Y of `BiquadClamp<C, Y>` for the three outputs.
struct OutI32 {
    using C = int32_t;
    using Y = int32_t;
    using Rec = idsp_biquad_clamp_i32;
    int frac;
    template <class T> C coef(T v) const { return to_q(v, frac); }
    template <class T> Y samp(T v) const { return as_i32(v); }
    static C add(C a, C b) { return C(uint32_t(a) + uint32_t(b)); }  // release-mode wrapping Q add
    static C sub(C a, C b) { return C(uint32_t(a) - uint32_t(b)); }
    // `i32 * Q32<F>` (dsp-fixedpoint/src/lib.rs:449-456): ((y as i64 * c) >> F) as i32
    Y mul(Y y, C c) const { return Y((int64_t(y) * int64_t(c)) >> frac); }
    static Y ymin() { return INT32_MIN; }
    static Y ymax() { return INT32_MAX; }
};
template <class F>
struct OutFloat {
    using C = F;
    using Y = F;
    template <class T> C coef(T v) const { return C(v); }
    template <class T> Y samp(T v) const { return Y(v); }
    static C add(C a, C b) { return a + b; }
    static C sub(C a, C b) { return a - b; }
    Y mul(Y y, C c) const { return y * c; }
    static Y ymin() { return -std::numeric_limits<F>::infinity(); }  // `Clamp::MIN`, src/num.rs:33-52
    static Y ymax() { return std::numeric_limits<F>::infinity(); }
};
struct OutF32 : OutFloat<float> { using Rec = idsp_biquad_clamp_f32; };
struct OutF64 : OutFloat<double> { using Rec = idsp_biquad_clamp_f64; };

// ---------------------------------------------------------------- iir::Error
int non_finite(const char *n) { return fail(IDSP_ENONFINITE, "parameter `%s` must be finite", n); }
int non_positive(const char *n) { return fail(IDSP_ENONPOSITIVE, "parameter `%s` must be positive", n); }
int out_of_range(const char *n) { return fail(IDSP_EOUTOFRANGE, "parameter `%s` is out of range", n); }
int inverted(const char *n) { return fail(IDSP_EINVERTED, "range `%s` is inverted", n); }
int sign_mismatch(const char *n) { return fail(IDSP_ESIGN, "parameter `%s` has incompatible sign", n); }

template <class T> constexpr T kPi = T(3.14159265358979323846264338327950288);
template <class T> constexpr T kTau = T(6.28318530717958647692528676655900577);
template <class T> constexpr T kLn2 = T(0.693147180559945309417232121458176568);

// ---------------------------------------------------------------- coefficients::Filter<T>
template <class T>
struct Filter {
    T frequency, gain, shelf, shape;
    int kind;
};

// src/iir/coefficients.rs:240-263
template <class T>
int filter_validate(const Filter<T> &f)
{
    if (!std::isfinite(f.frequency)) return non_finite("frequency");
    if (f.frequency < T(0) || f.frequency > kPi<T>) return out_of_range("frequency");
    if (!std::isfinite(f.gain) || f.gain <= T(0)) return non_positive("gain");
    if (!std::isfinite(f.shelf) || f.shelf <= T(0)) return non_positive("shelf");
    switch (f.kind) {
        case IDSP_SHAPE_Q:
            if (!std::isfinite(f.shape)) return non_finite("q");
            if (f.shape <= T(0)) return non_positive("q");
            return IDSP_OK;
        case IDSP_SHAPE_BANDWIDTH:
            if (!std::isfinite(f.shape)) return non_finite("bandwidth");
            return IDSP_OK;
        default:
            if (!std::isfinite(f.shape)) return non_finite("slope");
            if (f.shape <= T(0)) return non_positive("slope");
            return IDSP_OK;
    }
}

// src/iir/coefficients.rs:266-277
template <class T>
T filter_qi(const Filter<T> &f)
{
    switch (f.kind) {
        case IDSP_SHAPE_Q: return T(1) / f.shape;
        case IDSP_SHAPE_BANDWIDTH:
            return T(2) * std::sinh(kLn2<T> / T(2) * f.shape * f.frequency / std::sin(f.frequency));
        default: return std::sqrt((f.shelf + T(1) / f.shelf) * (T(1) / f.shape - T(1)) + T(2));
    }
}

// src/iir/coefficients.rs:280-495; out = [b0,b1,b2,a0,a1,a2]
template <class T>
void filter_build(const Filter<T> &f, int type, T out[6])
{
    const T fsin = std::sin(f.frequency), fcos = std::cos(f.frequency);
    const T alpha = T(0.5) * fsin * filter_qi(f);
    const T one = T(1), g = f.gain;
    T b[3], a[3] = {one + alpha, T(-2) * fcos, one - alpha};
    switch (type) {
        case IDSP_LOWPASS: {
            const T v = g * T(0.5) * (one - fcos);
            b[0] = v, b[1] = T(2) * v, b[2] = v;
            break;
        }
        case IDSP_HIGHPASS: {
            const T v = g * T(0.5) * (one + fcos);
            b[0] = v, b[1] = T(-2) * v, b[2] = v;
            break;
        }
        case IDSP_BANDPASS: {
            const T v = g * alpha;
            b[0] = v, b[1] = T(0), b[2] = -v;
            break;
        }
        case IDSP_NOTCH: {
            const T f2 = T(-2) * fcos;
            b[0] = g, b[1] = f2 * g, b[2] = g;
            a[1] = f2;
            break;
        }
        case IDSP_ALLPASS: {
            const T f2 = T(-2) * fcos;
            b[0] = (one - alpha) * g, b[1] = f2 * g, b[2] = (one + alpha) * g;
            a[1] = f2;
            break;
        }
        case IDSP_PEAKING: {
            const T s = std::sqrt(f.shelf), f2 = T(-2) * fcos;
            b[0] = (one + alpha * s) * g, b[1] = f2 * g, b[2] = (one - alpha * s) * g;
            a[0] = one + alpha / s, a[1] = f2, a[2] = one - alpha / s;
            break;
        }
        case IDSP_LOWSHELF: {
            const T s = std::sqrt(f.shelf), tsa = T(2) * std::sqrt(s) * alpha, sp1 = s + one, sm1 = s - one;
            b[0] = s * g * (sp1 - sm1 * fcos + tsa);
            b[1] = T(2) * s * g * (sm1 - sp1 * fcos);
            b[2] = s * g * (sp1 - sm1 * fcos - tsa);
            a[0] = sp1 + sm1 * fcos + tsa;
            a[1] = T(-2) * (sm1 + sp1 * fcos);
            a[2] = sp1 + sm1 * fcos - tsa;
            break;
        }
        case IDSP_HIGHSHELF: {
            const T s = std::sqrt(f.shelf), tsa = T(2) * std::sqrt(s) * alpha, sp1 = s + one, sm1 = s - one;
            b[0] = s * g * (sp1 + sm1 * fcos + tsa);
            b[1] = T(-2) * s * g * (sm1 + sp1 * fcos);
            b[2] = s * g * (sp1 + sm1 * fcos - tsa);
            a[0] = sp1 - sm1 * fcos + tsa;
            a[1] = T(2) * (sm1 - sp1 * fcos);
            a[2] = sp1 - sm1 * fcos - tsa;
            break;
        }
        default: {  // IDSP_IHO
            const T hs = T(0.5) * std::sin(f.frequency);
            const T av = (one + fcos) / (T(2) * f.shelf);
            b[0] = g * (one + alpha), b[1] = T(-2) * g * fcos, b[2] = g * (one - alpha);
            a[0] = av + hs, a[1] = T(-2) * av, a[2] = av - hs;
            break;
        }
    }
    for (int i = 0; i < 3; i++) out[i] = b[i], out[3 + i] = a[i];
}

// `From<[[T;3];2]> for Biquad<C>` (src/iir/biquad.rs:545-566) then `From<[T;5]>` (:570-576)
template <class T, class O>
void normalize(const T sos[6], const O &o, typename O::C ba[5])
{
    const T a0 = T(1) / sos[3];
    ba[0] = o.coef(sos[0] * a0);
    ba[1] = o.coef(sos[1] * a0);
    ba[2] = o.coef(sos[2] * a0);
    ba[3] = o.coef(-sos[4] * a0);
    ba[4] = o.coef(-sos[5] * a0);
}

// ---------------------------------------------------------------- pid::Builder<T>
template <class T>
struct Builder {
    int order;
    T gain[5], limit[5];
};

template <class T>
Builder<T> builder_from(const idsp_pid_builder &b)
{
    Builder<T> r;
    r.order = b.order;
    for (int i = 0; i < 5; i++) r.gain[i] = T(b.gain[i]), r.limit[i] = T(b.limit[i]);
    return r;
}

// llvm.powi (compiler-rt __powisf2/__powidf2): square-and-multiply, reciprocal for b < 0
template <class T>
T powi(T a, int b)
{
    const bool recip = b < 0;
    T r = 1;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? T(1) / r : r;
}

template <class T>
T signum(T v)
{
    return std::isnan(v) ? v : std::copysign(T(1), v);
}

// src/iir/pid.rs:195-222
template <class T>
int builder_validate(const Builder<T> &b, T period)
{
    if (!std::isfinite(period)) return non_finite("period");
    if (period <= T(0)) return non_positive("period");
    for (int i = 0; i < 5; i++)
        if (std::isnan(b.gain[i])) return non_finite("gain");
    for (int i = 0; i < 5; i++)
        if (std::isnan(b.limit[i])) return non_finite("limit");
    for (int action : {0, 1, 3, 4}) {
        const T gain = b.gain[action], limit = b.limit[action];
        if (std::isfinite(limit)) {
            if (limit == T(0)) return non_positive("limit");
            if (gain != T(0) && signum(gain) != signum(limit)) return sign_mismatch("gain/limit");
        }
    }
    return IDSP_OK;
}

// src/iir/pid.rs:256-313
template <class T, class O>
void builder_build(const Builder<T> &b, T period, const O &o, typename O::C out[5])
{
    using Cc = typename O::C;
    T z = powi(period, -b.order);
    T gl[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    for (int j = 2; j >= 0; j--) {  // zip(gl, enumerate(gain, limit).skip(order)).rev()
        const int i = b.order + j;
        gl[j][0] = b.gain[i] * z;
        gl[j][1] = i == 2 ? T(1) : gl[j][0] / b.limit[i];
        z = z * period;
    }
    const T a0i = T(1) / (gl[0][1] + gl[1][1] + gl[2][1]);
    static const int kernels[3][3] = {{1, 0, 0}, {1, -1, 0}, {1, -2, 1}};
    const Cc zero = o.coef(T(0));
    Cc ba[3][2] = {{zero, zero}, {zero, zero}, {zero, zero}};
    for (int j = 0; j < 3; j++) {
        // quantize the gains, not the coefficients
        const Cc g0 = o.coef(gl[j][0] * a0i), g1 = o.coef(gl[j][1] * a0i);
        for (int m = 0; m < 3; m++) {
            const int k = kernels[j][m];
            for (int r = 0; r < (k > 0 ? k : -k); r++) {
                if (k > 0) {
                    ba[m][0] = O::add(ba[m][0], g0);
                    ba[m][1] = O::sub(ba[m][1], g1);
                } else {
                    ba[m][0] = O::sub(ba[m][0], g0);
                    ba[m][1] = O::add(ba[m][1], g1);
                }
            }
        }
    }
    out[0] = ba[0][0], out[1] = ba[1][0], out[2] = ba[2][0], out[3] = ba[1][1], out[4] = ba[2][1];
}

template <class T, class O>
int pid_build(const idsp_pid_builder *b, double period, int validate, const O &o, typename O::C out[5])
{
    const Builder<T> bt = builder_from<T>(*b);
    if (validate) {
        const int rc = builder_validate(bt, T(period));
        if (rc) return rc;
    }
    builder_build(bt, T(period), o, out);
    return IDSP_OK;
}

int order_ok(int order)
{
    if (order < 0 || order > 2) return fail(IDSP_EINVAL, "pid order %d is none of P (2), I (1), I2 (0)", order);
    return IDSP_OK;
}

// ---------------------------------------------------------------- BiquadClamp<C, Y> assembly
template <class O>
void store_clamp(const O &, typename O::Rec *out, const typename O::C ba[5], typename O::Y u, typename O::Y mn,
                 typename O::Y mx)
{
    for (int i = 0; i < 5; i++) out->ba[i] = ba[i];
    out->u = u, out->min = mn, out->max = mx;
}
inline void set_frac(idsp_biquad_clamp_i32 *out, const OutI32 &o) { out->frac = o.frac; }
template <class R, class O> void set_frac(R *, const O &) {}

// src/iir/pid.rs:497-518
template <class T>
int units_pid_validate(const idsp_pid &p, const idsp_units &u)
{
    if (T(p.min) > T(p.max)) return inverted("output_limits");
    const char *names[3] = {"t", "x", "y"};
    const T vals[3] = {T(u.t), T(u.x), T(u.y)};
    for (int i = 0; i < 3; i++) {
        if (!std::isfinite(vals[i])) return non_finite(names[i]);
        if (vals[i] <= T(0)) return non_positive(names[i]);
    }
    return builder_validate(builder_from<T>(p.builder), T(u.t));
}

// src/iir/pid.rs:533-567
template <class T, class O>
int pid_build_clamp(const idsp_pid *p, const idsp_units *units, int validate, const O &o, typename O::Rec *out)
{
    if (validate) {
        const int rc = units_pid_validate<T>(*p, *units);
        if (rc) return rc;
    }
    const T yu = T(1) / T(units->y);
    const T yx = T(units->x) * yu;
    const T pg = T(p->builder.gain[2]);
    Builder<T> b;
    b.order = p->builder.order;
    for (int i = 0; i < 5; i++) {
        b.gain[i] = yx * std::copysign(T(p->builder.gain[i]), pg);
        T l = T(p->builder.limit[i]);
        if (std::isnan(l)) l = std::numeric_limits<T>::infinity();
        b.limit[i] = yx * std::copysign(l, pg);
    }
    typename O::C ba[5];
    builder_build(b, T(units->t), o, ba);
    // set_input_offset (src/iir/biquad.rs:253-255): u = i * (b0 + b1 + b2)
    const typename O::Y i = o.samp(-T(p->setpoint) * (T(1) / T(units->x)));
    const typename O::C fg = O::add(O::add(ba[0], ba[1]), ba[2]);
    std::memset(out, 0, sizeof(*out));
    store_clamp(o, out, ba, o.mul(i, fg), o.samp(T(p->min) * yu), o.samp(T(p->max) * yu));
    set_frac(out, o);
    return IDSP_OK;
}

// src/iir/config.rs:309-344
template <class T>
int check_offset_limits(T offset, T mn, T mx)
{
    if (!std::isfinite(offset)) return non_finite("offset");
    if (std::isnan(mn) || std::isnan(mx)) return non_finite("output_limits");
    if (mn > mx) return inverted("output_limits");
    return IDSP_OK;
}
template <class T>
int check_units(const idsp_units &u, bool check_t)
{
    const char *names[2] = {"x", "y"};
    const T vals[2] = {T(u.x), T(u.y)};
    for (int i = 0; i < 2; i++) {
        if (!std::isfinite(vals[i])) return non_finite(names[i]);
        if (vals[i] <= T(0)) return non_positive(names[i]);
    }
    if (check_t) {
        if (!std::isfinite(T(u.t))) return non_finite("t");
        if (T(u.t) <= T(0)) return non_positive("t");
    }
    return IDSP_OK;
}

// common tail of the Ba and Filter arms (config.rs:361-366,379-384)
template <class T, class O>
void finish_ba(T sos[6], T yx, T yu, T offset, T mn, T mx, const O &o, typename O::Rec *out)
{
    for (int i = 0; i < 3; i++) sos[i] = sos[i] * yx;
    typename O::C ba[5];
    normalize(sos, o, ba);
    std::memset(out, 0, sizeof(*out));
    store_clamp(o, out, ba, o.samp(offset * yu), o.samp(mn * yu), o.samp(mx * yu));
    set_frac(out, o);
}

// `BiquadConfig::Ba` (config.rs:359-367, 389-407)
template <class T, class O>
int config_ba_build(const idsp_ba_config *c, const idsp_units *units, int validate, const O &o, typename O::Rec *out)
{
    T sos[6];
    for (int i = 0; i < 6; i++) sos[i] = T(c->ba[i]);
    if (validate) {
        int rc = check_units<T>(*units, false);
        if (rc) return rc;
        if ((rc = check_offset_limits(T(c->offset), T(c->min), T(c->max)))) return rc;
        for (int i = 0; i < 6; i++)
            if (!std::isfinite(sos[i])) return non_finite("ba");
    }
    const T yu = T(1) / T(units->y);
    finish_ba(sos, T(units->x) * yu, yu, T(c->offset), T(c->min), T(c->max), o, out);
    return IDSP_OK;
}

// `BiquadConfig::Filter` (config.rs:372-385, 409-427)
template <class T, class O>
int config_filter_build(const idsp_filter_config *c, const idsp_units *units, int validate, const O &o,
                        typename O::Rec *out)
{
    if (validate) {
        int rc = check_units<T>(*units, true);
        if (rc) return rc;
        if ((rc = check_offset_limits(T(c->offset), T(c->min), T(c->max)))) return rc;
    }
    const T yu = T(1) / T(units->y);
    Filter<T> f;
    f.gain = std::pow(T(10), T(c->gain_db) / T(20));                 // gain_db (coefficients.rs:157-159)
    f.frequency = kTau<T> * (T(c->frequency) * T(units->t));          // critical_frequency (:131-133)
    f.shelf = std::pow(T(10), T(c->shelf_db) / T(20));               // shelf_db (:177-179)
    f.shape = T(c->shape), f.kind = c->shape_kind;
    if (validate) {
        const int rc = filter_validate(f);
        if (rc) return rc;
    }
    T sos[6];
    filter_build(f, c->typ, sos);
    finish_ba(sos, T(units->x) * yu, yu, T(c->offset), T(c->min), T(c->max), o, out);
    return IDSP_OK;
}

int type_ok(int type)
{
    if (type < IDSP_LOWPASS || type > IDSP_IHO) return fail(IDSP_EINVAL, "filter type %d not in 0..8", type);
    return IDSP_OK;
}
int shape_ok(int kind)
{
    if (kind < IDSP_SHAPE_Q || kind > IDSP_SHAPE_SLOPE) return fail(IDSP_EINVAL, "shape kind %d not in 0..2", kind);
    return IDSP_OK;
}
int frac_ok(int frac)
{
    if (frac < 0 || frac > 31) return fail(IDSP_EINVAL, "frac = %d not in 0..31", frac);
    return IDSP_OK;
}

#define IDSP_CHECK(expr)         \
    do {                         \
        const int rc_ = (expr);  \
        if (rc_) return rc_;     \
    } while (0)

}  // namespace
}  // namespace idsp

using namespace idsp;

extern "C" {

int idsp_filter_build(const idsp_filter *f, int type, int validate, double ba[6])
{
    if (!f || !ba) return fail(IDSP_EINVAL, "f or ba is NULL");
    IDSP_CHECK(type_ok(type));
    IDSP_CHECK(shape_ok(f->shape_kind));
    if (f->f32) {
        const Filter<float> ft{float(f->frequency), float(f->gain), float(f->shelf), float(f->shape), f->shape_kind};
        if (validate) IDSP_CHECK(filter_validate(ft));
        float o[6];
        filter_build(ft, type, o);
        for (int i = 0; i < 6; i++) ba[i] = double(o[i]);
    } else {
        const Filter<double> ft{f->frequency, f->gain, f->shelf, f->shape, f->shape_kind};
        if (validate) IDSP_CHECK(filter_validate(ft));
        filter_build(ft, type, ba);
    }
    return IDSP_OK;
}

int idsp_pid_build_i32(const idsp_pid_builder *b, double period, int validate, int frac, int32_t ba[5])
{
    if (!b || !ba) return fail(IDSP_EINVAL, "b or ba is NULL");
    IDSP_CHECK(order_ok(b->order));
    IDSP_CHECK(frac_ok(frac));
    const OutI32 o{frac};
    return b->f32 ? pid_build<float>(b, period, validate, o, ba) : pid_build<double>(b, period, validate, o, ba);
}

int idsp_pid_build_f32(const idsp_pid_builder *b, double period, int validate, float ba[5])
{
    if (!b || !ba) return fail(IDSP_EINVAL, "b or ba is NULL");
    IDSP_CHECK(order_ok(b->order));
    const OutF32 o{};
    return b->f32 ? pid_build<float>(b, period, validate, o, ba) : pid_build<double>(b, period, validate, o, ba);
}

int idsp_pid_build_f64(const idsp_pid_builder *b, double period, int validate, double ba[5])
{
    if (!b || !ba) return fail(IDSP_EINVAL, "b or ba is NULL");
    IDSP_CHECK(order_ok(b->order));
    const OutF64 o{};
    return b->f32 ? pid_build<float>(b, period, validate, o, ba) : pid_build<double>(b, period, validate, o, ba);
}

#define IDSP_CLAMP_ENTRY(FN, CFG_T, IMPL, PRECHECK)                                                            \
    int FN##_i32(const CFG_T *c, const idsp_units *units, int validate, int frac, idsp_biquad_clamp_i32 *out)  \
    {                                                                                                          \
        if (!c || !units || !out) return fail(IDSP_EINVAL, "NULL argument");                                   \
        IDSP_CHECK(frac_ok(frac));                                                                             \
        PRECHECK;                                                                                              \
        const OutI32 o{frac};                                                                                  \
        return F32_OF(c) ? IMPL<float>(c, units, validate, o, out) : IMPL<double>(c, units, validate, o, out); \
    }                                                                                                          \
    int FN##_f32(const CFG_T *c, const idsp_units *units, int validate, idsp_biquad_clamp_f32 *out)            \
    {                                                                                                          \
        if (!c || !units || !out) return fail(IDSP_EINVAL, "NULL argument");                                   \
        PRECHECK;                                                                                              \
        const OutF32 o{};                                                                                      \
        return F32_OF(c) ? IMPL<float>(c, units, validate, o, out) : IMPL<double>(c, units, validate, o, out); \
    }                                                                                                          \
    int FN##_f64(const CFG_T *c, const idsp_units *units, int validate, idsp_biquad_clamp_f64 *out)            \
    {                                                                                                          \
        if (!c || !units || !out) return fail(IDSP_EINVAL, "NULL argument");                                   \
        PRECHECK;                                                                                              \
        const OutF64 o{};                                                                                      \
        return F32_OF(c) ? IMPL<float>(c, units, validate, o, out) : IMPL<double>(c, units, validate, o, out); \
    }

#define F32_OF(c) ((c)->builder.f32)
IDSP_CLAMP_ENTRY(idsp_pid_build_clamp, idsp_pid, pid_build_clamp, IDSP_CHECK(order_ok(c->builder.order)))
#undef F32_OF
#define F32_OF(c) ((c)->f32)
IDSP_CLAMP_ENTRY(idsp_config_ba_build, idsp_ba_config, config_ba_build, (void)0)
IDSP_CLAMP_ENTRY(idsp_config_filter_build, idsp_filter_config, config_filter_build,
                 IDSP_CHECK(type_ok(c->typ)); IDSP_CHECK(shape_ok(c->shape_kind)))
#undef F32_OF

}  // extern "C"
