/*
 * idsp_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * See idsp_oracle.h for the role and pinning status of this file.
 *
 * Scalar restatement of the reference hot path (quartiq/idsp 0.22.0).  Each
 * function cites the reference lines it follows.  Arithmetic conventions:
 *   - Rust release-mode integer arithmetic wraps; done here through unsigned
 *     types so that no C signed-overflow UB is involved.
 *   - `>>` on signed values is arithmetic (gcc), Rust `as` integer casts
 *     truncate, float -> int `as` saturates with NaN -> 0.
 *   - build with -ffp-contract=off and without fast-math: Rust never fuses
 *     `a*b + c`.
 */
#include "idsp_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- helpers */

static inline int64_t wadd64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
static inline int32_t wadd32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int64_t wshl64(int64_t a, int s) { return (int64_t)((uint64_t)a << s); }
static inline int64_t mul_wide(int32_t c, int32_t v) { return (int64_t)c * (int64_t)v; }
static inline int32_t trunc32(int64_t a) { return (int32_t)(uint32_t)(uint64_t)a; }

static inline size_t idx_of(size_t f, size_t l, size_t lanes, size_t frames, int layout)
{
    return layout == IDSP_LANE_MAJOR ? l * frames + f : f * lanes + l;
}

static inline float f32_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static int check_common(const void *cfg, size_t n, const void *state, const void *x, const void *y,
                        size_t lanes, size_t frames, int layout)
{
    if (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR) return IDSP_EINVAL;
    if (n > IDSP_MAX_SECTIONS) return IDSP_EINVAL;
    if (n && !cfg) return IDSP_EINVAL;
    if (lanes && n && !state) return IDSP_EINVAL;
    if (lanes && frames && (!x || !y)) return IDSP_EINVAL;
    return IDSP_OK;
}

/* num_traits::clamp: `if input < min {min} else if input > max {max} else {input}` */
static inline int32_t clamp_i32(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float clamp_f32(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------ coefficient ingestion (a8) */

/* dsp-fixedpoint/src/num_traits_impl.rs:32-46: Q::new((v * (1/DELTA)).round().as_())
 * with DELTA = 2^-F (lib.rs:220-224); `round` = half away from zero; `as i32`
 * saturates, NaN -> 0. */
int32_t idsp_ref_quantize_f64(double v, int frac)
{
    double s = round(v * ldexp(1.0, frac));
    if (isnan(s)) return 0;
    if (s >= 2147483647.0) return INT32_MAX;
    if (s <= -2147483648.0) return INT32_MIN;
    return (int32_t)s;
}

/* src/iir/biquad.rs:545-566 `From<[[f64;3];2]>`: a0 = 1/ba[1][0];
 * [b0*a0, b1*a0, b2*a0, -a1*a0, -a2*a0] then `From<[T;5]>` (:570-576). */
int idsp_ref_biquad_i32_from_sos(const double sos[6], int frac, idsp_biquad_i32 *out)
{
    if (!sos || !out || frac < 0 || frac > 31) return IDSP_EINVAL;
    double a0 = 1.0 / sos[3];
    double ba[5] = {sos[0] * a0, sos[1] * a0, sos[2] * a0, -sos[4] * a0, -sos[5] * a0};
    for (int i = 0; i < 5; i++) out->ba[i] = idsp_ref_quantize_f64(ba[i], frac);
    out->frac = frac;
    return IDSP_OK;
}

int idsp_ref_biquad_f32_from_sos(const float sos[6], idsp_biquad_f32 *out)
{
    if (!sos || !out) return IDSP_EINVAL;
    float a0 = 1.0f / sos[3];
    out->ba[0] = sos[0] * a0;
    out->ba[1] = sos[1] * a0;
    out->ba[2] = sos[2] * a0;
    out->ba[3] = -sos[4] * a0;
    out->ba[4] = -sos[5] * a0;
    return IDSP_OK;
}

int idsp_ref_biquad_f32_from_sos_f64(const double sos[6], idsp_biquad_f32 *out)
{
    if (!sos || !out) return IDSP_EINVAL;
    double a0 = 1.0 / sos[3];
    double ba[5] = {sos[0] * a0, sos[1] * a0, sos[2] * a0, -sos[4] * a0, -sos[5] * a0};
    for (int i = 0; i < 5; i++) out->ba[i] = (float)ba[i];
    return IDSP_OK;
}

int idsp_ref_biquad_f64_from_sos(const double sos[6], idsp_biquad_f64 *out)
{
    if (!sos || !out) return IDSP_EINVAL;
    double a0 = 1.0 / sos[3];
    out->ba[0] = sos[0] * a0;
    out->ba[1] = sos[1] * a0;
    out->ba[2] = sos[2] * a0;
    out->ba[3] = -sos[4] * a0;
    out->ba[4] = -sos[5] * a0;
    return IDSP_OK;
}

/* src/iir/coefficients.rs:259-283: fcos_alpha() and lowpass(). */
void idsp_ref_filter_lowpass(double w0, double gain, double q, double sos[6])
{
    double fsin = sin(w0), fcos = cos(w0);
    double alpha = 0.5 * fsin * (1.0 / q);
    double b = gain * 0.5 * (1.0 - fcos);
    sos[0] = b; sos[1] = 2.0 * b; sos[2] = b;
    sos[3] = 1.0 + alpha; sos[4] = -2.0 * fcos; sos[5] = 1.0 - alpha;
}

/* src/iir/coefficients.rs:302-335: highpass(). */
void idsp_ref_filter_highpass(double w0, double gain, double q, double sos[6])
{
    double fsin = sin(w0), fcos = cos(w0);
    double alpha = 0.5 * fsin * (1.0 / q);
    double b = gain * 0.5 * (1.0 + fcos);
    sos[0] = b; sos[1] = -2.0 * b; sos[2] = b;
    sos[3] = 1.0 + alpha; sos[4] = -2.0 * fcos; sos[5] = 1.0 - alpha;
}

/* ------------------------------------------------------- biquad sections */
/* Local state is the per-lane word record of include/idsp_hip.h. */

/* src/iir/biquad.rs:366-383 with C = Q<i32,i64,F>: products widen to i64
 * (dsp-fixedpoint/src/ops.rs:91-97, lib.rs:310-312), Q+Q adds (ops.rs:63-76),
 * `.as_()` = quantize = (acc >> F) as i32 (num_traits_impl.rs:74-85,
 * lib.rs:297-299,270-272). */
static inline int64_t sum5_i32(const int32_t ba[5], int32_t x0, int32_t x1, int32_t x2, int32_t y1, int32_t y2)
{
    int64_t acc = mul_wide(ba[0], x0);
    acc = wadd64(acc, mul_wide(ba[1], x1));
    acc = wadd64(acc, mul_wide(ba[2], x2));
    acc = wadd64(acc, mul_wide(ba[3], y1));
    acc = wadd64(acc, mul_wide(ba[4], y2));
    return acc;
}

static inline int32_t df1_i32(const int32_t ba[5], int frac, uint32_t *s, int32_t x0)
{
    int32_t y0 = trunc32(sum5_i32(ba, x0, (int32_t)s[0], (int32_t)s[1], (int32_t)s[2], (int32_t)s[3]) >> frac);
    s[1] = s[0]; s[0] = (uint32_t)x0;
    s[3] = s[2]; s[2] = (uint32_t)y0;
    return y0;
}

/* src/iir/biquad.rs:394-404: clamp(inner + u, min, max); state.y[0][0] = y0. */
static inline int32_t df1_clamp_i32(const idsp_biquad_clamp_i32 *c, uint32_t *s, int32_t x0)
{
    int32_t y0 = clamp_i32(wadd32(df1_i32(c->ba, c->frac, s, x0), c->u), c->min, c->max);
    s[2] = (uint32_t)y0;
    return y0;
}

/* src/iir/biquad.rs:511-530. */
static inline int32_t dither_i32(const int32_t ba[5], int frac, uint32_t *s, int32_t x0)
{
    int64_t acc = wadd64((int64_t)(uint64_t)s[4],
                         sum5_i32(ba, x0, (int32_t)s[0], (int32_t)s[1], (int32_t)s[2], (int32_t)s[3]));
    acc = wshl64(acc, 32 - frac);
    /* `(acc as u32) >> (32 - F)`; for F = 0 the low word is zero after `<<= 32`
     * and Rust's release-mode shift masks the amount, giving 0. */
    s[4] = frac == 0 ? 0u : ((uint32_t)(uint64_t)acc) >> (32 - frac);
    int32_t y0 = trunc32(acc >> 32);
    s[1] = s[0]; s[0] = (uint32_t)x0;
    s[3] = s[2]; s[2] = (uint32_t)y0;
    return y0;
}

/* src/iir/biquad.rs:532-538. */
static inline int32_t dither_clamp_i32(const idsp_biquad_clamp_i32 *c, uint32_t *s, int32_t x0)
{
    int32_t y0 = clamp_i32(wadd32(dither_i32(c->ba, c->frac, s, x0), c->u), c->min, c->max);
    s[2] = (uint32_t)y0;
    return y0;
}

/* src/iir/biquad.rs:456-472.  Words: x0,x1,y0.lo,y0.hi,y1.lo,y1.hi. */
static inline int32_t wide_i32(const int32_t ba[5], int frac, uint32_t *s, int32_t x0)
{
    int64_t acc = mul_wide(ba[0], x0);
    acc = wadd64(acc, mul_wide(ba[1], (int32_t)s[0]));
    acc = wadd64(acc, mul_wide(ba[2], (int32_t)s[1]));
    s[1] = s[0]; s[0] = (uint32_t)x0;
    int64_t y0 = (int64_t)(((uint64_t)s[3] << 32) | s[2]);
    int64_t y1 = (int64_t)(((uint64_t)s[5] << 32) | s[4]);
    acc = wadd64(acc, ((int64_t)(uint64_t)(uint32_t)y0 * (int64_t)ba[3]) >> 32);
    acc = wadd64(acc, (int64_t)(int32_t)(y0 >> 32) * (int64_t)ba[3]);
    acc = wadd64(acc, ((int64_t)(uint64_t)(uint32_t)y1 * (int64_t)ba[4]) >> 32);
    acc = wadd64(acc, (int64_t)(int32_t)(y1 >> 32) * (int64_t)ba[4]);
    acc = wshl64(acc, 32 - frac);
    s[4] = s[2]; s[5] = s[3];
    s[2] = (uint32_t)(uint64_t)acc; s[3] = (uint32_t)((uint64_t)acc >> 32);
    return trunc32(acc >> 32);
}

/* src/iir/biquad.rs:474-480: state.y[0] = ((y0 as i64) << 32) | state.y[0] as u32 as i64. */
static inline int32_t wide_clamp_i32(const idsp_biquad_clamp_i32 *c, uint32_t *s, int32_t x0)
{
    int32_t y0 = clamp_i32(wadd32(wide_i32(c->ba, c->frac, s, x0), c->u), c->min, c->max);
    s[3] = (uint32_t)y0;
    return y0;
}

/* src/iir/biquad.rs:366-383 with C = T = A = f32: ((((b0*x0 + b1*x1) + b2*x2) + a1*y1) + a2*y2). */
static inline float df1_f32(const float ba[5], uint32_t *s, float x0)
{
    float x1 = f32_from_bits(s[0]), x2 = f32_from_bits(s[1]);
    float y1 = f32_from_bits(s[2]), y2 = f32_from_bits(s[3]);
    float acc = ba[0] * x0;
    acc = acc + ba[1] * x1;
    acc = acc + ba[2] * x2;
    acc = acc + ba[3] * y1;
    acc = acc + ba[4] * y2;
    s[1] = s[0]; s[0] = f32_to_bits(x0);
    s[3] = s[2]; s[2] = f32_to_bits(acc);
    return acc;
}

static inline float df1_clamp_f32(const idsp_biquad_clamp_f32 *c, uint32_t *s, float x0)
{
    float y0 = clamp_f32(df1_f32(c->ba, s, x0) + c->u, c->min, c->max);
    s[2] = f32_to_bits(y0);
    return y0;
}

/* src/iir/biquad.rs:418-428. */
static inline float df2t_f32(const float ba[5], uint32_t *s, float x0)
{
    float s0 = f32_from_bits(s[0]), s1 = f32_from_bits(s[1]);
    float y0 = s0 + ba[0] * x0;
    float n0 = (s1 + ba[1] * x0) + ba[3] * y0;
    float n1 = ba[2] * x0 + ba[4] * y0;
    s[0] = f32_to_bits(n0); s[1] = f32_to_bits(n1);
    return y0;
}

/* src/iir/biquad.rs:430-440. */
static inline float df2t_clamp_f32(const idsp_biquad_clamp_f32 *c, uint32_t *s, float x0)
{
    float s0 = f32_from_bits(s[0]), s1 = f32_from_bits(s[1]);
    float y0 = clamp_f32((s0 + c->ba[0] * x0) + c->u, c->min, c->max);
    float n0 = (s1 + c->ba[1] * x0) + c->ba[3] * y0;
    float n1 = c->ba[2] * x0 + c->ba[4] * y0;
    s[0] = f32_to_bits(n0); s[1] = f32_to_bits(n1);
    return y0;
}

/* f64: the same generic impls (src/iir/biquad.rs:366-383,418-440) with C = T = A = f64;
 * every value occupies two state words, low word first. */
static inline double f64_from_words(const uint32_t *s, int i)
{
    uint64_t u = (uint64_t)s[2 * i] | ((uint64_t)s[2 * i + 1] << 32);
    double d;
    memcpy(&d, &u, 8);
    return d;
}
static inline void f64_to_words(uint32_t *s, int i, double d)
{
    uint64_t u;
    memcpy(&u, &d, 8);
    s[2 * i] = (uint32_t)u;
    s[2 * i + 1] = (uint32_t)(u >> 32);
}
static inline double clamp_f64(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

static inline double df1_f64(const double ba[5], uint32_t *s, double x0)
{
    double x1 = f64_from_words(s, 0), x2 = f64_from_words(s, 1);
    double y1 = f64_from_words(s, 2), y2 = f64_from_words(s, 3);
    double acc = ba[0] * x0;
    acc = acc + ba[1] * x1;
    acc = acc + ba[2] * x2;
    acc = acc + ba[3] * y1;
    acc = acc + ba[4] * y2;
    f64_to_words(s, 1, x1); f64_to_words(s, 0, x0);
    f64_to_words(s, 3, y1); f64_to_words(s, 2, acc);
    return acc;
}

static inline double df1_clamp_f64(const idsp_biquad_clamp_f64 *c, uint32_t *s, double x0)
{
    double y0 = clamp_f64(df1_f64(c->ba, s, x0) + c->u, c->min, c->max);
    f64_to_words(s, 2, y0);
    return y0;
}

static inline double df2t_f64(const double ba[5], uint32_t *s, double x0)
{
    double s0 = f64_from_words(s, 0), s1 = f64_from_words(s, 1);
    double y0 = s0 + ba[0] * x0;
    double n0 = (s1 + ba[1] * x0) + ba[3] * y0;
    double n1 = ba[2] * x0 + ba[4] * y0;
    f64_to_words(s, 0, n0); f64_to_words(s, 1, n1);
    return y0;
}

static inline double df2t_clamp_f64(const idsp_biquad_clamp_f64 *c, uint32_t *s, double x0)
{
    double s0 = f64_from_words(s, 0), s1 = f64_from_words(s, 1);
    double y0 = clamp_f64((s0 + c->ba[0] * x0) + c->u, c->min, c->max);
    double n0 = (s1 + c->ba[1] * x0) + c->ba[3] * y0;
    double n1 = c->ba[2] * x0 + c->ba[4] * y0;
    f64_to_words(s, 0, n0); f64_to_words(s, 1, n1);
    return y0;
}

static int frac_ok_i32(const idsp_biquad_i32 *c, size_t n)
{
    for (size_t i = 0; i < n; i++) if (c[i].frac < 0 || c[i].frac > 31) return 0;
    return 1;
}
static int frac_ok_clamp(const idsp_biquad_clamp_i32 *c, size_t n)
{
    for (size_t i = 0; i < n; i++) if (c[i].frac < 0 || c[i].frac > 31) return 0;
    return 1;
}

/*
 * Lane driver.  Per lane it follows the serial slice composition
 * `[C] x [S]` STAGE-major (dsp-process/src/compose.rs:43-77: first section
 * `block(x, y)`, every further section `inplace(y)`), each of those being the
 * default per-sample loops of process.rs:122-127,137-141; the lanes are visited
 * one after the other like `Lanes::process_view` (compose.rs:478-494).
 */
#define LANE_DRIVER(NAME, CFG_T, T, W, FRAC_CHECK, CALL)                                          \
    static int NAME##_range(const CFG_T *cfg, size_t n, void *state, const T *x, T *y, size_t lanes, \
                            size_t frames, int layout, size_t l0, size_t l1)                       \
    {                                                                                              \
        uint32_t *st = (uint32_t *)state;                                                          \
        for (size_t l = l0; l < l1; l++) {                                                         \
            if (n == 0) {                                                                          \
                for (size_t f = 0; f < frames; f++) {                                              \
                    size_t i = idx_of(f, l, lanes, frames, layout);                                \
                    y[i] = x[i];                                                                   \
                }                                                                                  \
                continue;                                                                          \
            }                                                                                      \
            for (size_t k = 0; k < n; k++) {                                                       \
                const CFG_T *c = &cfg[k];                                                          \
                uint32_t s[W];                                                                     \
                for (int w = 0; w < W; w++) s[w] = st[(k * W + w) * lanes + l];                    \
                const T *src = k == 0 ? x : y;                                                     \
                for (size_t f = 0; f < frames; f++) {                                              \
                    size_t i = idx_of(f, l, lanes, frames, layout);                                \
                    T x0 = src[i];                                                                 \
                    y[i] = CALL;                                                                   \
                }                                                                                  \
                for (int w = 0; w < W; w++) st[(k * W + w) * lanes + l] = s[w];                    \
            }                                                                                      \
        }                                                                                          \
        return IDSP_OK;                                                                            \
    }                                                                                              \
    int NAME(const CFG_T *cfg, size_t n, void *state, const T *x, T *y, size_t lanes,              \
             size_t frames, int layout)                                                            \
    {                                                                                              \
        int rc = check_common(cfg, n, state, x, y, lanes, frames, layout);                         \
        if (rc) return rc;                                                                         \
        if (!(FRAC_CHECK)) return IDSP_EINVAL;                                                     \
        return NAME##_range(cfg, n, state, x, y, lanes, frames, layout, 0, lanes);                 \
    }

LANE_DRIVER(idsp_ref_biquad_i32_df1, idsp_biquad_i32, int32_t, 4, frac_ok_i32(cfg, n),
            df1_i32(c->ba, c->frac, s, x0))
LANE_DRIVER(idsp_ref_biquad_i32_df1_clamp, idsp_biquad_clamp_i32, int32_t, 4, frac_ok_clamp(cfg, n),
            df1_clamp_i32(c, s, x0))
LANE_DRIVER(idsp_ref_biquad_i32_dither, idsp_biquad_i32, int32_t, 5, frac_ok_i32(cfg, n),
            dither_i32(c->ba, c->frac, s, x0))
LANE_DRIVER(idsp_ref_biquad_i32_dither_clamp, idsp_biquad_clamp_i32, int32_t, 5, frac_ok_clamp(cfg, n),
            dither_clamp_i32(c, s, x0))
LANE_DRIVER(idsp_ref_biquad_i32_wide, idsp_biquad_i32, int32_t, 6, frac_ok_i32(cfg, n),
            wide_i32(c->ba, c->frac, s, x0))
LANE_DRIVER(idsp_ref_biquad_i32_wide_clamp, idsp_biquad_clamp_i32, int32_t, 6, frac_ok_clamp(cfg, n),
            wide_clamp_i32(c, s, x0))
LANE_DRIVER(idsp_ref_biquad_f32_df1, idsp_biquad_f32, float, 4, 1, df1_f32(c->ba, s, x0))
LANE_DRIVER(idsp_ref_biquad_f32_df1_clamp, idsp_biquad_clamp_f32, float, 4, 1, df1_clamp_f32(c, s, x0))
LANE_DRIVER(idsp_ref_biquad_f32_df2t, idsp_biquad_f32, float, 2, 1, df2t_f32(c->ba, s, x0))
LANE_DRIVER(idsp_ref_biquad_f32_df2t_clamp, idsp_biquad_clamp_f32, float, 2, 1, df2t_clamp_f32(c, s, x0))
LANE_DRIVER(idsp_ref_biquad_f64_df1, idsp_biquad_f64, double, 8, 1, df1_f64(c->ba, s, x0))
LANE_DRIVER(idsp_ref_biquad_f64_df1_clamp, idsp_biquad_clamp_f64, double, 8, 1, df1_clamp_f64(c, s, x0))
LANE_DRIVER(idsp_ref_biquad_f64_df2t, idsp_biquad_f64, double, 4, 1, df2t_f64(c->ba, s, x0))
LANE_DRIVER(idsp_ref_biquad_f64_df2t_clamp, idsp_biquad_clamp_f64, double, 4, 1, df2t_clamp_f64(c, s, x0))

/*
 * `ByLane<[C; N]>::process_view` (dsp-process/src/compose.rs:375-389): lane i is
 * filtered by configuration i with state i.  Coefficient planes:
 * coef[(k * CV + v) * lanes + l], CV = 5 (ba) or 8 (ba, u, min, max).
 */
#define FILL_PLAIN(c, k, l) do { for (int v = 0; v < 5; v++) (c).ba[v] = coef[((k) * 5 + v) * lanes + (l)]; } while (0)
#define FILL_CLAMP(c, k, l) do { for (int v = 0; v < 5; v++) (c).ba[v] = coef[((k) * 8 + v) * lanes + (l)]; \
        (c).u = coef[((k) * 8 + 5) * lanes + (l)]; (c).min = coef[((k) * 8 + 6) * lanes + (l)];             \
        (c).max = coef[((k) * 8 + 7) * lanes + (l)]; } while (0)
#define BYLANE_BODY(BASE, CFG_T, FILL, SETFRAC)                                                    \
    {                                                                                              \
        int rc = check_common(coef, n, state, x, y, lanes, frames, layout);                        \
        if (rc) return rc;                                                                         \
        CFG_T cfg[IDSP_MAX_SECTIONS];                                                              \
        for (size_t l = 0; l < lanes; l++) {                                                       \
            for (size_t k = 0; k < n; k++) { FILL(cfg[k], k, l); SETFRAC; }                        \
            rc = BASE##_range(cfg, n, state, x, y, lanes, frames, layout, l, l + 1);               \
            if (rc) return rc;                                                                     \
        }                                                                                          \
        return IDSP_OK;                                                                            \
    }
#define BYLANE_I32(NAME, CFG_T, FILL)                                                              \
    int idsp_ref_biquad_i32_##NAME##_bylane(const int32_t *coef, int frac, size_t n, void *state,  \
            const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout)                 \
    {                                                                                              \
        if (frac < 0 || frac > 31) return IDSP_EINVAL;                                             \
        BYLANE_BODY(idsp_ref_biquad_i32_##NAME, CFG_T, FILL, cfg[k].frac = frac)                   \
    }
#define BYLANE_F(T, TN, NAME, CFG_T, FILL)                                                         \
    int idsp_ref_biquad_##TN##_##NAME##_bylane(const T *coef, size_t n, void *state, const T *x,   \
            T *y, size_t lanes, size_t frames, int layout)                                         \
    BYLANE_BODY(idsp_ref_biquad_##TN##_##NAME, CFG_T, FILL, (void)0)

BYLANE_I32(df1, idsp_biquad_i32, FILL_PLAIN)
BYLANE_I32(df1_clamp, idsp_biquad_clamp_i32, FILL_CLAMP)
BYLANE_I32(dither, idsp_biquad_i32, FILL_PLAIN)
BYLANE_I32(dither_clamp, idsp_biquad_clamp_i32, FILL_CLAMP)
BYLANE_I32(wide, idsp_biquad_i32, FILL_PLAIN)
BYLANE_I32(wide_clamp, idsp_biquad_clamp_i32, FILL_CLAMP)
BYLANE_F(float, f32, df1, idsp_biquad_f32, FILL_PLAIN)
BYLANE_F(float, f32, df1_clamp, idsp_biquad_clamp_f32, FILL_CLAMP)
BYLANE_F(float, f32, df2t, idsp_biquad_f32, FILL_PLAIN)
BYLANE_F(float, f32, df2t_clamp, idsp_biquad_clamp_f32, FILL_CLAMP)
BYLANE_F(double, f64, df1, idsp_biquad_f64, FILL_PLAIN)
BYLANE_F(double, f64, df1_clamp, idsp_biquad_clamp_f64, FILL_CLAMP)
BYLANE_F(double, f64, df2t, idsp_biquad_f64, FILL_PLAIN)
BYLANE_F(double, f64, df2t_clamp, idsp_biquad_clamp_f64, FILL_CLAMP)

/* `Cascade<[Biquad<C>; N]>` x `DirectForm<T, N>` (src/iir/biquad.rs:339-364):
 * sample-major fold over sections; section k's input history is the output
 * history of section k-1 (x for k = 0).  Words: x0,x1,(y0,y1) x n. */
int idsp_ref_cascade_i32_df1(const idsp_biquad_i32 *cfg, size_t n, void *state, const int32_t *x,
                             int32_t *y, size_t lanes, size_t frames, int layout)
{
    int rc = check_common(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if (n < 1 || n > 8 || !frac_ok_i32(cfg, n)) return IDSP_EINVAL;
    uint32_t *st = (uint32_t *)state;
    size_t W = 2 + 2 * n;
    for (size_t l = 0; l < lanes; l++) {
        int32_t s[18];
        for (size_t w = 0; w < W; w++) s[w] = (int32_t)st[w * lanes + l];
        for (size_t f = 0; f < frames; f++) {
            size_t i = idx_of(f, l, lanes, frames, layout);
            int32_t x0 = x[i];
            int32_t *xh = &s[0]; /* fold accumulator `(x0, &mut x)` */
            for (size_t k = 0; k < n; k++) {
                int32_t *yh = &s[2 + 2 * k];
                int32_t y0 = trunc32(sum5_i32(cfg[k].ba, x0, xh[0], xh[1], yh[0], yh[1]) >> cfg[k].frac);
                xh[1] = xh[0]; xh[0] = x0; /* *x = [x0, x[0]] */
                x0 = y0; xh = yh;
            }
            xh[1] = xh[0]; xh[0] = x0; /* *y = [y0, y[0]] */
            y[i] = x0;
        }
        for (size_t w = 0; w < W; w++) st[w * lanes + l] = (uint32_t)s[w];
    }
    return IDSP_OK;
}

int idsp_ref_cascade_f32_df1(const idsp_biquad_f32 *cfg, size_t n, void *state, const float *x,
                             float *y, size_t lanes, size_t frames, int layout)
{
    int rc = check_common(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if (n < 1 || n > 8) return IDSP_EINVAL;
    uint32_t *st = (uint32_t *)state;
    size_t W = 2 + 2 * n;
    for (size_t l = 0; l < lanes; l++) {
        float s[18];
        for (size_t w = 0; w < W; w++) s[w] = f32_from_bits(st[w * lanes + l]);
        for (size_t f = 0; f < frames; f++) {
            size_t i = idx_of(f, l, lanes, frames, layout);
            float x0 = x[i];
            float *xh = &s[0];
            for (size_t k = 0; k < n; k++) {
                float *yh = &s[2 + 2 * k];
                const float *ba = cfg[k].ba;
                float acc = ba[0] * x0;
                acc = acc + ba[1] * xh[0];
                acc = acc + ba[2] * xh[1];
                acc = acc + ba[3] * yh[0];
                acc = acc + ba[4] * yh[1];
                xh[1] = xh[0]; xh[0] = x0;
                x0 = acc; xh = yh;
            }
            xh[1] = xh[0]; xh[0] = x0;
            y[i] = x0;
        }
        for (size_t w = 0; w < W; w++) st[w * lanes + l] = f32_to_bits(s[w]);
    }
    return IDSP_OK;
}

int idsp_ref_cascade_f64_df1(const idsp_biquad_f64 *cfg, size_t n, void *state, const double *x,
                             double *y, size_t lanes, size_t frames, int layout)
{
    int rc = check_common(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if (n < 1 || n > 8) return IDSP_EINVAL;
    uint32_t *st = (uint32_t *)state;
    size_t V = 2 + 2 * n;
    for (size_t l = 0; l < lanes; l++) {
        double s[18];
        for (size_t v = 0; v < V; v++) {
            uint32_t w[2] = {st[(2 * v) * lanes + l], st[(2 * v + 1) * lanes + l]};
            s[v] = f64_from_words(w, 0);
        }
        for (size_t f = 0; f < frames; f++) {
            size_t i = idx_of(f, l, lanes, frames, layout);
            double x0 = x[i];
            double *xh = &s[0];
            for (size_t k = 0; k < n; k++) {
                double *yh = &s[2 + 2 * k];
                const double *ba = cfg[k].ba;
                double acc = ba[0] * x0;
                acc = acc + ba[1] * xh[0];
                acc = acc + ba[2] * xh[1];
                acc = acc + ba[3] * yh[0];
                acc = acc + ba[4] * yh[1];
                xh[1] = xh[0]; xh[0] = x0;
                x0 = acc; xh = yh;
            }
            xh[1] = xh[0]; xh[0] = x0;
            y[i] = x0;
        }
        for (size_t v = 0; v < V; v++) {
            uint32_t w[2];
            f64_to_words(w, 0, s[v]);
            st[(2 * v) * lanes + l] = w[0];
            st[(2 * v + 1) * lanes + l] = w[1];
        }
    }
    return IDSP_OK;
}

/* --------------------------------------------------- threaded CPU baseline */

/*
 * FRAME_MAJOR traversal: the reference's `Lanes<C>: SplitProcess<[X; N], [Y; N], [S; N]>`
 * (dsp-process/src/compose.rs:468-476) under the default `block()` loop of process.rs:122-127 —
 * frames outer, the lanes of a frame inner, state[i] indexed by lane.  Per sample the sections run
 * as the serial slice composition's `process()` (compose.rs:43-60).  Results equal the lane-outer
 * drivers above (lanes never interact; each section sees the same input sequence either way).
 */
/* The block's state is held as the reference holds it — one record per lane, `[S; N]` — and moved
 * from / to the ABI's word planes at entry / exit. */
static uint32_t *fm_state_in(const uint32_t *st, size_t words, size_t lanes, size_t l0, size_t L)
{
    uint32_t *s = (uint32_t *)malloc((L ? L : 1) * words * sizeof(uint32_t));
    if (!s) return NULL;
    for (size_t l = 0; l < L; l++) for (size_t w = 0; w < words; w++) s[l * words + w] = st[w * lanes + l0 + l];
    return s;
}
static void fm_state_out(uint32_t *st, uint32_t *s, size_t words, size_t lanes, size_t l0, size_t L)
{
    for (size_t l = 0; l < L; l++) for (size_t w = 0; w < words; w++) st[w * lanes + l0 + l] = s[l * words + w];
    free(s);
}

static int fm_rows_i32_df1(const idsp_biquad_i32 *cfg, size_t n, uint32_t *st, const int32_t *x, int32_t *y,
                           size_t lanes, size_t frames, size_t l0, size_t l1)
{
    const size_t L = l1 - l0, W = 4 * n;
    uint32_t *rec = fm_state_in(st, W, lanes, l0, L);
    if (!rec) return IDSP_EINVAL;
    for (size_t f = 0; f < frames; f++) {
        const int32_t *xr = x + f * lanes + l0;
        int32_t *yr = y + f * lanes + l0;
        for (size_t l = 0; l < L; l++) {
            int32_t v = xr[l];
            for (size_t k = 0; k < n; k++) v = df1_i32(cfg[k].ba, cfg[k].frac, rec + l * W + 4 * k, v);
            yr[l] = v;
        }
    }
    fm_state_out(st, rec, W, lanes, l0, L);
    return IDSP_OK;
}

static int fm_rows_f32_df2t(const idsp_biquad_f32 *cfg, size_t n, uint32_t *st, const float *x, float *y,
                            size_t lanes, size_t frames, size_t l0, size_t l1)
{
    const size_t L = l1 - l0, W = 2 * n;
    uint32_t *rec = fm_state_in(st, W, lanes, l0, L);
    if (!rec) return IDSP_EINVAL;
    for (size_t f = 0; f < frames; f++) {
        const float *xr = x + f * lanes + l0;
        float *yr = y + f * lanes + l0;
        for (size_t l = 0; l < L; l++) {
            float v = xr[l];
            for (size_t k = 0; k < n; k++) v = df2t_f32(cfg[k].ba, rec + l * W + 2 * k, v);
            yr[l] = v;
        }
    }
    fm_state_out(st, rec, W, lanes, l0, L);
    return IDSP_OK;
}

typedef struct {
    int kind; const void *cfg; size_t n; void *state; const void *x; void *y;
    size_t lanes, frames; int layout; size_t l0, l1; int reps; int cpu;
    int rc; /* first non-zero status of this block's passes (a failed scratch allocation in the FRAME_MAJOR drivers) */
} mt_job;

static void mt_run_block(mt_job *j)
{
    j->rc = IDSP_OK;
    for (int r = 0; r < j->reps && j->rc == IDSP_OK; r++) {
        if (j->kind == 0) {
            if (j->layout == IDSP_FRAME_MAJOR)
                j->rc = fm_rows_i32_df1((const idsp_biquad_i32 *)j->cfg, j->n, (uint32_t *)j->state, (const int32_t *)j->x,
                                (int32_t *)j->y, j->lanes, j->frames, j->l0, j->l1);
            else
                j->rc = idsp_ref_biquad_i32_df1_range((const idsp_biquad_i32 *)j->cfg, j->n, j->state, (const int32_t *)j->x,
                                              (int32_t *)j->y, j->lanes, j->frames, j->layout, j->l0, j->l1);
        } else {
            if (j->layout == IDSP_FRAME_MAJOR)
                j->rc = fm_rows_f32_df2t((const idsp_biquad_f32 *)j->cfg, j->n, (uint32_t *)j->state, (const float *)j->x,
                                 (float *)j->y, j->lanes, j->frames, j->l0, j->l1);
            else
                j->rc = idsp_ref_biquad_f32_df2t_range((const idsp_biquad_f32 *)j->cfg, j->n, j->state, (const float *)j->x,
                                               (float *)j->y, j->lanes, j->frames, j->layout, j->l0, j->l1);
        }
    }
}

static void *mt_worker(void *p)
{
    mt_job *j = (mt_job *)p;
    if (j->cpu >= 0) {               /* one thread per allowed CPU: keep it there */
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(j->cpu, &one);
        pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
    }
    mt_run_block(j);
    return NULL;
}

/* Number of CPUs this process may run on (sched_getaffinity; what `nproc` prints). */
int idsp_ref_host_cpus(void)
{
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) != 0) return 1;
    int c = CPU_COUNT(&set);
    return c < 1 ? 1 : c;
}

/*
 * kind 0 = idsp_ref_biquad_i32_df1 (cfg: idsp_biquad_i32), kind 1 = idsp_ref_biquad_f32_df2t (cfg:
 * idsp_biquad_f32).  The lanes are cut into `threads` contiguous blocks, one POSIX thread each, pinned
 * to the allowed CPUs in order when threads > 1; every thread runs its block `reps` times back to back
 * (state carried from pass to pass like consecutive block() calls) so that thread start-up is paid once
 * per timed region, not once per pass.  LANE_MAJOR blocks take the lane-outer driver (compose.rs:478-494),
 * FRAME_MAJOR blocks the frame-outer one above.  threads == 1 runs on the calling thread.
 */
int idsp_ref_biquad_mt_reps(int kind, const void *cfg, size_t n, void *state, const void *x, void *y,
                            size_t lanes, size_t frames, int layout, int threads, int reps)
{
    int rc = check_common(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if (kind < 0 || kind > 1 || threads < 1 || threads > 4096 || reps < 1) return IDSP_EINVAL;
    if (kind == 0 && !frac_ok_i32((const idsp_biquad_i32 *)cfg, n)) return IDSP_EINVAL;
    if (threads == 1) {
        mt_job j = {kind, cfg, n, state, x, y, lanes, frames, layout, 0, lanes, reps, -1, IDSP_OK};
        mt_run_block(&j);
        return j.rc;
    }
    cpu_set_t allowed;
    int ncpu = 0, cpus[CPU_SETSIZE];
    if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0)
        for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &allowed)) cpus[ncpu++] = c;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * (size_t)threads);
    unsigned char *started = (unsigned char *)calloc((size_t)threads, 1);
    if (!tid || !jobs || !started) { free(tid); free(jobs); free(started); return IDSP_EINVAL; }
    for (int t = 0; t < threads; t++) {
        mt_job j = {kind, cfg, n, state, x, y, lanes, frames, layout,
                    lanes * (size_t)t / (size_t)threads, lanes * (size_t)(t + 1) / (size_t)threads, reps,
                    ncpu > 0 ? cpus[t % ncpu] : -1, IDSP_OK};
        jobs[t] = j;
        started[t] = pthread_create(&tid[t], NULL, mt_worker, &jobs[t]) == 0;
        if (!started[t]) mt_run_block(&jobs[t]);     /* could not start a thread: do the block here */
    }
    for (int t = 0; t < threads; t++) if (started[t]) pthread_join(tid[t], NULL);
    rc = IDSP_OK;
    for (int t = 0; t < threads && rc == IDSP_OK; t++) rc = jobs[t].rc;  /* a block that failed must not read as a result */
    free(tid); free(jobs); free(started);
    return rc;
}

int idsp_ref_biquad_i32_df1_mt(const idsp_biquad_i32 *cfg, size_t n, void *state, const int32_t *x,
                               int32_t *y, size_t lanes, size_t frames, int layout, int threads)
{
    return idsp_ref_biquad_mt_reps(0, cfg, n, state, x, y, lanes, frames, layout, threads, 1);
}

int idsp_ref_biquad_f32_df2t_mt(const idsp_biquad_f32 *cfg, size_t n, void *state, const float *x,
                                float *y, size_t lanes, size_t frames, int layout, int threads)
{
    return idsp_ref_biquad_mt_reps(1, cfg, n, state, x, y, lanes, frames, layout, threads, 1);
}

/* ------------------------------------------------------------------- hbf */

/* src/hbf.rs:308-349 (HBF_TAPS) and :258-292 (HBF_TAPS_98); index = tuple index. */
static const int HBF_M[2][5] = {{23, 10, 5, 4, 3}, {15, 6, 3, 3, 2}};
static const float HBF_TAPS_TBL[2][5][23] = {
    {
        {7.60375795e-07f, -3.77494111e-06f, 1.26458559e-05f, -3.43188253e-05f, 8.10687478e-05f,
         -1.72971467e-04f, 3.40845059e-04f, -6.29522864e-04f, 1.10128831e-03f, -1.83933299e-03f,
         2.95124926e-03f, -4.57290964e-03f, 6.87374176e-03f, -1.00656257e-02f, 1.44199840e-02f,
         -2.03025100e-02f, 2.82462332e-02f, -3.91128509e-02f, 5.44795658e-02f, -7.77002672e-02f,
         1.17523452e-01f, -2.06185388e-01f, 6.34588695e-01f},
        {-1.12811343e-05f, 1.12724671e-04f, -6.07439343e-04f, 2.31904511e-03f, -7.00322950e-03f,
         1.78225473e-02f, -4.01209836e-02f, 8.43315989e-02f, -1.83189521e-01f, 6.26346521e-01f},
        {0.0007686f, -0.00768669f, 0.0386536f, -0.14002434f, 0.60828885f},
        {-0.00261331f, 0.02476858f, -0.12112638f, 0.59897111f},
        {0.01186105f, -0.09808109f, 0.58622005f},
    },
    {
        {7.02144012e-05f, -2.43279582e-04f, 6.35026936e-04f, -1.39782541e-03f, 2.74613582e-03f,
         -4.96403839e-03f, 8.41806912e-03f, -1.35827601e-02f, 2.11004053e-02f, -3.19267647e-02f,
         4.77024289e-02f, -7.18014345e-02f, 1.12942004e-01f, -2.03279594e-01f, 6.33592923e-01f},
        {-0.00086943f, 0.00577837f, -0.02201674f, 0.06357869f, -0.16627679f, 0.61979312f},
        {0.01414651f, -0.10439639f, 0.59026742f},
        {0.01227974f, -0.09930782f, 0.58702834f},
        {-0.06291796f, 0.5629161f},
    },
};

static int hbf_fill(int tap_set, int stages, int dec, idsp_hbf_cascade_f32 *out)
{
    if (!out || tap_set < 0 || tap_set > 1 || stages < 1 || stages > 5) return IDSP_EINVAL;
    memset(out, 0, sizeof(*out));
    out->stages = stages;
    for (int s = 0; s < stages; s++) {
        /* decimator: HBF_DEC_CASCADE nests tuple index stages-1 (highest rate) first,
         * src/hbf.rs:412-421; interpolator: index 0 first, src/hbf.rs:503-512 */
        int t = dec ? stages - 1 - s : s;
        out->m[s] = HBF_M[tap_set][t];
        for (int k = 0; k < out->m[s]; k++) out->taps[s][k] = HBF_TAPS_TBL[tap_set][t][k];
    }
    return IDSP_OK;
}

int idsp_ref_hbf_dec_cascade(int tap_set, int stages, idsp_hbf_cascade_f32 *out) { return hbf_fill(tap_set, stages, 1, out); }
int idsp_ref_hbf_int_cascade(int tap_set, int stages, idsp_hbf_cascade_f32 *out) { return hbf_fill(tap_set, stages, 0, out); }

static int hbf_cfg_ok(const idsp_hbf_cascade_f32 *c)
{
    if (!c || c->stages < 1 || c->stages > IDSP_HBF_MAX_STAGES) return 0;
    for (int s = 0; s < c->stages; s++) if (c->m[s] < 1 || c->m[s] > IDSP_HBF_MAX_TAPS) return 0;
    return 1;
}

/* src/hbf.rs:424-448: from the highest-rate stage down: n = n/2 + LEN(stage),
 * LEN = 2M-1 (src/hbf.rs:76-78 `len()`). */
int idsp_ref_hbf_dec_response_length(const idsp_hbf_cascade_f32 *c)
{
    if (!hbf_cfg_ok(c)) return IDSP_EINVAL;
    int n = 0;
    for (int s = 0; s < c->stages; s++) { n /= 2; n += 2 * c->m[s] - 1; }
    return n;
}

/* src/hbf.rs:515-539: n = (n + LEN(stage)) * 2 from the lowest-rate stage up. */
int idsp_ref_hbf_int_response_length(const idsp_hbf_cascade_f32 *c)
{
    if (!hbf_cfg_ok(c)) return IDSP_EINVAL;
    int n = 0;
    for (int s = 0; s < c->stages; s++) { n += 2 * c->m[s] - 1; n *= 2; }
    return n;
}

size_t idsp_ref_hbf_dec_state_words(const idsp_hbf_cascade_f32 *c)
{
    if (!hbf_cfg_ok(c)) return 0;
    size_t w = 0;
    for (int s = 0; s < c->stages; s++) w += (size_t)(3 * c->m[s] - 2);
    return w;
}

size_t idsp_ref_hbf_int_state_words(const idsp_hbf_cascade_f32 *c)
{
    if (!hbf_cfg_ok(c)) return 0;
    size_t w = 0;
    for (int s = 0; s < c->stages; s++) w += (size_t)(2 * c->m[s] - 1);
    return w;
}

/* src/hbf.rs:46-68 `get::<_,_,M,false,true>` for ONE window w[0..2M):
 * old = w[..M], new = w[M..] reversed; Σ (new + old) * tap, accumulated by
 * `f32::sum` — a sequential fold seeded with -0.0 (Rust >= 1.83; edition 2024
 * implies >= 1.85). */
static inline float hbf_get(const float *taps, int m, const float *w)
{
    float acc = -0.0f;
    for (int k = 0; k < m; k++) acc = acc + (w[2 * m - 1 - k] + w[k]) * taps[k];
    return acc;
}

#define HBF_BLOCK 32 /* any block size gives the same result (src/hbf.rs:166) */

/* One `HbfDec` stage over n output samples; in = n pairs [even, odd]
 * (src/hbf.rs:163-185), literal including the block loop and copy_within.
 * st = even[m-1] ++ odd[2m-1]. */
static void hbf_dec_stage(const float *taps, int m, float *st, const float *in, float *out, size_t n)
{
    int len = 2 * m - 1;
    float even[IDSP_HBF_MAX_TAPS - 1 + HBF_BLOCK], odd[2 * IDSP_HBF_MAX_TAPS - 1 + HBF_BLOCK];
    memcpy(even, st, sizeof(float) * (size_t)(m - 1));
    memcpy(odd, st + (m - 1), sizeof(float) * (size_t)len);
    for (size_t p = 0; p < n; p += HBF_BLOCK) {
        size_t c = n - p < HBF_BLOCK ? n - p : HBF_BLOCK;
        for (size_t i = 0; i < c; i++) {
            even[(size_t)(m - 1) + i] = in[2 * (p + i)];
            odd[(size_t)len + i] = in[2 * (p + i) + 1];
        }
        for (size_t i = 0; i < c; i++) out[p + i] = hbf_get(taps, m, odd + i) + even[i];
        memmove(even, even + c, sizeof(float) * (size_t)(m - 1));
        memmove(odd, odd + c, sizeof(float) * (size_t)len);
    }
    memcpy(st, even, sizeof(float) * (size_t)(m - 1));
    memcpy(st + (m - 1), odd, sizeof(float) * (size_t)len);
}

/* One `HbfInt` stage over n input samples -> n pairs (src/hbf.rs:207-227). st = x[2m-1]. */
static void hbf_int_stage(const float *taps, int m, float *st, const float *in, float *out, size_t n)
{
    int len = 2 * m - 1;
    float xb[2 * IDSP_HBF_MAX_TAPS - 1 + HBF_BLOCK];
    memcpy(xb, st, sizeof(float) * (size_t)len);
    for (size_t p = 0; p < n; p += HBF_BLOCK) {
        size_t c = n - p < HBF_BLOCK ? n - p : HBF_BLOCK;
        memcpy(xb + len, in + p, sizeof(float) * c);
        for (size_t i = 0; i < c; i++) {
            out[2 * (p + i)] = hbf_get(taps, m, xb + i); /* interpolated */
            out[2 * (p + i) + 1] = xb[(size_t)m + i];    /* centre tap: identity */
        }
        memmove(xb, xb + c, sizeof(float) * (size_t)len);
    }
    memcpy(st, xb, sizeof(float) * (size_t)len);
}

/* Cascade per lane, whole-buffer stage-major.  The reference runs the same
 * stages chunk-wise through `Major` scratch (dsp-process/src/compose.rs:581-593)
 * with `ChunkIn<_,2>` regrouping (adapters.rs:333-339); each stage is a causal
 * streaming operator, so chunking does not change any value. */
int idsp_ref_hbf_dec_f32(const idsp_hbf_cascade_f32 *cfg, void *state, const float *x, float *y,
                         size_t lanes, size_t frames, int layout)
{
    if (!hbf_cfg_ok(cfg) || (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR)) return IDSP_EINVAL;
    if (lanes && (!state || (frames && (!x || !y)))) return IDSP_EINVAL;
    size_t R = (size_t)1 << cfg->stages, W = idsp_ref_hbf_dec_state_words(cfg);
    uint32_t *st = (uint32_t *)state;
    float *a = (float *)malloc(sizeof(float) * (frames * R + 1));
    float *b = (float *)malloc(sizeof(float) * (frames * R / 2 + 1));
    float *ls = (float *)malloc(sizeof(float) * W);
    if (!a || !b || !ls) { free(a); free(b); free(ls); return IDSP_EINVAL; }
    for (size_t l = 0; l < lanes; l++) {
        for (size_t w = 0; w < W; w++) ls[w] = f32_from_bits(st[w * lanes + l]);
        for (size_t f = 0; f < frames; f++)
            memcpy(a + f * R, x + idx_of(f, l, lanes, frames, layout) * R, sizeof(float) * R);
        size_t n = frames * R, off = 0;
        float *src = a, *dst = b;
        for (int s = 0; s < cfg->stages; s++) {
            n /= 2;
            hbf_dec_stage(cfg->taps[s], cfg->m[s], ls + off, src, dst, n);
            off += (size_t)(3 * cfg->m[s] - 2);
            float *t = src; src = dst; dst = t;
        }
        for (size_t f = 0; f < frames; f++) y[idx_of(f, l, lanes, frames, layout)] = src[f];
        for (size_t w = 0; w < W; w++) st[w * lanes + l] = f32_to_bits(ls[w]);
    }
    free(a); free(b); free(ls);
    return IDSP_OK;
}

int idsp_ref_hbf_int_f32(const idsp_hbf_cascade_f32 *cfg, void *state, const float *x, float *y,
                         size_t lanes, size_t frames, int layout)
{
    if (!hbf_cfg_ok(cfg) || (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR)) return IDSP_EINVAL;
    if (lanes && (!state || (frames && (!x || !y)))) return IDSP_EINVAL;
    size_t R = (size_t)1 << cfg->stages, W = idsp_ref_hbf_int_state_words(cfg);
    uint32_t *st = (uint32_t *)state;
    float *a = (float *)malloc(sizeof(float) * (frames * R + 1));
    float *b = (float *)malloc(sizeof(float) * (frames * R + 1));
    float *ls = (float *)malloc(sizeof(float) * W);
    if (!a || !b || !ls) { free(a); free(b); free(ls); return IDSP_EINVAL; }
    for (size_t l = 0; l < lanes; l++) {
        for (size_t w = 0; w < W; w++) ls[w] = f32_from_bits(st[w * lanes + l]);
        for (size_t f = 0; f < frames; f++) a[f] = x[idx_of(f, l, lanes, frames, layout)];
        size_t n = frames, off = 0;
        float *src = a, *dst = b;
        for (int s = 0; s < cfg->stages; s++) {
            hbf_int_stage(cfg->taps[s], cfg->m[s], ls + off, src, dst, n);
            n *= 2;
            off += (size_t)(2 * cfg->m[s] - 1);
            float *t = src; src = dst; dst = t;
        }
        for (size_t f = 0; f < frames; f++)
            memcpy(y + idx_of(f, l, lanes, frames, layout) * R, src + f * R, sizeof(float) * R);
        for (size_t w = 0; w < W; w++) st[w * lanes + l] = f32_to_bits(ls[w]);
    }
    free(a); free(b); free(ls);
    return IDSP_OK;
}

/* `type_fir!` same-rate FIR (src/hbf.rs:70-138) with `get::<_,_,M,ODD,SYM>` (:46-68):
 * literal block loop incl. copy_within; st = last LEN inputs. */
size_t idsp_ref_fir_sym_state_words(const idsp_fir_sym_f32 *c)
{
    if (!c || c->kind < 0 || c->kind > 3 || c->m < 1 || c->m > IDSP_HBF_MAX_TAPS) return 0;
    int odd = (c->kind == IDSP_FIR_ODD_SYMMETRIC || c->kind == IDSP_FIR_ODD_ANTISYMMETRIC);
    return (size_t)(2 * c->m - 1 + odd);
}

int idsp_ref_fir_sym_f32_process(const idsp_fir_sym_f32 *c, void *state, const float *x, float *y,
                                 size_t lanes, size_t frames, int layout)
{
    size_t len = idsp_ref_fir_sym_state_words(c);
    if (!len || (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR)) return IDSP_EINVAL;
    if (lanes && (!state || (frames && (!x || !y)))) return IDSP_EINVAL;
    int m = c->m;
    int odd = (c->kind == IDSP_FIR_ODD_SYMMETRIC || c->kind == IDSP_FIR_ODD_ANTISYMMETRIC);
    int sym = (c->kind == IDSP_FIR_ODD_SYMMETRIC || c->kind == IDSP_FIR_EVEN_SYMMETRIC);
    uint32_t *st = (uint32_t *)state;
    for (size_t l = 0; l < lanes; l++) {
        float buf[2 * IDSP_HBF_MAX_TAPS + HBF_BLOCK];
        for (size_t w = 0; w < len; w++) buf[w] = f32_from_bits(st[w * lanes + l]);
        for (size_t p = 0; p < frames; p += HBF_BLOCK) {
            size_t n = frames - p < HBF_BLOCK ? frames - p : HBF_BLOCK;
            for (size_t i = 0; i < n; i++) buf[len + i] = x[idx_of(p + i, l, lanes, frames, layout)];
            for (size_t i = 0; i < n; i++) {
                const float *w = buf + i; /* window of 2M + odd samples */
                float acc = -0.0f;
                for (int k = 0; k < m; k++) {
                    float nw = w[2 * m - 1 + odd - k], od = w[k];
                    acc = acc + (sym ? nw + od : nw - od) * c->taps[k];
                }
                y[idx_of(p + i, l, lanes, frames, layout)] = (odd && sym) ? acc + w[m] : acc;
            }
            memmove(buf, buf + n, sizeof(float) * len);
        }
        for (size_t w = 0; w < len; w++) st[w * lanes + l] = f32_to_bits(buf[w]);
    }
    return IDSP_OK;
}

/* ---- the same half-band / FIR processors on f64 (`EvenSymmetric<[f64; M]>`, `T = f64`: the reference types are generic in
 * the sample type, src/hbf.rs:70-138,142-236).  Restated, not macro-generated from the f32 code above: `f64::sum` folds
 * from -0.0 like `f32::sum`.  State value v occupies words 2v (low) and 2v + 1 (high). */
static int hbf_cfg_ok_f64(const idsp_hbf_cascade_f64 *c)
{
    if (!c || c->stages < 1 || c->stages > IDSP_HBF_MAX_STAGES) return 0;
    for (int s = 0; s < c->stages; s++) if (c->m[s] < 1 || c->m[s] > IDSP_HBF_MAX_TAPS) return 0;
    return 1;
}

static int hbf_fill_f64(int tap_set, int stages, int dec, idsp_hbf_cascade_f64 *out)
{
    idsp_hbf_cascade_f32 f;
    int rc = hbf_fill(tap_set, stages, dec, &f);
    if (rc || !out) return IDSP_EINVAL;
    memset(out, 0, sizeof(*out));
    out->stages = f.stages;
    for (int s = 0; s < f.stages; s++) {
        out->m[s] = f.m[s];
        for (int k = 0; k < f.m[s]; k++) out->taps[s][k] = (double)f.taps[s][k];
    }
    return IDSP_OK;
}
int idsp_ref_hbf_dec_cascade_f64(int tap_set, int stages, idsp_hbf_cascade_f64 *out) { return hbf_fill_f64(tap_set, stages, 1, out); }
int idsp_ref_hbf_int_cascade_f64(int tap_set, int stages, idsp_hbf_cascade_f64 *out) { return hbf_fill_f64(tap_set, stages, 0, out); }

size_t idsp_ref_hbf_dec_state_words_f64(const idsp_hbf_cascade_f64 *c)
{
    if (!hbf_cfg_ok_f64(c)) return 0;
    size_t w = 0;
    for (int s = 0; s < c->stages; s++) w += (size_t)(3 * c->m[s] - 2);
    return 2 * w;
}
size_t idsp_ref_hbf_int_state_words_f64(const idsp_hbf_cascade_f64 *c)
{
    if (!hbf_cfg_ok_f64(c)) return 0;
    size_t w = 0;
    for (int s = 0; s < c->stages; s++) w += (size_t)(2 * c->m[s] - 1);
    return 2 * w;
}
size_t idsp_ref_fir_sym_state_words_f64(const idsp_fir_sym_f64 *c)
{
    if (!c || c->kind < 0 || c->kind > 3 || c->m < 1 || c->m > IDSP_HBF_MAX_TAPS) return 0;
    int odd = (c->kind == IDSP_FIR_ODD_SYMMETRIC || c->kind == IDSP_FIR_ODD_ANTISYMMETRIC);
    return (size_t)(2 * (2 * c->m - 1 + odd));
}

static inline double plane_f64(const uint32_t *st, size_t v, size_t lanes, size_t l)
{
    uint64_t u = ((uint64_t)st[(2 * v + 1) * lanes + l] << 32) | st[(2 * v) * lanes + l];
    double d; memcpy(&d, &u, 8); return d;
}
static inline void plane_f64_put(uint32_t *st, size_t v, size_t lanes, size_t l, double d)
{
    uint64_t u; memcpy(&u, &d, 8);
    st[(2 * v) * lanes + l] = (uint32_t)u;
    st[(2 * v + 1) * lanes + l] = (uint32_t)(u >> 32);
}

/* src/hbf.rs:46-68 on f64 */
static inline double hbf_get_f64(const double *taps, int m, const double *w)
{
    double acc = -0.0;
    for (int k = 0; k < m; k++) acc = acc + (w[2 * m - 1 - k] + w[k]) * taps[k];
    return acc;
}

/* src/hbf.rs:163-185 */
static void hbf_dec_stage_f64(const double *taps, int m, double *st, const double *in, double *out, size_t n)
{
    int len = 2 * m - 1;
    double even[IDSP_HBF_MAX_TAPS - 1 + HBF_BLOCK], odd[2 * IDSP_HBF_MAX_TAPS - 1 + HBF_BLOCK];
    memcpy(even, st, sizeof(double) * (size_t)(m - 1));
    memcpy(odd, st + (m - 1), sizeof(double) * (size_t)len);
    for (size_t p = 0; p < n; p += HBF_BLOCK) {
        size_t c = n - p < HBF_BLOCK ? n - p : HBF_BLOCK;
        for (size_t i = 0; i < c; i++) {
            even[(size_t)(m - 1) + i] = in[2 * (p + i)];
            odd[(size_t)len + i] = in[2 * (p + i) + 1];
        }
        for (size_t i = 0; i < c; i++) out[p + i] = hbf_get_f64(taps, m, odd + i) + even[i];
        memmove(even, even + c, sizeof(double) * (size_t)(m - 1));
        memmove(odd, odd + c, sizeof(double) * (size_t)len);
    }
    memcpy(st, even, sizeof(double) * (size_t)(m - 1));
    memcpy(st + (m - 1), odd, sizeof(double) * (size_t)len);
}

/* src/hbf.rs:207-227 */
static void hbf_int_stage_f64(const double *taps, int m, double *st, const double *in, double *out, size_t n)
{
    int len = 2 * m - 1;
    double xb[2 * IDSP_HBF_MAX_TAPS - 1 + HBF_BLOCK];
    memcpy(xb, st, sizeof(double) * (size_t)len);
    for (size_t p = 0; p < n; p += HBF_BLOCK) {
        size_t c = n - p < HBF_BLOCK ? n - p : HBF_BLOCK;
        memcpy(xb + len, in + p, sizeof(double) * c);
        for (size_t i = 0; i < c; i++) {
            out[2 * (p + i)] = hbf_get_f64(taps, m, xb + i);
            out[2 * (p + i) + 1] = xb[(size_t)m + i];
        }
        memmove(xb, xb + c, sizeof(double) * (size_t)len);
    }
    memcpy(st, xb, sizeof(double) * (size_t)len);
}

int idsp_ref_hbf_dec_f64(const idsp_hbf_cascade_f64 *cfg, void *state, const double *x, double *y,
                         size_t lanes, size_t frames, int layout)
{
    if (!hbf_cfg_ok_f64(cfg) || (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR)) return IDSP_EINVAL;
    if (lanes && (!state || (frames && (!x || !y)))) return IDSP_EINVAL;
    size_t R = (size_t)1 << cfg->stages, W = idsp_ref_hbf_dec_state_words_f64(cfg) / 2;
    uint32_t *st = (uint32_t *)state;
    double *a = (double *)malloc(sizeof(double) * (frames * R + 1));
    double *b = (double *)malloc(sizeof(double) * (frames * R / 2 + 1));
    double *ls = (double *)malloc(sizeof(double) * W);
    if (!a || !b || !ls) { free(a); free(b); free(ls); return IDSP_EINVAL; }
    for (size_t l = 0; l < lanes; l++) {
        for (size_t w = 0; w < W; w++) ls[w] = plane_f64(st, w, lanes, l);
        for (size_t f = 0; f < frames; f++)
            memcpy(a + f * R, x + idx_of(f, l, lanes, frames, layout) * R, sizeof(double) * R);
        size_t n = frames * R, off = 0;
        double *src = a, *dst = b;
        for (int s = 0; s < cfg->stages; s++) {
            n /= 2;
            hbf_dec_stage_f64(cfg->taps[s], cfg->m[s], ls + off, src, dst, n);
            off += (size_t)(3 * cfg->m[s] - 2);
            double *t = src; src = dst; dst = t;
        }
        for (size_t f = 0; f < frames; f++) y[idx_of(f, l, lanes, frames, layout)] = src[f];
        for (size_t w = 0; w < W; w++) plane_f64_put(st, w, lanes, l, ls[w]);
    }
    free(a); free(b); free(ls);
    return IDSP_OK;
}

int idsp_ref_hbf_int_f64(const idsp_hbf_cascade_f64 *cfg, void *state, const double *x, double *y,
                         size_t lanes, size_t frames, int layout)
{
    if (!hbf_cfg_ok_f64(cfg) || (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR)) return IDSP_EINVAL;
    if (lanes && (!state || (frames && (!x || !y)))) return IDSP_EINVAL;
    size_t R = (size_t)1 << cfg->stages, W = idsp_ref_hbf_int_state_words_f64(cfg) / 2;
    uint32_t *st = (uint32_t *)state;
    double *a = (double *)malloc(sizeof(double) * (frames * R + 1));
    double *b = (double *)malloc(sizeof(double) * (frames * R + 1));
    double *ls = (double *)malloc(sizeof(double) * W);
    if (!a || !b || !ls) { free(a); free(b); free(ls); return IDSP_EINVAL; }
    for (size_t l = 0; l < lanes; l++) {
        for (size_t w = 0; w < W; w++) ls[w] = plane_f64(st, w, lanes, l);
        for (size_t f = 0; f < frames; f++) a[f] = x[idx_of(f, l, lanes, frames, layout)];
        size_t n = frames, off = 0;
        double *src = a, *dst = b;
        for (int s = 0; s < cfg->stages; s++) {
            hbf_int_stage_f64(cfg->taps[s], cfg->m[s], ls + off, src, dst, n);
            n *= 2;
            off += (size_t)(2 * cfg->m[s] - 1);
            double *t = src; src = dst; dst = t;
        }
        for (size_t f = 0; f < frames; f++)
            memcpy(y + idx_of(f, l, lanes, frames, layout) * R, src + f * R, sizeof(double) * R);
        for (size_t w = 0; w < W; w++) plane_f64_put(st, w, lanes, l, ls[w]);
    }
    free(a); free(b); free(ls);
    return IDSP_OK;
}

/* src/hbf.rs:70-138 on f64 */
int idsp_ref_fir_sym_f64_process(const idsp_fir_sym_f64 *c, void *state, const double *x, double *y,
                                 size_t lanes, size_t frames, int layout)
{
    size_t len = idsp_ref_fir_sym_state_words_f64(c) / 2;
    if (!len || (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR)) return IDSP_EINVAL;
    if (lanes && (!state || (frames && (!x || !y)))) return IDSP_EINVAL;
    int m = c->m;
    int odd = (c->kind == IDSP_FIR_ODD_SYMMETRIC || c->kind == IDSP_FIR_ODD_ANTISYMMETRIC);
    int sym = (c->kind == IDSP_FIR_ODD_SYMMETRIC || c->kind == IDSP_FIR_EVEN_SYMMETRIC);
    uint32_t *st = (uint32_t *)state;
    for (size_t l = 0; l < lanes; l++) {
        double buf[2 * IDSP_HBF_MAX_TAPS + HBF_BLOCK];
        for (size_t w = 0; w < len; w++) buf[w] = plane_f64(st, w, lanes, l);
        for (size_t p = 0; p < frames; p += HBF_BLOCK) {
            size_t n = frames - p < HBF_BLOCK ? frames - p : HBF_BLOCK;
            for (size_t i = 0; i < n; i++) buf[len + i] = x[idx_of(p + i, l, lanes, frames, layout)];
            for (size_t i = 0; i < n; i++) {
                const double *w = buf + i;
                double acc = -0.0;
                for (int k = 0; k < m; k++) {
                    double nw = w[2 * m - 1 + odd - k], od = w[k];
                    acc = acc + (sym ? nw + od : nw - od) * c->taps[k];
                }
                y[idx_of(p + i, l, lanes, frames, layout)] = (odd && sym) ? acc + w[m] : acc;
            }
            memmove(buf, buf + n, sizeof(double) * len);
        }
        for (size_t w = 0; w < len; w++) plane_f64_put(st, w, lanes, l, buf[w]);
    }
    return IDSP_OK;
}

/* ----------------------------------------------------------------- cossin */

#define COSSIN_DEPTH 7
static uint32_t g_cossin[1 << COSSIN_DEPTH];
static pthread_once_t g_cossin_once = PTHREAD_ONCE_INIT;

/* build.rs:8-41: midpoint samples, AMPLITUDE = u16::MAX,
 * cos = round((cos*2 - 1)*A - 1), sin = round(sin*A), entry = cos + (sin << 16). */
static void cossin_init(void)
{
    const double A = 65535.0;
    for (int i = 0; i < (1 << COSSIN_DEPTH); i++) {
        double th = M_PI / 4. * (((double)i + 0.5) / (double)(1 << COSSIN_DEPTH));
        double s = sin(th), c = cos(th);
        uint32_t ci = (uint32_t)round((c * 2. - 1.) * A - 1.);
        uint32_t si = (uint32_t)round(s * A);
        g_cossin[i] = ci + (si << 16);
    }
}

const uint32_t *idsp_ref_cossin_table(void)
{
    pthread_once(&g_cossin_once, cossin_init);
    return g_cossin;
}

/* src/cossin.rs:14-67. */
void idsp_ref_cossin(int32_t phase_in, int32_t *cos_out, int32_t *sin_out)
{
    const uint32_t *lut = idsp_ref_cossin_table();
    uint32_t octant = (uint32_t)phase_in;
    uint32_t ph = (uint32_t)phase_in;
    if (octant & (1u << 29)) ph = ~ph;
    const int ALIGN_MSB = 32 - 16 - 1;
    ph = (ph << 3) >> (32 - COSSIN_DEPTH - ALIGN_MSB);
    int32_t phase = (int32_t)ph;
    uint32_t lookup = lut[phase >> ALIGN_MSB];
    phase &= (1 << ALIGN_MSB) - 1;
    phase -= 1 << (ALIGN_MSB - 1);
    const int32_t PI4 = (int32_t)(M_PI / 4. * (double)(1 << 16));
    int32_t dphi = (phase * PI4) >> 16;
    int32_t c = (int32_t)(uint16_t)lookup + (1 << 16);
    int32_t s = (int32_t)(lookup >> 16);
    int32_t dcos = (s * dphi) >> COSSIN_DEPTH;
    int32_t dsin = (c * dphi) >> (COSSIN_DEPTH + 1);
    c = (int32_t)((uint32_t)c << (ALIGN_MSB - 1)) - dcos;
    s = (int32_t)((uint32_t)s << ALIGN_MSB) + dsin;
    octant ^= octant >> 1;
    if (octant & (1u << 29)) { int32_t t = c; c = s; s = t; }
    if (octant & (1u << 30)) c = (int32_t)(0u - (uint32_t)c);
    if (octant & (1u << 31)) s = (int32_t)(0u - (uint32_t)s);
    *cos_out = c;
    *sin_out = s;
}

int idsp_ref_cossin_i32(const int32_t *phase, int32_t *out, size_t n)
{
    if (n && (!phase || !out)) return IDSP_EINVAL;
    for (size_t i = 0; i < n; i++) idsp_ref_cossin(phase[i], &out[2 * i], &out[2 * i + 1]);
    return IDSP_OK;
}

/* ----------------------------------------------------------- Normal, Wdf */
/* `Normal<C>` x `DirectForm1<T>` (src/iir/normal.rs:37-58); ba = [b0, b1, b2, p.re, p.im] */
static inline int32_t normal_i32(const int32_t ba[5], int frac, uint32_t *s, int32_t x0)
{
    int32_t x1 = (int32_t)s[0], x2 = (int32_t)s[1], y0o = (int32_t)s[2], y1o = (int32_t)s[3];
    int32_t nim = (int32_t)(0u - (uint32_t)ba[4]);
    int64_t acc = (int64_t)ba[0] * x0;
    acc = wadd64(acc, (int64_t)ba[1] * x1);
    acc = wadd64(acc, (int64_t)ba[2] * x2);
    acc = wadd64(acc, (int64_t)ba[3] * y1o);
    acc = wadd64(acc, (int64_t)nim * y0o);
    int32_t y1 = (int32_t)(acc >> frac);
    int32_t y0 = (int32_t)(wadd64((int64_t)ba[4] * y1o, (int64_t)ba[3] * y0o) >> frac);
    s[1] = (uint32_t)x1; s[0] = (uint32_t)x0; s[2] = (uint32_t)y0; s[3] = (uint32_t)y1;
    return y0;
}
static inline float normal_f32(const float ba[5], uint32_t *s, float x0)
{
    float x1 = f32_from_bits(s[0]), x2 = f32_from_bits(s[1]), y0o = f32_from_bits(s[2]), y1o = f32_from_bits(s[3]);
    float acc = ba[0] * x0;
    acc = acc + ba[1] * x1;
    acc = acc + ba[2] * x2;
    acc = acc + ba[3] * y1o;
    acc = acc + (-ba[4]) * y0o;
    float y0 = ba[4] * y1o + ba[3] * y0o;
    s[1] = s[0]; s[0] = f32_to_bits(x0); s[2] = f32_to_bits(y0); s[3] = f32_to_bits(acc);
    return y0;
}
static inline double normal_f64(const double ba[5], uint32_t *s, double x0)
{
    double x1 = f64_from_words(s, 0), x2 = f64_from_words(s, 1), y0o = f64_from_words(s, 2), y1o = f64_from_words(s, 3);
    double acc = ba[0] * x0;
    acc = acc + ba[1] * x1;
    acc = acc + ba[2] * x2;
    acc = acc + ba[3] * y1o;
    acc = acc + (-ba[4]) * y0o;
    double y0 = ba[4] * y1o + ba[3] * y0o;
    f64_to_words(s, 1, x1); f64_to_words(s, 0, x0); f64_to_words(s, 2, y0); f64_to_words(s, 3, acc);
    return y0;
}
LANE_DRIVER(idsp_ref_normal_i32_df1, idsp_biquad_i32, int32_t, 4, frac_ok_i32(cfg, n), normal_i32(c->ba, c->frac, s, x0))
LANE_DRIVER(idsp_ref_normal_f32_df1, idsp_biquad_f32, float, 4, 1, normal_f32(c->ba, s, x0))
LANE_DRIVER(idsp_ref_normal_f64_df1, idsp_biquad_f64, double, 8, 1, normal_f64(c->ba, s, x0))

/* src/iir/normal.rs:62-76 */
int idsp_ref_normal_from_sos(const double sos[6], double out[5])
{
    if (!sos || !out) return IDSP_EINVAL;
    double a0 = 1.0 / sos[3], p2 = -0.5 * sos[4], pq = sos[3] * sos[5] - p2 * p2;
    if (!(pq >= 0.0)) return IDSP_EINVAL;
    out[0] = sos[0] * a0; out[1] = sos[1] * a0; out[2] = sos[2] * a0;
    out[3] = p2 * a0; out[4] = sqrt(pq) * a0;
    return IDSP_OK;
}

/* `Tpa::adapt` (src/iir/wdf.rs:65-100) */
static inline int32_t wdf_mulq(int32_t c, int32_t a) { return (int32_t)(((int64_t)c * (int64_t)a) >> 32); }
static void tpa_adapt(uint32_t nib, int32_t a, const int32_t x[2], int32_t o[2])
{
    int32_t c, y;
    switch (nib) {
    case 0xA: c = wsub32(x[1], x[0]); y = wadd32(wdf_mulq(c, a), x[1]); o[0] = wadd32(y, c); o[1] = y; break;
    case 0xB: c = wsub32(x[0], x[1]); y = wadd32(wdf_mulq(c, a), x[1]); o[0] = y; o[1] = wadd32(y, c); break;
    case 0xE: c = wsub32(x[0], x[1]); y = wdf_mulq(c, a); o[0] = wadd32(y, x[1]); o[1] = wadd32(y, x[0]); break;
    case 0x1: o[0] = x[1]; o[1] = x[0]; break;
    case 0xC: c = wsub32(x[1], x[0]); y = wsub32(wdf_mulq(c, a), x[1]); o[0] = y; o[1] = wadd32(y, c); break;
    case 0xF: c = wsub32(x[1], x[0]); y = wdf_mulq(c, a); o[0] = wsub32(y, x[1]); o[1] = wsub32(y, x[0]); break;
    case 0xD: c = wsub32(x[0], x[1]); y = wsub32(wdf_mulq(c, a), x[1]); o[0] = wadd32(y, c); o[1] = y; break;
    default: o[0] = x[0]; o[1] = x[1]; break;
    }
}
/* `Wdf::process` (src/iir/wdf.rs:153-169) */
static int32_t wdf_step(const idsp_wdf *c, int32_t *z, int32_t x)
{
    int32_t y = 0;
    int32_t *dst = &y;
    uint32_t m = c->m;
    for (int i = 0; i < c->n; i++, m >>= 4) {
        int32_t in[2] = {x, z[i]}, o[2];
        tpa_adapt(m & 0xf, c->a[i], in, o);
        *dst = o[0];
        x = o[1];
        dst = &z[i];
    }
    *dst = x;
    return y;
}
static int wdf_ok(const idsp_wdf *c, size_t n)
{
    for (size_t k = 0; k < n; k++) if (c[k].n < 1 || c[k].n > IDSP_WDF_MAX_ORDER) return 0;
    return 1;
}
size_t idsp_ref_wdf_state_words(const idsp_wdf *cfg, size_t n)
{
    if (!cfg || !wdf_ok(cfg, n)) return 0;
    size_t w = 0;
    for (size_t k = 0; k < n; k++) w += (size_t)cfg[k].n;
    return w;
}
/* `Wdf::quantize` (src/iir/wdf.rs:126-137) with `Tpa::quantize` (:50-62) */
int idsp_ref_wdf_quantize(int n, uint32_t m, const double *g, idsp_wdf *out)
{
    if (!g || !out || n < 1 || n > IDSP_WDF_MAX_ORDER) return IDSP_EINVAL;
    memset(out, 0, sizeof(*out));
    out->n = n; out->m = m;
    for (int i = 0; i < n; i++, m >>= 4) {
        double a;
        switch (m & 0xf) {
        case 0xA: a = g[i] - 1.0; break;
        case 0xB: case 0xE: a = -g[i]; break;
        case 0xC: case 0xF: a = g[i]; break;
        case 0xD: a = -1.0 - g[i]; break;
        default: a = 0.0; break;
        }
        if (!(a >= -0.5 && a <= 0.0)) return IDSP_EOUTOFRANGE;
        out->a[i] = idsp_ref_quantize_f64(a, 32);
    }
    return IDSP_OK;
}
int idsp_ref_wdf_i32(const idsp_wdf *cfg, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes,
                     size_t frames, int layout)
{
    int rc = check_common(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if (!wdf_ok(cfg, n)) return IDSP_EINVAL;
    uint32_t *st = (uint32_t *)state;
    for (size_t l = 0; l < lanes; l++) {
        size_t w = 0;
        if (n == 0)
            for (size_t f = 0; f < frames; f++) { size_t i = idx_of(f, l, lanes, frames, layout); y[i] = x[i]; }
        for (size_t k = 0; k < n; k++) { /* stage-major: section k over the whole lane, like compose.rs:43-77 */
            int32_t z[IDSP_WDF_MAX_ORDER];
            for (int i = 0; i < cfg[k].n; i++) z[i] = (int32_t)st[(w + i) * lanes + l];
            const int32_t *src = k == 0 ? x : y;
            for (size_t f = 0; f < frames; f++) {
                size_t i = idx_of(f, l, lanes, frames, layout);
                y[i] = wdf_step(&cfg[k], z, src[i]);
            }
            for (int i = 0; i < cfg[k].n; i++) st[(w + i) * lanes + l] = (uint32_t)z[i];
            w += (size_t)cfg[k].n;
        }
    }
    return IDSP_OK;
}

/* -------------------------------------------------------------------- Cic */
/* `Cic<T, N, M>` (src/cic.rs); per-lane state values: zoh, combs[N][M], integrators[N]. */
static int cic_ok(const idsp_cic *c)
{
    return c && c->order >= 1 && c->order <= IDSP_CIC_MAX_ORDER && c->comb_delay >= 1 && c->comb_delay <= IDSP_CIC_MAX_DELAY;
}
int64_t idsp_ref_cic_gain(const idsp_cic *c)
{
    if (!cic_ok(c)) return 0;
    uint64_t b = (uint64_t)c->comb_delay * ((uint64_t)c->rate + 1), g = 1; /* cic.rs:103-105 */
    for (int i = 0; i < c->order; i++) g *= b;
    return (int64_t)g;
}
int idsp_ref_cic_gain_log2(const idsp_cic *c)
{
    if (!cic_ok(c)) return IDSP_EINVAL;
    uint32_t v = (uint32_t)c->comb_delay * c->rate + (uint32_t)(c->comb_delay - 1); /* cic.rs:111-113 */
    return (v ? 32 - __builtin_clz(v) : 0) * c->order;
}
size_t idsp_ref_cic_response_length(const idsp_cic *c) { return cic_ok(c) ? (size_t)c->rate * (size_t)c->order : 0; }
size_t idsp_ref_cic_state_words(const idsp_cic *c, int bits)
{
    if (!cic_ok(c) || (bits != 32 && bits != 64)) return 0;
    return (size_t)(1 + c->order * c->comb_delay + c->order) * (size_t)(bits / 32);
}

#define CIC_IMPL(SUF, T, UT, VW)                                                                        \
    static T cic_ld_##SUF(const uint32_t *st, size_t lanes, size_t l, int v)                             \
    {                                                                                                   \
        UT u = 0;                                                                                       \
        for (int w = 0; w < VW; w++) u |= (UT)st[(size_t)(v * VW + w) * lanes + l] << (32 * w);         \
        return (T)u;                                                                                    \
    }                                                                                                   \
    static void cic_st_##SUF(uint32_t *st, size_t lanes, size_t l, int v, T x)                           \
    {                                                                                                   \
        for (int w = 0; w < VW; w++) st[(size_t)(v * VW + w) * lanes + l] = (uint32_t)((UT)x >> (32 * w)); \
    }                                                                                                   \
    /* cic.rs:166-171 / 197-203 */                                                                      \
    static T cic_combs_##SUF(T *comb, int n_, int m, T x)                                                \
    {                                                                                                   \
        for (int n = 0; n < n_; n++) {                                                                  \
            T *c = comb + n * m;                                                                        \
            T y = (T)((UT)x - (UT)c[0]);                                                                \
            memmove(c, c + 1, (size_t)(m - 1) * sizeof(T));                                             \
            c[m - 1] = x;                                                                               \
            x = y;                                                                                      \
        }                                                                                               \
        return x;                                                                                       \
    }                                                                                                   \
    static T cic_integ_##SUF(T *integ, int n_, T x)                                                      \
    {                                                                                                   \
        for (int n = 0; n < n_; n++) { integ[n] = (T)((UT)integ[n] + (UT)x); x = integ[n]; }            \
        return x;                                                                                       \
    }                                                                                                   \
    static int cic_run_##SUF(const idsp_cic *cfg, void *state, const T *x, T *y, size_t lanes,           \
                             size_t frames, int layout, int dec)                                        \
    {                                                                                                   \
        if (!cic_ok(cfg)) return IDSP_EINVAL;                                                           \
        int rc = check_common(cfg, 1, state, x, y, lanes, frames, layout);                              \
        if (rc) return rc;                                                                              \
        uint32_t *st = (uint32_t *)state;                                                               \
        const int N = cfg->order, M = cfg->comb_delay;                                                  \
        const size_t R = (size_t)cfg->rate + 1;                                                         \
        for (size_t l = 0; l < lanes; l++) {                                                            \
            T zoh = cic_ld_##SUF(st, lanes, l, 0), comb[IDSP_CIC_MAX_ORDER * IDSP_CIC_MAX_DELAY],       \
              integ[IDSP_CIC_MAX_ORDER];                                                                \
            for (int i = 0; i < N * M; i++) comb[i] = cic_ld_##SUF(st, lanes, l, 1 + i);                \
            for (int i = 0; i < N; i++) integ[i] = cic_ld_##SUF(st, lanes, l, 1 + N * M + i);           \
            uint32_t index = 0;                                                                         \
            for (size_t f = 0; f < frames; f++) {                                                       \
                size_t lo = idx_of(f, l, lanes, frames, layout);                                        \
                if (dec) { /* Process<T, Option<T>>, cic.rs:186-207 */                                  \
                    int ticks = 0;                                                                      \
                    for (size_t r = 0; r < R; r++) {                                                    \
                        T v = cic_integ_##SUF(integ, N, x[lo * R + r]);                                 \
                        if (index > 0) { index--; continue; }                                           \
                        index = cfg->rate;                                                              \
                        zoh = cic_combs_##SUF(comb, N, M, v);                                           \
                        y[lo] = zoh;                                                                    \
                        ticks++;                                                                        \
                    }                                                                                   \
                    if (ticks != 1) return IDSP_EINVAL; /* Decimator: exactly one tick per chunk */     \
                } else { /* Process<Option<T>, T>, cic.rs:160-182 */                                    \
                    for (size_t r = 0; r < R; r++) {                                                    \
                        if (r == 0) { index = cfg->rate; zoh = cic_combs_##SUF(comb, N, M, x[lo]); }    \
                        else index--;                                                                   \
                        y[lo * R + r] = cic_integ_##SUF(integ, N, zoh);                                 \
                    }                                                                                   \
                }                                                                                       \
            }                                                                                           \
            cic_st_##SUF(st, lanes, l, 0, zoh);                                                         \
            for (int i = 0; i < N * M; i++) cic_st_##SUF(st, lanes, l, 1 + i, comb[i]);                 \
            for (int i = 0; i < N; i++) cic_st_##SUF(st, lanes, l, 1 + N * M + i, integ[i]);            \
        }                                                                                               \
        return IDSP_OK;                                                                                 \
    }                                                                                                   \
    int idsp_ref_cic_dec_##SUF(const idsp_cic *cfg, void *state, const T *x, T *y, size_t lanes,         \
                               size_t frames, int layout)                                               \
    { return cic_run_##SUF(cfg, state, x, y, lanes, frames, layout, 1); }                               \
    int idsp_ref_cic_int_##SUF(const idsp_cic *cfg, void *state, const T *x, T *y, size_t lanes,         \
                               size_t frames, int layout)                                               \
    { return cic_run_##SUF(cfg, state, x, y, lanes, frames, layout, 0); }

CIC_IMPL(i32, int32_t, uint32_t, 1)
CIC_IMPL(i64, int64_t, uint64_t, 2)

/* ------------------------------------------------------------------ atan2 */
static uint32_t g_atan2_base[16];
static int32_t g_atan2_slope[16];
static pthread_once_t g_atan2_once = PTHREAD_ONCE_INIT;

/* build.rs:43-66 */
static void atan2_init(void)
{
    const double Q31 = (double)((int64_t)1 << 31);
    for (int i = 0; i < 16; i++) {
        double x0 = 1.0 + (double)i / 16.0, x1 = 1.0 + (double)(i + 1) / 16.0;
        g_atan2_base[i] = (uint32_t)round(Q31 / x0);
        g_atan2_slope[i] = (int32_t)round((1.0 / x1 - 1.0 / x0) * Q31);
    }
}

/* src/atan2.rs:6-9 */
static inline uint32_t mul_q31(uint32_t x, uint32_t y) { return (uint32_t)(((uint64_t)x * (uint64_t)y) >> 31); }

/* src/atan2.rs:12-29 */
static uint32_t atan2_divi(uint32_t y, uint32_t x)
{
    if (x == 0) return 0;
    int shift = __builtin_clz(x);
    y <<= shift;
    x <<= shift;
    const int FRAC_BITS = 31 - 4;
    uint32_t rem = x & ((1u << FRAC_BITS) - 1u);
    uint32_t idx = (x << 1) >> (1 + FRAC_BITS);
    uint32_t step = (uint32_t)(((int64_t)g_atan2_slope[idx] * (int64_t)rem) >> FRAC_BITS);
    uint32_t r0 = g_atan2_base[idx] + step;
    return mul_q31(y, mul_q31(r0, 0u - mul_q31(x, r0)));
}

/* src/atan2.rs:32-49: `(r * x2) + a` on Q32<32>: ((r*x2) >> 32) as i32, wrapping add */
static uint32_t atan2_atani(uint32_t x)
{
    static const int32_t ATANI[6] = {0x0517c2cd, -0x06c6496b, 0x0fbdb021, -0x25b32e0a, 0x43b34c81, -0x3bc823dd};
    int32_t x2 = (int32_t)(((int64_t)x * (int64_t)x) >> 32);
    int32_t r = 0;
    for (int i = 5; i >= 0; i--) r = wadd32((int32_t)(((int64_t)r * (int64_t)x2) >> 32), ATANI[i]);
    return (uint32_t)(((int64_t)r * (int64_t)x) >> 28);
}

/* src/atan2.rs:66-82 */
int32_t idsp_ref_atan2(int32_t y, int32_t x)
{
    pthread_once(&g_atan2_once, atan2_init);
    uint32_t k = 0;
    if (y < 0) { y = y == INT32_MIN ? INT32_MAX : -y; k ^= UINT32_MAX; }
    if (x < 0) { x = x == INT32_MIN ? INT32_MAX : -x; k ^= UINT32_MAX >> 1; }
    if (y > x) { int32_t t = y; y = x; x = t; k ^= UINT32_MAX >> 2; }
    return (int32_t)(atan2_atani(atan2_divi((uint32_t)y, (uint32_t)x)) ^ k);
}

int idsp_ref_atan2_i32(const int32_t *xy, int32_t *out, size_t n)
{
    if (n && (!xy || !out)) return IDSP_EINVAL;
    for (size_t i = 0; i < n; i++) out[i] = idsp_ref_atan2(xy[2 * i + 1], xy[2 * i]);
    return IDSP_OK;
}

/* FM discriminator + deemphasis: examples/fm_disc.rs:25-50.  State words {has_prev, prev.re, prev.im, x0, x1, y0, y1}. */
int idsp_ref_fm_disc_i32(const idsp_fm_disc *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames,
                         int layout)
{
    int rc = check_common(cfg, 1, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if (cfg->deemph.frac < 0 || cfg->deemph.frac > 31) return IDSP_EINVAL;
    uint32_t *st = (uint32_t *)state;
    for (size_t l = 0; l < lanes; l++) {
        uint32_t has_prev = st[l], s[4];
        int32_t pre = (int32_t)st[lanes + l], pim = (int32_t)st[2 * lanes + l];
        for (int w = 0; w < 4; w++) s[w] = st[(size_t)(3 + w) * lanes + l];
        for (size_t f = 0; f < frames; f++) {
            size_t i = idx_of(f, l, lanes, frames, layout);
            int32_t xr = x[2 * i], xi = x[2 * i + 1], d = 0;
            if (has_prev) { /* `prev.replace(x)` returned Some(p): fm_disc.rs:33-37 */
                int32_t cim = (int32_t)(0u - (uint32_t)pim);                       /* conj, complex.rs:55-57 */
                int64_t re = (int64_t)((uint64_t)((int64_t)xr * pre) - (uint64_t)((int64_t)xi * cim)); /* complex.rs:128-133 */
                int64_t im = wadd64((int64_t)xr * cim, (int64_t)xi * pre);
                d = (int32_t)((uint32_t)idsp_ref_atan2((int32_t)(im >> 32), (int32_t)(re >> 32)) - (uint32_t)cfg->carrier);
            }
            has_prev = 1; pre = xr; pim = xi;
            y[i] = df1_i32(cfg->deemph.ba, cfg->deemph.frac, s, d);
        }
        st[l] = has_prev; st[lanes + l] = (uint32_t)pre; st[2 * lanes + l] = (uint32_t)pim;
        for (int w = 0; w < 4; w++) st[(size_t)(3 + w) * lanes + l] = s[w];
    }
    return IDSP_OK;
}

/* src/accu.rs:34-41 (state += step; yield state) -> src/complex.rs:237-240. */
int idsp_ref_dds_i32(void *state, int32_t *out, size_t lanes, size_t frames, int layout)
{
    if (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR) return IDSP_EINVAL;
    if (lanes && (!state || (frames && !out))) return IDSP_EINVAL;
    uint32_t *st = (uint32_t *)state;
    for (size_t l = 0; l < lanes; l++) {
        uint32_t acc = st[l], step = st[lanes + l];
        for (size_t f = 0; f < frames; f++) {
            acc += step;
            size_t i = idx_of(f, l, lanes, frames, layout);
            idsp_ref_cossin((int32_t)acc, &out[2 * i], &out[2 * i + 1]);
        }
        st[l] = acc;
    }
    return IDSP_OK;
}

/* ---------------------------------------------------- lowpass and lock-in */

/* i32::saturating_sub */
static inline int32_t sat_sub_i32(int32_t a, int32_t b)
{
    int64_t d = (int64_t)a - (int64_t)b;
    return d > INT32_MAX ? INT32_MAX : (d < INT32_MIN ? INT32_MIN : (int32_t)d);
}

static inline int64_t wmul64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }

/* src/lowpass.rs:47-78; s = [i64; N]. */
static inline int32_t lowpass_step(int order, const int32_t *k, int64_t *s, int32_t x)
{
    int64_t d = (int64_t)sat_sub_i32(x, (int32_t)(s[0] >> 32)) * (int64_t)k[0];
    int32_t y;
    if (order == 1) {
        s[0] = wadd64(s[0], d);
        y = (int32_t)(s[0] >> 32);
        s[0] = wadd64(s[0], d);
    } else {
        d = wadd64(d, wmul64(s[1] >> 32, (int64_t)k[1]));
        s[1] = wadd64(s[1], d);
        s[0] = wadd64(s[0], s[1]);
        y = (int32_t)(s[0] >> 32);
        s[0] = wadd64(s[0], s[1]);
        s[1] = wadd64(s[1], d);
    }
    return y;
}

static int lockin_cfg_ok(const idsp_lockin_i32 *c)
{
    return c && (c->order == 1 || c->order == 2) && c->cascade >= 1 && c->cascade <= IDSP_LOCKIN_MAX_CASCADE;
}

size_t idsp_ref_lockin_state_words(const idsp_lockin_i32 *c)
{
    return lockin_cfg_ok(c) ? (size_t)(2 + 2 * c->cascade * c->order * 2) : 0;
}

/* `[Lowpass<N>; K]` array composition, sample-major `process`
 * (dsp-process/src/compose.rs:84-93). */
static inline int32_t lowpass_cascade(const idsp_lockin_i32 *c, int64_t *s, int32_t x)
{
    for (int k = 0; k < c->cascade; k++) x = lowpass_step(c->order, c->k[k], s + k * c->order, x);
    return x;
}

int idsp_ref_lowpass_i32(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y,
                         size_t lanes, size_t frames, int layout)
{
    if (!lockin_cfg_ok(cfg) || (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR)) return IDSP_EINVAL;
    if (lanes && (!state || (frames && (!x || !y)))) return IDSP_EINVAL;
    uint32_t *st = (uint32_t *)state;
    int ns = cfg->cascade * cfg->order;
    for (size_t l = 0; l < lanes; l++) {
        int64_t s[IDSP_LOCKIN_MAX_CASCADE * 2];
        for (int j = 0; j < ns; j++)
            s[j] = (int64_t)(((uint64_t)st[(size_t)(2 * j + 1) * lanes + l] << 32) | st[(size_t)(2 * j) * lanes + l]);
        for (size_t f = 0; f < frames; f++) {
            size_t i = idx_of(f, l, lanes, frames, layout);
            y[i] = lowpass_cascade(cfg, s, x[i]);
        }
        for (int j = 0; j < ns; j++) {
            st[(size_t)(2 * j) * lanes + l] = (uint32_t)(uint64_t)s[j];
            st[(size_t)(2 * j + 1) * lanes + l] = (uint32_t)((uint64_t)s[j] >> 32);
        }
    }
    return IDSP_OK;
}

/* src/lockin.rs:30-39 then :17-27; `x * lo` = `i32 * Q32<32>` =
 * ((q as i64 * x as i64) >> 32) as i32 (dsp-fixedpoint/src/lib.rs:449-456,324-326). */
enum { LOCKIN_IQ = 0, LOCKIN_ARG = 1, LOCKIN_NORM_SQR = 2 };

static int lockin_run(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, void *yv, size_t lanes, size_t frames,
                      int layout, int mode)
{
    int32_t *y = (int32_t *)yv;
    if (!lockin_cfg_ok(cfg) || (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR)) return IDSP_EINVAL;
    if (lanes && (!state || (frames && (!x || !y)))) return IDSP_EINVAL;
    uint32_t *st = (uint32_t *)state;
    int ns = cfg->cascade * cfg->order;
    for (size_t l = 0; l < lanes; l++) {
        uint32_t acc = st[l], step = st[lanes + l];
        int64_t s[2][IDSP_LOCKIN_MAX_CASCADE * 2];
        for (int q = 0; q < 2; q++)
            for (int j = 0; j < ns; j++) {
                size_t w = (size_t)(2 + (q * ns + j) * 2);
                s[q][j] = (int64_t)(((uint64_t)st[(w + 1) * lanes + l] << 32) | st[w * lanes + l]);
            }
        for (size_t f = 0; f < frames; f++) {
            size_t i = idx_of(f, l, lanes, frames, layout);
            acc += step;
            int32_t c, sn;
            idsp_ref_cossin((int32_t)acc, &c, &sn);
            int32_t xi = trunc32(mul_wide(c, x[i]) >> 32);
            int32_t xq = trunc32(mul_wide(sn, x[i]) >> 32);
            int32_t re = lowpass_cascade(cfg, s[0], xi);
            int32_t im = lowpass_cascade(cfg, s[1], xq);
            if (mode == LOCKIN_IQ) {
                y[2 * i] = re;
                y[2 * i + 1] = im;
            } else if (mode == LOCKIN_ARG) {
                y[i] = idsp_ref_atan2(im, re); /* Complex::arg, src/complex.rs:254-256 */
            } else {
                /* Complex::<i32>::norm_sqr, src/complex.rs:214-217; the sum wraps for (MIN, MIN) as in release */
                ((int64_t *)yv)[i] = (int64_t)((uint64_t)mul_wide(re, re) + (uint64_t)mul_wide(im, im));
            }
        }
        st[l] = acc;
        for (int q = 0; q < 2; q++)
            for (int j = 0; j < ns; j++) {
                size_t w = (size_t)(2 + (q * ns + j) * 2);
                st[w * lanes + l] = (uint32_t)(uint64_t)s[q][j];
                st[(w + 1) * lanes + l] = (uint32_t)((uint64_t)s[q][j] >> 32);
            }
    }
    return IDSP_OK;
}

int idsp_ref_lockin_i32_process(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y,
                                size_t lanes, size_t frames, int layout)
{
    return lockin_run(cfg, state, x, y, lanes, frames, layout, LOCKIN_IQ);
}

/* ---- `Lockin<C>` as the reference defines it (src/lockin.rs:16-39): any arm filter C, the LO from a phase or given ----
 * Arms below: C = `[Biquad<Q32<F>>; n]` x `[DirectForm1<i32>; n]` (src/iir/biquad.rs:366-383, array composition
 * dsp-process/src/compose.rs:80-113, sample-major), n = 1 being a plain `Biquad<Q32<F>>`; and `[Biquad<f32>; n]` for the
 * f32 graph of examples/ddc_lockin.rs:35-42.  The same n sections run on I (state[0]) and on Q (state[1]).
 * External LO (src/lockin.rs:17-27): `Complex::new(C(state[0], x * lo.re), C(state[1], x * lo.im))`; `i32 * Q32<32>` =
 * ((q as i64 * x as i64) >> 32) as i32 (dsp-fixedpoint/src/lib.rs:449-456), `f32 * f32` one rounded multiply. */
static int biquad_arms_ok(const void *sections, size_t n, const void *state, const void *x, const void *y, size_t lanes,
                          size_t frames, int layout)
{
    if (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR) return 0;
    if (!sections || n < 1 || n > IDSP_LOCKIN_MAX_SECTIONS) return 0;
    if (lanes && (!state || (frames && (!x || !y)))) return 0;
    return 1;
}

size_t idsp_ref_lockin_biquad_state_words(size_t n, int with_accu)
{
    return n >= 1 && n <= IDSP_LOCKIN_MAX_SECTIONS ? (size_t)(with_accu ? 2 : 0) + 8 * n : 0;
}

/* phase form (src/lockin.rs:30-39) with biquad arms; state words { accu.state, accu.step, I: n x {x0,x1,y0,y1}, Q: n x {..} } */
int idsp_ref_lockin_i32_biquad_process(const idsp_biquad_i32 *sec, size_t n, void *state, const int32_t *x, int32_t *y,
                                       size_t lanes, size_t frames, int layout)
{
    if (!biquad_arms_ok(sec, n, state, x, y, lanes, frames, layout)) return IDSP_EINVAL;
    for (size_t k = 0; k < n; k++)
        if (sec[k].frac < 0 || sec[k].frac > 31) return IDSP_EINVAL;
    uint32_t *st = (uint32_t *)state;
    for (size_t l = 0; l < lanes; l++) {
        uint32_t acc = st[l], step = st[lanes + l];
        uint32_t s[2][IDSP_LOCKIN_MAX_SECTIONS][4];
        for (int q = 0; q < 2; q++)
            for (size_t k = 0; k < n; k++)
                for (int w = 0; w < 4; w++) s[q][k][w] = st[(2 + (q * n + k) * 4 + w) * lanes + l];
        for (size_t f = 0; f < frames; f++) {
            size_t i = idx_of(f, l, lanes, frames, layout);
            acc += step;
            int32_t c, sn;
            idsp_ref_cossin((int32_t)acc, &c, &sn);
            int32_t v[2] = {trunc32(mul_wide(c, x[i]) >> 32), trunc32(mul_wide(sn, x[i]) >> 32)};
            for (int q = 0; q < 2; q++) {
                for (size_t k = 0; k < n; k++) v[q] = df1_i32(sec[k].ba, sec[k].frac, s[q][k], v[q]);
                y[2 * i + q] = v[q];
            }
        }
        st[l] = acc;
        for (int q = 0; q < 2; q++)
            for (size_t k = 0; k < n; k++)
                for (int w = 0; w < 4; w++) st[(2 + (q * n + k) * 4 + w) * lanes + l] = s[q][k][w];
    }
    return IDSP_OK;
}

/* external LO, `[Lowpass<N>; K]` arms; state words = those of idsp_lockin_state_words WITHOUT the two accumulator words */
int idsp_ref_lockin_i32_lo_process(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, const int32_t *lo, int32_t *y,
                                   size_t lanes, size_t frames, int layout)
{
    if (!lockin_cfg_ok(cfg) || (layout != IDSP_FRAME_MAJOR && layout != IDSP_LANE_MAJOR)) return IDSP_EINVAL;
    if (lanes && (!state || (frames && (!x || !lo || !y)))) return IDSP_EINVAL;
    uint32_t *st = (uint32_t *)state;
    int ns = cfg->cascade * cfg->order;
    for (size_t l = 0; l < lanes; l++) {
        int64_t s[2][IDSP_LOCKIN_MAX_CASCADE * 2];
        for (int q = 0; q < 2; q++)
            for (int j = 0; j < ns; j++) {
                size_t w = (size_t)((q * ns + j) * 2);
                s[q][j] = (int64_t)(((uint64_t)st[(w + 1) * lanes + l] << 32) | st[w * lanes + l]);
            }
        for (size_t f = 0; f < frames; f++) {
            size_t i = idx_of(f, l, lanes, frames, layout);
            for (int q = 0; q < 2; q++) y[2 * i + q] = lowpass_cascade(cfg, s[q], trunc32(mul_wide(lo[2 * i + q], x[i]) >> 32));
        }
        for (int q = 0; q < 2; q++)
            for (int j = 0; j < ns; j++) {
                size_t w = (size_t)((q * ns + j) * 2);
                st[w * lanes + l] = (uint32_t)(uint64_t)s[q][j];
                st[(w + 1) * lanes + l] = (uint32_t)((uint64_t)s[q][j] >> 32);
            }
    }
    return IDSP_OK;
}

/* external LO, biquad arms (i32); state words { I: n x {x0,x1,y0,y1}, Q: n x {..} } */
int idsp_ref_lockin_i32_biquad_lo_process(const idsp_biquad_i32 *sec, size_t n, void *state, const int32_t *x, const int32_t *lo,
                                          int32_t *y, size_t lanes, size_t frames, int layout)
{
    if (!biquad_arms_ok(sec, n, state, x, y, lanes, frames, layout) || (lanes && frames && !lo)) return IDSP_EINVAL;
    for (size_t k = 0; k < n; k++)
        if (sec[k].frac < 0 || sec[k].frac > 31) return IDSP_EINVAL;
    uint32_t *st = (uint32_t *)state;
    for (size_t l = 0; l < lanes; l++) {
        uint32_t s[2][IDSP_LOCKIN_MAX_SECTIONS][4];
        for (int q = 0; q < 2; q++)
            for (size_t k = 0; k < n; k++)
                for (int w = 0; w < 4; w++) s[q][k][w] = st[((q * n + k) * 4 + w) * lanes + l];
        for (size_t f = 0; f < frames; f++) {
            size_t i = idx_of(f, l, lanes, frames, layout);
            for (int q = 0; q < 2; q++) {
                int32_t v = trunc32(mul_wide(lo[2 * i + q], x[i]) >> 32);
                for (size_t k = 0; k < n; k++) v = df1_i32(sec[k].ba, sec[k].frac, s[q][k], v);
                y[2 * i + q] = v;
            }
        }
        for (int q = 0; q < 2; q++)
            for (size_t k = 0; k < n; k++)
                for (int w = 0; w < 4; w++) st[((q * n + k) * 4 + w) * lanes + l] = s[q][k][w];
    }
    return IDSP_OK;
}

/* external LO, `Biquad<f32>` arms: the `mix * lowpass.lanes()` graph of examples/ddc_lockin.rs:35-42 with lo = (cos, -sin) */
int idsp_ref_lockin_f32_biquad_lo_process(const idsp_biquad_f32 *sec, size_t n, void *state, const float *x, const float *lo,
                                          float *y, size_t lanes, size_t frames, int layout)
{
    if (!biquad_arms_ok(sec, n, state, x, y, lanes, frames, layout) || (lanes && frames && !lo)) return IDSP_EINVAL;
    uint32_t *st = (uint32_t *)state;
    for (size_t l = 0; l < lanes; l++) {
        uint32_t s[2][IDSP_LOCKIN_MAX_SECTIONS][4];
        for (int q = 0; q < 2; q++)
            for (size_t k = 0; k < n; k++)
                for (int w = 0; w < 4; w++) s[q][k][w] = st[((q * n + k) * 4 + w) * lanes + l];
        for (size_t f = 0; f < frames; f++) {
            size_t i = idx_of(f, l, lanes, frames, layout);
            for (int q = 0; q < 2; q++) {
                float v = x[i] * lo[2 * i + q];
                for (size_t k = 0; k < n; k++) v = df1_f32(sec[k].ba, s[q][k], v);
                y[2 * i + q] = v;
            }
        }
        for (int q = 0; q < 2; q++)
            for (size_t k = 0; k < n; k++)
                for (int w = 0; w < 4; w++) st[((q * n + k) * 4 + w) * lanes + l] = s[q][k][w];
    }
    return IDSP_OK;
}

/* `Lockin::process(..).arg()` (src/lockin.rs:30-39, src/complex.rs:254-256) */
int idsp_ref_lockin_i32_arg(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y,
                            size_t lanes, size_t frames, int layout)
{
    return lockin_run(cfg, state, x, y, lanes, frames, layout, LOCKIN_ARG);
}

/* `Lockin::process(..).norm_sqr()` (src/lockin.rs:30-39, src/complex.rs:214-217) */
int idsp_ref_lockin_i32_norm_sqr(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int64_t *y,
                                 size_t lanes, size_t frames, int layout)
{
    return lockin_run(cfg, state, x, y, lanes, frames, layout, LOCKIN_NORM_SQR);
}
