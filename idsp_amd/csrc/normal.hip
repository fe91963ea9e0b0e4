// normal.hip — C-ABI entry points (include/idsp_hip.h) of `iir::normal::Normal` (src/iir/normal.rs:37-58); device code in
// biquad_sections.h.
#include <cmath>

#include "biquad_sections.h"

using namespace idsp;
using namespace idsp::bq;

extern "C" {

int idsp_normal_i32_df1(const idsp_biquad_i32 *cfg, size_t n, void *state, const int32_t *x, int32_t *y, size_t lanes,
                        size_t frames, int layout, void *stream)
{
    return entry_i32<NormalI32, idsp_biquad_i32, FillI32>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

int idsp_normal_f32_df1(const idsp_biquad_f32 *cfg, size_t n, void *state, const float *x, float *y, size_t lanes,
                        size_t frames, int layout, void *stream)
{
    return entry_f32<NormalF32, idsp_biquad_f32, FillF32>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

int idsp_normal_f64_df1(const idsp_biquad_f64 *cfg, size_t n, void *state, const double *x, double *y, size_t lanes,
                        size_t frames, int layout, void *stream)
{
    return entry_f64<NormalF64, idsp_biquad_f64, FillF64>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

int idsp_normal_from_sos(const double sos[6], double out[5])
{
    if (!sos || !out) return fail(IDSP_EINVAL, "sos or out is NULL");
    // src/iir/normal.rs:62-76
    const double a0 = 1.0 / sos[3];
    const double p2 = -0.5 * sos[4];
    const double pq = sos[3] * sos[5] - p2 * p2;
    if (!(pq >= 0.0)) return fail(IDSP_EINVAL, "Normal::from: real poles (assert!(pq >= 0.0), src/iir/normal.rs:69)");
    out[0] = sos[0] * a0, out[1] = sos[1] * a0, out[2] = sos[2] * a0;
    out[3] = p2 * a0, out[4] = std::sqrt(pq) * a0;
    return IDSP_OK;
}

}  // extern "C"
