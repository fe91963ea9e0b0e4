// cascade.hip — C-ABI entry points (include/idsp_hip.h) of the i32 cascade with shared delay lines; device code in biquad_sections.h.
// (f32: cascade_f32.hip, f64: biquad_f64.hip — translation units sized for the parallel build.)
#include "biquad_sections.h"

using namespace idsp;
using namespace idsp::bq;

extern "C" {

int idsp_cascade_i32_df1(const idsp_biquad_i32 *cfg, size_t n, void *state, const int32_t *x, int32_t *y,
                         size_t lanes, size_t frames, int layout, void *stream)
{
    int rc = check_stream_args(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if (n > size_t(kMaxCascade)) return fail(IDSP_EINVAL, "cascade sections n = %zu > %d", n, kMaxCascade);
    for (size_t k = 0; k < n; k++)
        if ((rc = check_frac(cfg[k].frac, k))) return rc;
    return run_cascade<int32_t>(FillI32{cfg}, n, state, x, y, lanes, frames, layout, as_stream(stream));
}

// explicit row pitches (include/idsp_hip.h, "_pitch" entries)
int idsp_cascade_i32_df1_pitch(const idsp_biquad_i32 *cfg, size_t n, void *state, const int32_t *x, size_t x_pitch, int32_t *y,
                               size_t y_pitch, size_t lanes, size_t frames, int layout, void *stream)
{
    int rc = check_stream_args(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    if (n > size_t(kMaxCascade)) return fail(IDSP_EINVAL, "cascade sections n = %zu > %d", n, kMaxCascade);
    for (size_t k = 0; k < n; k++)
        if ((rc = check_frac(cfg[k].frac, k))) return rc;
    return run_cascade<int32_t>(FillI32{cfg}, n, state, x, y, lanes, frames, layout, as_stream(stream), Pitch{x_pitch, y_pitch});
}

}  // extern "C"
