// cascade_f32.hip — C-ABI entry points (include/idsp_hip.h) of the f32 cascade with shared delay lines; device code in biquad_sections.h.
#include "biquad_sections.h"

using namespace idsp;
using namespace idsp::bq;

extern "C" {

int idsp_cascade_f32_df1(const idsp_biquad_f32 *cfg, size_t n, void *state, const float *x, float *y,
                         size_t lanes, size_t frames, int layout, void *stream)
{
    int rc = check_stream_args(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    return run_cascade<float>(FillF32{cfg}, n, state, x, y, lanes, frames, layout, as_stream(stream));
}

// explicit row pitches (include/idsp_hip.h, "_pitch" entries)
int idsp_cascade_f32_df1_pitch(const idsp_biquad_f32 *cfg, size_t n, void *state, const float *x, size_t x_pitch, float *y,
                               size_t y_pitch, size_t lanes, size_t frames, int layout, void *stream)
{
    int rc = check_stream_args(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    return run_cascade<float>(FillF32{cfg}, n, state, x, y, lanes, frames, layout, as_stream(stream), Pitch{x_pitch, y_pitch});
}

}  // extern "C"
