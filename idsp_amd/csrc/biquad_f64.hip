// biquad_f64.hip — C-ABI entry points (include/idsp_hip.h) of the f64 DF1 biquads and the f64 cascade; device code in
// biquad_sections.h.  (DF2T: biquad_f64_df2t.hip — translation units sized for the parallel build.)
#include "biquad_sections.h"

using namespace idsp;
using namespace idsp::bq;

extern "C" {

int idsp_biquad_f64_df1(const idsp_biquad_f64 *cfg, size_t n, void *state, const double *x, double *y,
                        size_t lanes, size_t frames, int layout, void *stream)
{
    return entry_f64<Df1F64<false>, idsp_biquad_f64, FillF64>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

int idsp_biquad_f64_df1_clamp(const idsp_biquad_clamp_f64 *cfg, size_t n, void *state, const double *x, double *y,
                              size_t lanes, size_t frames, int layout, void *stream)
{
    return entry_f64<Df1F64<true>, idsp_biquad_clamp_f64, FillClampF64>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

int idsp_cascade_f64_df1(const idsp_biquad_f64 *cfg, size_t n, void *state, const double *x, double *y,
                         size_t lanes, size_t frames, int layout, void *stream)
{
    int rc = check_stream_args(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    return run_cascade<double>(FillF64{cfg}, n, state, x, y, lanes, frames, layout, as_stream(stream));
}

// explicit row pitches (include/idsp_hip.h, "_pitch" entries)
IDSP_PITCH_TWIN(idsp_biquad_f64_df1, idsp_biquad_f64, double, entry_f64, Df1F64<false>, idsp_biquad_f64, FillF64)
IDSP_PITCH_TWIN(idsp_biquad_f64_df1_clamp, idsp_biquad_clamp_f64, double, entry_f64, Df1F64<true>, idsp_biquad_clamp_f64, FillClampF64)

int idsp_cascade_f64_df1_pitch(const idsp_biquad_f64 *cfg, size_t n, void *state, const double *x, size_t x_pitch, double *y,
                               size_t y_pitch, size_t lanes, size_t frames, int layout, void *stream)
{
    int rc = check_stream_args(cfg, n, state, x, y, lanes, frames, layout);
    if (rc) return rc;
    return run_cascade<double>(FillF64{cfg}, n, state, x, y, lanes, frames, layout, as_stream(stream), Pitch{x_pitch, y_pitch});
}

}  // extern "C"
