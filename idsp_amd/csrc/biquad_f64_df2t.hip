// biquad_f64_df2t.hip — C-ABI entry points (include/idsp_hip.h) of the f64 DF2T biquads; device code in biquad_sections.h.
#include "biquad_sections.h"

using namespace idsp;
using namespace idsp::bq;

extern "C" {

int idsp_biquad_f64_df2t(const idsp_biquad_f64 *cfg, size_t n, void *state, const double *x, double *y,
                         size_t lanes, size_t frames, int layout, void *stream)
{
    return entry_f64<Df2tF64<false>, idsp_biquad_f64, FillF64>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

int idsp_biquad_f64_df2t_clamp(const idsp_biquad_clamp_f64 *cfg, size_t n, void *state, const double *x, double *y,
                               size_t lanes, size_t frames, int layout, void *stream)
{
    return entry_f64<Df2tF64<true>, idsp_biquad_clamp_f64, FillClampF64>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

// explicit row pitches (include/idsp_hip.h, "_pitch" entries)
IDSP_PITCH_TWIN(idsp_biquad_f64_df2t, idsp_biquad_f64, double, entry_f64, Df2tF64<false>, idsp_biquad_f64, FillF64)
IDSP_PITCH_TWIN(idsp_biquad_f64_df2t_clamp, idsp_biquad_clamp_f64, double, entry_f64, Df2tF64<true>, idsp_biquad_clamp_f64, FillClampF64)

}  // extern "C"
