// biquad_f32_df2t.hip — C-ABI entry points (include/idsp_hip.h) of this family; device code in biquad_sections.h.
#include "biquad_sections.h"

using namespace idsp;
using namespace idsp::bq;

extern "C" {

int idsp_biquad_f32_df2t(const idsp_biquad_f32 *cfg, size_t n, void *state, const float *x, float *y,
                         size_t lanes, size_t frames, int layout, void *stream)
{
    return entry_f32<Df2tF32<false>, idsp_biquad_f32, FillF32>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

int idsp_biquad_f32_df2t_clamp(const idsp_biquad_clamp_f32 *cfg, size_t n, void *state, const float *x, float *y,
                               size_t lanes, size_t frames, int layout, void *stream)
{
    return entry_f32<Df2tF32<true>, idsp_biquad_clamp_f32, FillClampF32>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

// explicit row pitches (include/idsp_hip.h, "_pitch" entries)
IDSP_PITCH_TWIN(idsp_biquad_f32_df2t, idsp_biquad_f32, float, entry_f32, Df2tF32<false>, idsp_biquad_f32, FillF32)
IDSP_PITCH_TWIN(idsp_biquad_f32_df2t_clamp, idsp_biquad_clamp_f32, float, entry_f32, Df2tF32<true>, idsp_biquad_clamp_f32, FillClampF32)

}  // extern "C"
