// biquad_i32_wide.hip — C-ABI entry points (include/idsp_hip.h) of this family; device code in biquad_sections.h.
#include "biquad_sections.h"

using namespace idsp;
using namespace idsp::bq;

extern "C" {

int idsp_biquad_i32_wide(const idsp_biquad_i32 *cfg, size_t n, void *state, const int32_t *x, int32_t *y,
                         size_t lanes, size_t frames, int layout, void *stream)
{
    return entry_i32<WideI32<false>, idsp_biquad_i32, FillI32>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

int idsp_biquad_i32_wide_clamp(const idsp_biquad_clamp_i32 *cfg, size_t n, void *state, const int32_t *x,
                               int32_t *y, size_t lanes, size_t frames, int layout, void *stream)
{
    return entry_i32<WideI32<true>, idsp_biquad_clamp_i32, FillClampI32>(cfg, n, state, x, y, lanes, frames, layout, stream);
}

// explicit row pitches (include/idsp_hip.h, "_pitch" entries)
IDSP_PITCH_TWIN(idsp_biquad_i32_wide, idsp_biquad_i32, int32_t, entry_i32, WideI32<false>, idsp_biquad_i32, FillI32)
IDSP_PITCH_TWIN(idsp_biquad_i32_wide_clamp, idsp_biquad_clamp_i32, int32_t, entry_i32, WideI32<true>, idsp_biquad_clamp_i32, FillClampI32)

}  // extern "C"
