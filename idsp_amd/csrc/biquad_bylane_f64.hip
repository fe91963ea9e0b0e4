// biquad_bylane_f64.hip — C-ABI entry points (include/idsp_hip.h) of the per-lane-coefficient f64 biquads
// (`ByLane<[Biquad<f32>; N]>`, dsp-process/src/compose.rs:363-390); device code in biquad_sections.h.
#include "biquad_sections.h"

using namespace idsp;
using namespace idsp::bq;

#define IDSP_BYLANE_F(ty, tn, name, sec)                                                                      \
    int idsp_biquad_##tn##_##name##_bylane(const ty *coef, size_t n, void *state, const ty *x, ty *y,         \
                                           size_t lanes, size_t frames, int layout, void *stream)             \
    {                                                                                                         \
        return entry_bylane<sec>(coef, 0, n, state, x, y, lanes, frames, layout, stream);                     \
    }                                                                                                         \
    int idsp_biquad_##tn##_##name##_bylane_pitch(const ty *coef, size_t n, void *state, const ty *x, size_t x_pitch, ty *y, \
                                                 size_t y_pitch, size_t lanes, size_t frames, int layout, void *stream)     \
    {                                                                                                         \
        return entry_bylane<sec>(coef, 0, n, state, x, y, lanes, frames, layout, stream, Pitch{x_pitch, y_pitch}); \
    }

extern "C" {
IDSP_BYLANE_F(double, f64, df1, Df1F64<false>)
IDSP_BYLANE_F(double, f64, df1_clamp, Df1F64<true>)
IDSP_BYLANE_F(double, f64, df2t, Df2tF64<false>)
IDSP_BYLANE_F(double, f64, df2t_clamp, Df2tF64<true>)
}  // extern "C"
