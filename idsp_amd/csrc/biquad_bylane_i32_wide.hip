// biquad_bylane_i32_wide.hip — (second half of biquad_bylane_i32.hip: the clamped dither and the wide sections) C-ABI entry points (include/idsp_hip.h) of the per-lane-coefficient i32 biquads
// (`ByLane<[Biquad<Q32<F>>; N]>`, dsp-process/src/compose.rs:363-390); device code in biquad_sections.h.
#include "biquad_sections.h"

using namespace idsp;
using namespace idsp::bq;

#define IDSP_BYLANE_I32(name, sec)                                                                              \
    int idsp_biquad_i32_##name##_bylane(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, \
                                        int32_t *y, size_t lanes, size_t frames, int layout, void *stream)      \
    {                                                                                                           \
        return entry_bylane<sec>(coef, frac, n, state, x, y, lanes, frames, layout, stream);                    \
    }                                                                                                           \
    int idsp_biquad_i32_##name##_bylane_pitch(const int32_t *coef, int frac, size_t n, void *state, const int32_t *x, size_t x_pitch, \
                                              int32_t *y, size_t y_pitch, size_t lanes, size_t frames, int layout, void *stream)      \
    {                                                                                                           \
        return entry_bylane<sec>(coef, frac, n, state, x, y, lanes, frames, layout, stream, Pitch{x_pitch, y_pitch}); \
    }

extern "C" {
IDSP_BYLANE_I32(dither_clamp, DitherI32<true>)
IDSP_BYLANE_I32(wide, WideI32<false>)
IDSP_BYLANE_I32(wide_clamp, WideI32<true>)
}  // extern "C"
