// hbf_wave_dec.hip — instantiates the specialised decimator cascades of hbf_wave.h.
#include "hbf_wave.h"

namespace idsp {
int hbf_wave_dec(int tap_set, int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames,
                 bool lane_major, hipStream_t stream)
{
    if (tap_set == 0) return hbfw::launch_wave_s<0, true>(stages, st, x, y, lanes, frames, lane_major, stream);
    if (tap_set == 1) return hbfw::launch_wave_s<1, true>(stages, st, x, y, lanes, frames, lane_major, stream);
    return 1;
}
}  // namespace idsp
