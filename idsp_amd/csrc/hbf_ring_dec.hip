// hbf_ring_dec.hip — instantiates the LDS-DMA ring decimator cascades of hbf_ring.h (FrameMajor /16).
#include "hbf_ring.h"

namespace idsp {
namespace {
template <int TS>
int launch_s(int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames, bool lm, hipStream_t stream)
{
    return stages == 4 ? hbfr::launch_ring<TS, 4>(st, x, y, lanes, frames, lm, stream) : 1;
}
}  // namespace

int hbf_ring_dec(int tap_set, int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames,
                 bool lane_major, hipStream_t stream)
{
    if (tap_set == 0) return launch_s<0>(stages, st, x, y, lanes, frames, lane_major, stream);
    if (tap_set == 1) return launch_s<1>(stages, st, x, y, lanes, frames, lane_major, stream);
    return 1;
}
}  // namespace idsp
