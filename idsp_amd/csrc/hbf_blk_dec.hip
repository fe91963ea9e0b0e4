// hbf_blk_dec.hip — instantiates the register-blocked half-band decimator cascades of hbf_blk.h.
#include "hbf_blk.h"

namespace idsp {
namespace {
template <int TS>
int launch_s(int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames, bool lm, hipStream_t stream)
{
    switch (stages) {
        case 1: return hbfb::launch_blk<TS, 1>(st, x, y, lanes, frames, lm, stream);
        case 2: return hbfb::launch_blk<TS, 2>(st, x, y, lanes, frames, lm, stream);
        case 3: return hbfb::launch_blk<TS, 3>(st, x, y, lanes, frames, lm, stream);
        case 4: return hbfb::launch_blk<TS, 4>(st, x, y, lanes, frames, lm, stream);
        case 5: return hbfb::launch_blk<TS, 5>(st, x, y, lanes, frames, lm, stream);
        default: return 1;
    }
}
}  // namespace

int hbf_blk_dec(int tap_set, int stages, uint32_t *st, const float *x, float *y, size_t lanes, size_t frames,
                bool lane_major, hipStream_t stream)
{
    if (tap_set == 0) return launch_s<0>(stages, st, x, y, lanes, frames, lane_major, stream);
    if (tap_set == 1) return launch_s<1>(stages, st, x, y, lanes, frames, lane_major, stream);
    return 1;
}
}  // namespace idsp
