// lockin_waves_iq.hip — the MODE_IQ instantiations of lockin_waves.h (one translation unit per read-out so the
// three sets of 32 kernels compile in parallel).
#include "lockin_waves.h"

namespace idsp {

int lockin_waves_iq(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, void *y, size_t lanes, size_t frames, int layout,
                   int waves, hipStream_t s, size_t pitch)
{
    return launch_lockin_waves<MODE_IQ>(cfg, state, x, y, lanes, frames, layout, waves, s, pitch);
}

}  // namespace idsp
