// lowpass.hip — `[Lowpass<N>; K]` (src/lowpass.rs:47-78, idsp_lowpass_i32) on the stream kernels (lockin_stream_procs.h; one translation unit per read-out so that they compile in parallel).
#include "lockin_stream_procs.h"

namespace idsp {

int lowpass_stream(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, hipStream_t s)
{
    return dispatch_nk<LowpassProc, int32_t>(cfg, state, x, y, lanes, frames, layout, s);
}

}  // namespace idsp
