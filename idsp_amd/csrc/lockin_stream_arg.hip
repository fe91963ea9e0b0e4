// lockin_stream_arg.hip — lock-in with the `arg` read-out fused, on the stream kernels (lockin_stream_procs.h; one translation unit per read-out so that they compile in parallel).
#include "lockin_stream_procs.h"

namespace idsp {

int lockin_stream_arg(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, hipStream_t s, size_t pitch)
{
    return dispatch_nk<LockinArgProc, int32_t>(cfg, state, x, y, lanes, frames, layout, s, pitch);
}

}  // namespace idsp
