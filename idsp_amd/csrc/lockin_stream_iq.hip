// lockin_stream_iq.hip — `Complex<i32>` lock-in on the stream kernels: one thread per lane, or the I and Q arms on two adjacent threads (lockin_stream_procs.h; one translation unit per read-out so that they compile in parallel).
#include "lockin_stream_procs.h"

namespace idsp {

int lockin_stream_iq(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout, bool split, hipStream_t s, size_t pitch)
{
    if (split) return dispatch_nk<LockinSplitProc, int32_t>(cfg, state, x, y, 2 * lanes, frames, layout, s, pitch);
    return dispatch_nk<LockinProc, Cplx>(cfg, state, x, reinterpret_cast<Cplx *>(y), lanes, frames, layout, s, pitch);
}

}  // namespace idsp
