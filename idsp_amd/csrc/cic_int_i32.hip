// cic_int_i32.hip — C-ABI entry point idsp_cic_int_i32 (include/idsp_hip.h); device code in cic_kernels.h.
#include "cic_kernels.h"

extern "C" int idsp_cic_int_i32(const idsp_cic *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout,
                                void *stream)
{
    return idsp::cic::run<int32_t, false>(cfg, state, x, y, lanes, frames, layout, stream);
}
