// cic_dec_i64.hip — C-ABI entry point idsp_cic_dec_i64 (include/idsp_hip.h); device code in cic_kernels.h.
#include "cic_kernels.h"

extern "C" int idsp_cic_dec_i64(const idsp_cic *cfg, void *state, const int64_t *x, int64_t *y, size_t lanes, size_t frames, int layout,
                                void *stream)
{
    return idsp::cic::run<int64_t, true>(cfg, state, x, y, lanes, frames, layout, stream);
}
