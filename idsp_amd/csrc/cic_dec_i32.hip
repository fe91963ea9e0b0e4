// cic_dec_i32.hip — C-ABI entry point idsp_cic_dec_i32 (include/idsp_hip.h) and its kernels for orders 1..3;
// orders 4..6 are instantiated in cic_dec_i32_hi.hip.  Device code in cic_kernels.h.
#include "cic_kernels.h"

namespace idsp {
namespace cic {
extern template int run_orders<int32_t, true, 4>(const idsp_cic *, void *, const int32_t *, int32_t *, size_t, size_t, int, void *);
}  // namespace cic
}  // namespace idsp

extern "C" int idsp_cic_dec_i32(const idsp_cic *cfg, void *state, const int32_t *x, int32_t *y, size_t lanes, size_t frames, int layout,
                                void *stream)
{
    return idsp::cic::run<int32_t, true>(cfg, state, x, y, lanes, frames, layout, stream);
}
