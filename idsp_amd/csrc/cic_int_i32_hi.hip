// cic_int_i32_hi.hip — kernels of idsp_cic_int_i32 for orders 4..6 (entry point in cic_int_i32.hip).
#include "cic_kernels.h"

namespace idsp {
namespace cic {
template int run_orders<int32_t, false, 4>(const idsp_cic *, void *, const int32_t *, int32_t *, size_t, size_t, int, void *);
}  // namespace cic
}  // namespace idsp
