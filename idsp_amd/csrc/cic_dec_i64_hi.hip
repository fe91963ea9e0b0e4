// cic_dec_i64_hi.hip — kernels of idsp_cic_dec_i64 for orders 4..6 (entry point in cic_dec_i64.hip).
#include "cic_kernels.h"

namespace idsp {
namespace cic {
template int run_orders<int64_t, true, 4>(const idsp_cic *, void *, const int64_t *, int64_t *, size_t, size_t, int, void *);
}  // namespace cic
}  // namespace idsp
