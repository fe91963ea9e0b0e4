// cic.hip — host helpers of the Cic entries (include/idsp_hip.h); kernels in cic_kernels.h, one
// translation unit per (direction, sample type): cic_{dec,int}_{i32,i64}.hip.
#include "cic_kernels.h"

using namespace idsp;
using namespace idsp::cic;

extern "C" {

int64_t idsp_cic_gain(const idsp_cic *cfg)
{
    if (cfg_check(cfg)) return 0;
    // (M * (rate + 1)).pow(N) in i64, wrapping (src/cic.rs:103-105)
    const uint64_t b = uint64_t(cfg->comb_delay) * (uint64_t(cfg->rate) + 1);
    uint64_t g = 1;
    for (int i = 0; i < cfg->order; i++) g *= b;
    return int64_t(g);
}

int idsp_cic_gain_log2(const idsp_cic *cfg)
{
    const int rc = cfg_check(cfg);
    if (rc) return rc;
    // (u32::BITS - (M * rate + (M - 1)).leading_zeros()) * N  (src/cic.rs:111-113; u32 arithmetic)
    const uint32_t v = uint32_t(cfg->comb_delay) * cfg->rate + uint32_t(cfg->comb_delay - 1);
    const int bits = v ? 32 - __builtin_clz(v) : 0;
    return bits * cfg->order;
}

size_t idsp_cic_response_length(const idsp_cic *cfg)
{
    if (cfg_check(cfg)) return 0;
    return size_t(cfg->rate) * size_t(cfg->order);  // src/cic.rs:116-118
}

size_t idsp_cic_state_words(const idsp_cic *cfg, int bits)
{
    if (cfg_check(cfg) || (bits != 32 && bits != 64)) return 0;
    return size_t(1 + cfg->order * cfg->comb_delay + cfg->order) * size_t(bits / 32);
}

}  // extern "C"
