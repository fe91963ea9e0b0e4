// lockin_waves_lo.hip — `Lockin<[Lowpass<N>; K]>` fed by a per-sample oscillator (src/lockin.rs:17-27; idsp_lockin_i32_lo_process)
// on the multi-wave lock-in kernel: `LpBank<N, K>` with the LO taken from the caller's buffer instead of `Accu` -> cossin.
#include "lockin_waves.h"

namespace idsp {
namespace {

template <int N, int K>
struct LpLoBank : LpBank<N, K> {
    static constexpr bool kExtLo = true, kSixWaves = false;
    static const char *name() { return "[Lowpass<N>; K], LO"; }
};

}  // namespace

int lockin_waves_lowpass_lo(const idsp_lockin_i32 *cfg, void *state, const int32_t *x, const int32_t *lo, int32_t *y, size_t lanes, size_t frames,
                            int layout, hipStream_t s, size_t pitch)
{
    const LpParams p = lp_params(cfg);
#define IDSP_CASE(N, K)                                                                                                                    \
    if (cfg->order == N && cfg->cascade == K)                                                                                               \
    return launch_lockin_waves_bank<MODE_IQ, LpLoBank<N, K>>(p, static_cast<uint32_t *>(state), x, reinterpret_cast<Cplx *>(y), lanes, frames, \
                                                             layout, 4, s, lo, pitch)
    IDSP_CASE(1, 1);
    IDSP_CASE(1, 2);
    IDSP_CASE(1, 3);
    IDSP_CASE(1, 4);
    IDSP_CASE(2, 1);
    IDSP_CASE(2, 2);
    IDSP_CASE(2, 3);
    IDSP_CASE(2, 4);
#undef IDSP_CASE
    return fail(IDSP_EINVAL, "unsupported lowpass configuration");
}

}  // namespace idsp
