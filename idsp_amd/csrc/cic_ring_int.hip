// cic_ring_int.hip — host side and instantiations of the wave-per-lane Cic interpolator (cic_ring.h).
#include "cic_ring_host.h"

namespace idsp {
namespace {

using namespace cicr;
using namespace cicr_host;

template <class T, int N, int PPT>
int launch(const idsp_cic *cfg, uint32_t *st, const T *x, T *y, size_t lanes, size_t frames, bool lm, hipStream_t stream)
{
    using UT = typename std::make_unsigned<T>::type;
    constexpr size_t R = size_t(PPT) * 16 / sizeof(T);
    IntCoef<T, N> coef{};
    coef.scan = scan_coef<T, N>(R);
    // h = the chain after R steps on constant input 1 from zero (src/cic.rs:174-180), wrapping
    uint64_t z[N] = {};
    for (size_t s = 0; s < R; s++) {
        uint64_t v = 1;
        for (int n = 0; n < N; n++) {
            z[n] += v;
            v = z[n];
        }
    }
    for (int n = 0; n < N; n++) coef.h[n] = UT(z[n]);
    if (!lm) {
        constexpr size_t fbytes = size_t(kW) * (kFmLanes * PPT + 1) * 16 + size_t(kW) * (kFmLanes + 1) * sizeof(T);
        static_assert(fbytes <= 160 * 1024, "one workgroup per CU at least");
        if (ensure_dyn_lds<&cic_int_ring_fm<T, N, PPT>>(fbytes)) return 2;
        const size_t ngroups = lanes / kFmLanes;
        note_kernel("cic_int_ring[FrameMajor]");
        hipLaunchKernelGGL((cic_int_ring_fm<T, N, PPT>), dim3(unsigned(8 * ((ngroups + 7) / 8))), dim3(kFmLanes * kW), fbytes, stream, coef,
                           int(cfg->comb_delay), st, x, y, lanes, frames);
        return 0;
    }
    constexpr size_t bytes = size_t(kW) * (PPT + 1) * 16;
    if (ensure_dyn_lds<&cic_int_ring_lm<T, N, PPT>>(bytes)) return 2;
    note_kernel("cic_int_ring[LaneMajor]");
    hipLaunchKernelGGL((cic_int_ring_lm<T, N, PPT>), dim3(unsigned(lanes)), dim3(kW), bytes, stream, coef, int(cfg->comb_delay), st, x, y, lanes,
                       frames);
    return 0;
}

template <class T, int N>
int by_width(int ppt, const idsp_cic *cfg, uint32_t *st, const T *x, T *y, size_t lanes, size_t frames, bool lm, hipStream_t stream)
{
    switch (ppt) {
        case 1: return launch<T, N, 1>(cfg, st, x, y, lanes, frames, lm, stream);
        case 2: return launch<T, N, 2>(cfg, st, x, y, lanes, frames, lm, stream);
        case 4: return launch<T, N, 4>(cfg, st, x, y, lanes, frames, lm, stream);
        case 8: return launch<T, N, 8>(cfg, st, x, y, lanes, frames, lm, stream);
    }
    return 1;
}

template <class T>
int dispatch(const idsp_cic *cfg, uint32_t *st, const T *x, T *y, size_t lanes, size_t frames, bool lm, hipStream_t stream)
{
    const int ppt = ring_pieces<T>(cfg, y, lanes, frames);
    if (!ppt) return 1;
    // FRAME_MAJOR: whole groups of 16 lanes whose rows of y start on 16-byte boundaries (they do: whole pieces per chunk)
    if (!lm && lanes % kFmLanes != 0) return 1;
    switch (cfg->order) {
        case 1: return by_width<T, 1>(ppt, cfg, st, x, y, lanes, frames, lm, stream);
        case 2: return by_width<T, 2>(ppt, cfg, st, x, y, lanes, frames, lm, stream);
        case 3: return by_width<T, 3>(ppt, cfg, st, x, y, lanes, frames, lm, stream);
        case 4: return by_width<T, 4>(ppt, cfg, st, x, y, lanes, frames, lm, stream);
        case 5: return by_width<T, 5>(ppt, cfg, st, x, y, lanes, frames, lm, stream);
        case 6: return by_width<T, 6>(ppt, cfg, st, x, y, lanes, frames, lm, stream);
    }
    return 1;
}

}  // namespace

int cic_ring_int(const idsp_cic *cfg, uint32_t *st, const int32_t *x, int32_t *y, size_t lanes, size_t frames, bool lane_major,
                 hipStream_t stream)
{
    return dispatch<int32_t>(cfg, st, x, y, lanes, frames, lane_major, stream);
}
int cic_ring_int(const idsp_cic *cfg, uint32_t *st, const int64_t *x, int64_t *y, size_t lanes, size_t frames, bool lane_major,
                 hipStream_t stream)
{
    return dispatch<int64_t>(cfg, st, x, y, lanes, frames, lane_major, stream);
}

}  // namespace idsp
