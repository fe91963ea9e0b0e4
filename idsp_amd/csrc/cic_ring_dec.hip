// cic_ring_dec.hip — host side and instantiations of the wave-per-lane Cic decimator (cic_ring.h).
#include "cic_ring_host.h"

namespace idsp {
namespace {

using namespace cicr;
using namespace cicr_host;

template <class T, int N, int PPT>
int launch(const idsp_cic *cfg, uint32_t *st, const T *x, T *y, size_t lanes, size_t frames, hipStream_t stream)
{
    const ScanCoef<T, N> coef = scan_coef<T, N>(size_t(PPT) * 16 / sizeof(T));
    constexpr size_t bytes = 2 * size_t(PPT) * 1024;
    if (ensure_dyn_lds<&cic_dec_ring_lm<T, N, PPT>>(bytes)) return 2;
    note_kernel("cic_dec_ring[LaneMajor]");
    hipLaunchKernelGGL((cic_dec_ring_lm<T, N, PPT>), dim3(unsigned(lanes)), dim3(kW), bytes, stream, coef, int(cfg->comb_delay), st, x, y, lanes,
                       frames);
    return 0;
}

template <class T, int N>
int by_width(int ppt, const idsp_cic *cfg, uint32_t *st, const T *x, T *y, size_t lanes, size_t frames, hipStream_t stream)
{
    switch (ppt) {
        case 1: return launch<T, N, 1>(cfg, st, x, y, lanes, frames, stream);
        case 2: return launch<T, N, 2>(cfg, st, x, y, lanes, frames, stream);
        case 4: return launch<T, N, 4>(cfg, st, x, y, lanes, frames, stream);
        case 8: return launch<T, N, 8>(cfg, st, x, y, lanes, frames, stream);
    }
    return 1;
}

template <class T>
int dispatch(const idsp_cic *cfg, uint32_t *st, const T *x, T *y, size_t lanes, size_t frames, hipStream_t stream)
{
    const int ppt = ring_pieces<T>(cfg, x, lanes, frames);
    if (!ppt) return 1;
    switch (cfg->order) {
        case 1: return by_width<T, 1>(ppt, cfg, st, x, y, lanes, frames, stream);
        case 2: return by_width<T, 2>(ppt, cfg, st, x, y, lanes, frames, stream);
        case 3: return by_width<T, 3>(ppt, cfg, st, x, y, lanes, frames, stream);
        case 4: return by_width<T, 4>(ppt, cfg, st, x, y, lanes, frames, stream);
        case 5: return by_width<T, 5>(ppt, cfg, st, x, y, lanes, frames, stream);
        case 6: return by_width<T, 6>(ppt, cfg, st, x, y, lanes, frames, stream);
    }
    return 1;
}

}  // namespace

int cic_ring_dec(const idsp_cic *cfg, uint32_t *st, const int32_t *x, int32_t *y, size_t lanes, size_t frames, hipStream_t stream)
{
    return dispatch<int32_t>(cfg, st, x, y, lanes, frames, stream);
}
int cic_ring_dec(const idsp_cic *cfg, uint32_t *st, const int64_t *x, int64_t *y, size_t lanes, size_t frames, hipStream_t stream)
{
    return dispatch<int64_t>(cfg, st, x, y, lanes, frames, stream);
}

}  // namespace idsp
