// cic_ring_dec.hip — host side and instantiations of the wave-per-lane Cic decimator (cic_ring.h).
#include "cic_ring.h"

namespace idsp {
namespace {

using namespace cicr;

// g_d(n) = (A^n)[d][0], d = 0 .. N-1, for n = R 2^k: Toeplitz triangles multiply like polynomials truncated to N terms;
// A itself is all ones.  u64 wrapping arithmetic — the low 32 bits are the i32 coefficients.
struct Poly {
    uint64_t c[IDSP_CIC_MAX_ORDER];
};
Poly mul(const Poly &a, const Poly &b, int n)
{
    Poly r{};
    for (int i = 0; i < n; i++)
        for (int j = 0; i + j < n; j++) r.c[i + j] += a.c[i] * b.c[j];
    return r;
}
Poly power(uint64_t e, int n)
{
    Poly base{}, r{};
    for (int i = 0; i < n; i++) base.c[i] = 1;
    r.c[0] = 1;
    while (e) {
        if (e & 1) r = mul(r, base, n);
        base = mul(base, base, n);
        e >>= 1;
    }
    return r;
}

inline bool no_ring()
{
    static const bool v = diag_env("IDSP_CIC_NO_RING") != nullptr;
    return v;
}

template <class T, int N, int PPT>
int launch(const idsp_cic *cfg, uint32_t *st, const T *x, T *y, size_t lanes, size_t frames, hipStream_t stream)
{
    constexpr size_t R = size_t(PPT) * 16 / sizeof(T);
    ScanCoef<T, N> coef{};
    Poly g = power(R, N);
    for (int k = 0; k < kSteps; k++) {
        for (int d = 1; d < N; d++) coef.g[k][d - 1] = static_cast<typename std::make_unsigned<T>::type>(g.c[d]);
        g = mul(g, g, N);
    }
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
    // whole 16-byte pieces per frame on a 16-byte aligned tensor, at least one whole block of 64 frames (below that the
    // lane-per-thread kernels have as much parallelism and less to set up), lanes within the grid limit
    const size_t fb = (size_t(cfg->rate) + 1) * sizeof(T);
    if (no_ring() || fb % 16 != 0 || fb > 128 || (fb & (fb - 1)) != 0 || frames < size_t(kW) || reinterpret_cast<uintptr_t>(x) % 16 != 0 ||
        lanes > 0x7fffffffu)
        return 1;
    const int ppt = int(fb / 16);
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
