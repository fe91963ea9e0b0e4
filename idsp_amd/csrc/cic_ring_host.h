// cic_ring_host.h — host-side coefficient helpers of the wave-per-lane Cic kernels (cic_ring.h).
#pragma once

#include "cic_ring.h"

namespace idsp {
namespace cicr_host {

// g_d(n) = (A^n)[d][0], d = 0 .. N-1, for n = R 2^k: Toeplitz triangles multiply like polynomials truncated to N terms;
// A itself is all ones.  u64 wrapping arithmetic — the low 32 bits are the i32 coefficients.
struct Poly {
    uint64_t c[IDSP_CIC_MAX_ORDER];
};
inline Poly mul(const Poly &a, const Poly &b, int n)
{
    Poly r{};
    for (int i = 0; i < n; i++)
        for (int j = 0; i + j < n; j++) r.c[i + j] += a.c[i] * b.c[j];
    return r;
}
inline Poly power(uint64_t e, int n)
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

// scan coefficients g_d(R 2^k), k = 0 .. 5
template <class T, int N>
inline cicr::ScanCoef<T, N> scan_coef(size_t R)
{
    cicr::ScanCoef<T, N> coef{};
    Poly g = power(R, N);
    for (int k = 0; k < cicr::kSteps; k++) {
        for (int d = 1; d < N; d++) coef.g[k][d - 1] = static_cast<typename std::make_unsigned<T>::type>(g.c[d]);
        g = mul(g, g, N);
    }
    return coef;
}

// whole 16-byte pieces per chunk (1, 2, 4 or 8 of them), at least one whole block of 64 chunks (below that the
// lane-per-thread kernels have as much parallelism and less to set up), lanes within the grid limit; `hi` = the
// high-rate tensor.  Returns the pieces per chunk, or 0.
template <class T>
inline int ring_pieces(const idsp_cic *cfg, const void *hi, size_t lanes, size_t frames)
{
    const size_t fb = (size_t(cfg->rate) + 1) * sizeof(T);
    if (no_ring() || fb % 16 != 0 || fb > 128 || (fb & (fb - 1)) != 0 || frames < size_t(cicr::kW) || reinterpret_cast<uintptr_t>(hi) % 16 != 0 ||
        lanes > 0x7fffffffu)
        return 0;
    return int(fb / 16);
}

}  // namespace cicr_host
}  // namespace idsp
