"""The algebra behind the wave-per-lane Cic kernels (idsp_amd/csrc/cic_ring.h), checked on the CPU against the serial
definition (`oracle/spec.py` Cic = src/cic.rs:160-207): the wrapping integrators are linear maps over Z / 2^W, a chunk of
R samples maps the integrator vector z -> A^R z + L with A^n Toeplitz (entries C(n + d - 1, d)), A^(R 2^k) comes from
repeated squaring of truncated polynomials, six Hillis-Steele steps over 64 chunks leave the true state behind every
chunk, and the ticked output is the chunk-local value plus the sum of the state the chunk started from.  The model below
is the kernels' schedule statement by statement in Python integers; no GPU, no library."""
from math import comb

import numpy as np
import pytest

from oracle import spec

W = 64  # chunks per block = threads per wave


def poly_mul(a, b, n, mask):
    r = [0] * n
    for i in range(n):
        for j in range(n - i):
            r[i + j] = (r[i + j] + a[i] * b[j]) & mask
    return r


def poly_pow(e, n, mask):
    base, r = [1] * n, [1] + [0] * (n - 1)
    while e:
        if e & 1:
            r = poly_mul(r, base, n, mask)
        base = poly_mul(base, base, n, mask)
        e >>= 1
    return r


def scan_coef(R, n, mask):
    g, out = poly_pow(R, n, mask), []
    for _ in range(6):
        out.append(g)
        g = poly_mul(g, g, n, mask)
    return out


def test_toeplitz_powers_are_binomials():
    mask = (1 << 64) - 1
    for n_steps in (1, 2, 7, 16, 1000):
        g = poly_pow(n_steps, 6, mask)
        assert g == [comb(n_steps + d - 1, d) & mask for d in range(6)]


def scan(z, g6, n, mask):
    """z[t] = per-chunk maps' offsets; afterwards z[t] = state behind chunk t"""
    for k in range(6):
        d, g = 1 << k, g6[k]
        w = [z[t - d] if t >= d else [0] * n for t in range(W)]
        z = [[(z[t][i] + sum(g[i - j] * w[t][j] for j in range(i + 1))) & mask for i in range(n)] for t in range(W)]
    return z


def model_decimate(x, R, n, m, bits, state):
    """x: the lane's samples; state = (zoh, combs[n][m], integrators[n]).  Returns outputs, new state."""
    mask = (1 << bits) - 1
    zoh, combs, S = state
    g6 = scan_coef(R, n, mask)
    frames = len(x) // R
    cprev = [[0] * W for _ in range(n)]
    for i in range(n):
        for j in range(m):
            cprev[i][W - m + j] = combs[i][j]
    cold = [row[:] for row in cprev]
    out, nlast = [], 0
    for c in range((frames + W - 1) // W):
        nlast = min(W, frames - c * W)
        z, u = [], []
        for t in range(W):
            zt = list(S) if t == 0 else [0] * n
            ut = 0
            for s in range(R):
                idx = (c * W + t) * R + s
                v = x[idx] & mask if idx < len(x) else 12345  # chunks past the end hold anything
                for i in range(n):
                    zt[i] = (zt[i] + v) & mask
                    v = zt[i]
                if s == 0:
                    ut = v
            z.append(zt)
            u.append(ut)
        z = scan(z, g6, n, mask)
        u = [(u[t] + (sum(z[t - 1]) if t else 0)) & mask for t in range(W)]
        S = z[nlast - 1]
        for i in range(n):
            d = [u[t - m] if t >= m else cprev[i][(t - m) % W] for t in range(W)]
            cold[i], cprev[i] = cprev[i], u[:]
            u = [(u[t] - d[t]) & mask for t in range(W)]
        out += u[:nlast]
        zoh = u[nlast - 1]
    new_combs = [[(cprev[i][nlast + j - m] if nlast + j - m >= 0 else cold[i][W + nlast + j - m]) for j in range(m)] for i in range(n)]
    return out, (zoh, new_combs, list(S))


def model_interpolate(x, R, n, m, bits, state):
    mask = (1 << bits) - 1
    zoh, combs, S = state
    g6 = scan_coef(R, n, mask)
    h = [0] * n  # the chain after R steps on constant input 1 from zero
    for _ in range(R):
        v = 1
        for i in range(n):
            h[i] = (h[i] + v) & mask
            v = h[i]
    frames = len(x)
    cprev = [[0] * W for _ in range(n)]
    for i in range(n):
        for j in range(m):
            cprev[i][W - m + j] = combs[i][j]
    cold = [row[:] for row in cprev]
    out, nlast = [], 0
    for c in range((frames + W - 1) // W):
        nlast = min(W, frames - c * W)
        v = [x[c * W + t] & mask if c * W + t < frames else 0 for t in range(W)]
        for i in range(n):
            d = [v[t - m] if t >= m else cprev[i][(t - m) % W] for t in range(W)]
            cold[i], cprev[i] = cprev[i], v[:]
            v = [(v[t] - d[t]) & mask for t in range(W)]
        zoh = v[nlast - 1]
        ars = [(S[i] + sum(g6[0][i - j] * S[j] for j in range(i))) & mask for i in range(n)]
        z = [[((ars[i] if t == 0 else 0) + h[i] * v[t]) & mask for i in range(n)] for t in range(W)]
        z = scan(z, g6, n, mask)
        for t in range(nlast):
            p = list(S) if t == 0 else list(z[t - 1])
            for _ in range(R):
                a = v[t]
                for i in range(n):
                    p[i] = (p[i] + a) & mask
                    a = p[i]
                out.append(a)
        S = z[nlast - 1]
    new_combs = [[(cprev[i][nlast + j - m] if nlast + j - m >= 0 else cold[i][W + nlast + j - m]) for j in range(m)] for i in range(n)]
    return out, (zoh, new_combs, list(S))


def spec_state(c):
    return c.zoh, [list(row) for row in c.combs], list(c.integrators)


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("kind", ["dec", "int"])
def test_scan_model_equals_the_serial_definition(kind, bits):
    rng = np.random.default_rng(bits + (1 if kind == "dec" else 0))
    mask = (1 << bits) - 1
    for n, m, R, frames in [(1, 1, 4, 64), (3, 1, 16, 65), (3, 2, 8, 130), (6, 4, 4, 1), (4, 3, 2, 3), (2, 4, 16, 200), (5, 1, 32, 70)]:
        ref = spec.Cic(n, m, R - 1, bits)
        # arbitrary starting state
        ref.zoh = int(rng.integers(0, 1 << 31))
        ref.combs = [[int(rng.integers(0, 1 << 31)) for _ in range(m)] for _ in range(n)]
        ref.integrators = [int(rng.integers(0, 1 << 31)) for _ in range(n)]
        state = spec_state(ref)
        for part in range(2):
            cnt = frames * (R if kind == "dec" else 1)
            x = [int(v) for v in rng.integers(-(1 << (bits - 1)), (1 << (bits - 1)) - 1, size=cnt, dtype=np.int64 if bits == 64 else np.int64)]
            if kind == "dec":
                want = [o & mask for o in (ref.decimate(v) for v in x) if o is not None]
                got, state = model_decimate(x, R, n, m, bits, state)
            else:
                want = [ref.interpolate(x[f] if k == 0 else None) & mask for f in range(frames) for k in range(R)]
                got, state = model_interpolate(x, R, n, m, bits, state)
            assert got == want, (n, m, R, frames, part)
            zoh, combs, integ = spec_state(ref)
            assert state[0] == zoh & mask and state[2] == [v & mask for v in integ], (n, m, R, frames, part)
            assert state[1] == [[v & mask for v in row] for row in combs], (n, m, R, frames, part)
