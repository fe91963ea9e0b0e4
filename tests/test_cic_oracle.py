"""Cic on the CPU oracle vs the independent Python restatement (oracle/spec.py `Cic`, itself checked
against the modular Integrator/Comb composition the reference's tests assert, src/cic.rs:348-383):
outputs and written-back state, both layouts, i32 and i64, chunked continuation."""
import ctypes as C

import numpy as np
import pytest

from idsp_amd import _abi
from oracle import spec
from tests import _cic_cases as K
from tests._backends import OracleBackend


@pytest.fixture(scope="module")
def ob(oracle_lib):
    return OracleBackend()


def _spec_state(c: spec.Cic, bits):
    vals = [c.zoh] + [v for row in c.combs for v in row] + list(c.integrators)
    words = []
    for v in vals:
        u = v & ((1 << bits) - 1)
        words += [u & 0xFFFFFFFF] + ([u >> 32] if bits == 64 else [])
    return words


@pytest.mark.parametrize("dtype", [np.int32, np.int64], ids=["i32", "i64"])
@pytest.mark.parametrize("kind", ["dec", "int"])
@pytest.mark.parametrize("layout", [K.FM, K.LM])
def test_cic_oracle_equals_spec(ob, kind, dtype, layout):
    bits = 64 if dtype == np.int64 else 32
    rng = np.random.default_rng(bits + layout + (7 if kind == "dec" else 0))
    for n, m, rate in K.CONFIGS:
        cfg = _abi.Cic(n, m, rate)
        R = rate + 1
        lanes, frames = 3, 11
        words = K.state_words(ob, cfg, dtype)
        assert words == (1 + n * m + n) * (bits // 32)
        st = np.zeros((words, lanes), np.uint32)
        models = [spec.Cic(n, m, rate, bits) for _ in range(lanes)]
        for part in range(2):  # second call continues from the written-back state
            n_in = lanes * frames * (R if kind == "dec" else 1)
            x = K.samples(rng, dtype, n_in)
            rc, y = K.run(ob, kind, dtype, cfg, st, x, lanes, frames, layout)
            assert rc == 0
            hi = (x if kind == "dec" else y).reshape((frames, lanes, R) if layout == K.FM else (lanes, frames, R))
            lo = (y if kind == "dec" else x).reshape((frames, lanes) if layout == K.FM else (lanes, frames))
            for l, c in enumerate(models):
                for f in range(frames):
                    chunk = hi[f, l] if layout == K.FM else hi[l, f]
                    lov = int(lo[f, l] if layout == K.FM else lo[l, f])
                    if kind == "dec":
                        outs = [c.decimate(int(v)) for v in chunk]
                        assert [o for o in outs if o is not None] == [lov] and outs[0] is not None
                    else:
                        assert [c.interpolate(lov if k == 0 else None) for k in range(R)] == [int(v) for v in chunk]
                assert c.index == 0
                assert st[:, l].tolist() == _spec_state(c, bits), (n, m, rate, part)


def test_cic_helpers_and_errors(ob):
    for n, m, rate in K.CONFIGS + [(3, 1, 0xFFFFFFFF)]:
        cfg = _abi.Cic(n, m, rate)
        c = spec.Cic(n, m, rate)
        assert ob.helper("cic_gain", C.byref(cfg)) == c.gain()
        assert ob.helper("cic_gain_log2", C.byref(cfg)) == (((m * rate + m - 1) & 0xFFFFFFFF).bit_length()) * n
        assert ob.helper("cic_response_length", C.byref(cfg)) == c.response_length()
    x = np.zeros(8, np.int32)
    st = np.zeros((16, 1), np.uint32)
    for bad in [(0, 1, 1), (7, 1, 1), (3, 0, 1), (3, 5, 1)]:  # cic.rs:36 M > 0; ABI limits N <= 6, M <= 4
        cfg = _abi.Cic(*bad)
        assert K.run(ob, "dec", np.int32, cfg, st, x, 1, 4, K.LM)[0] < 0
        assert ob.helper("cic_state_words", C.byref(cfg), 32) == 0
    assert ob.helper("cic_state_words", C.byref(_abi.Cic(3, 1, 1)), 16) == 0
