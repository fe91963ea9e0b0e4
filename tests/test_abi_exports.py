"""CPU-side checks of the drop-in boundary: the shared library loads without a
GPU and exports exactly the symbols include/idsp_hip.h declares; the oracle
exports a twin for every processing entry point.  No compute calls here."""
import ctypes
import os
import re

from idsp_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "idsp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(idsp_[a-z0-9_]+)\s*\(", text)))


def test_header_and_python_prototypes_agree():
    names = declared_symbols()
    assert names, "no declarations parsed"
    assert names == sorted("idsp_" + n for n in _abi.exported_names())


def test_library_loads_and_exports_every_declared_symbol():
    from idsp_amd._lib import LIB_PATH, load

    assert os.path.exists(LIB_PATH), "build the HIP engine first (make lib)"
    fn, lib = load()
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/idsp_hip.h but not exported"
    assert fn["version"]() == _abi.ABI_VERSION == 4
    assert isinstance(fn["last_error"](), bytes)


def test_structs_match_header_sizes():
    # sizes implied by the C declarations in include/idsp_hip.h
    assert ctypes.sizeof(_abi.BiquadI32) == 24
    assert ctypes.sizeof(_abi.BiquadClampI32) == 36
    assert ctypes.sizeof(_abi.BiquadF32) == 20
    assert ctypes.sizeof(_abi.BiquadClampF32) == 32
    assert ctypes.sizeof(_abi.HbfCascadeF32) == 4 + 4 * 5 + 4 * 5 * 32
    assert ctypes.sizeof(_abi.LockinI32) == 8 + 4 * 4 * 2


def test_oracle_has_a_twin_for_every_processing_entry_point(oracle_lib):
    for name in list(_abi.PROCESSING) + list(_abi.HELPERS):
        assert hasattr(oracle_lib.lib, "idsp_ref_" + name)


def test_host_helpers_agree_with_oracle_without_gpu(oracle_lib):
    """Coefficient ingestion and the hbf tables are host code in the product
    library; they must match the oracle's independent restatement."""
    import ctypes as C
    import math
    import random

    from idsp_amd._lib import load

    fn, _ = load()
    rnd = random.Random(1)
    for _ in range(2000):
        sos = [rnd.uniform(-3, 3) for _ in range(6)]
        if rnd.random() < 0.2:
            sos[rnd.randrange(6)] *= 1e6  # saturating
        if rnd.random() < 0.05:
            sos[rnd.randrange(6)] = math.nan
        frac = rnd.randrange(32)
        a, b = _abi.BiquadI32(), _abi.BiquadI32()
        assert fn["biquad_i32_from_sos"]((C.c_double * 6)(*sos), frac, C.byref(a)) == 0
        assert oracle_lib.fn["biquad_i32_from_sos"]((C.c_double * 6)(*sos), frac, C.byref(b)) == 0
        assert list(a.ba) == list(b.ba) and a.frac == b.frac == frac
        fa, fb = _abi.BiquadF32(), _abi.BiquadF32()
        fn["biquad_f32_from_sos_f64"]((C.c_double * 6)(*sos), C.byref(fa))
        oracle_lib.fn["biquad_f32_from_sos_f64"]((C.c_double * 6)(*sos), C.byref(fb))
        assert bytes(fa) == bytes(fb)
        fn["biquad_f32_from_sos"]((C.c_float * 6)(*sos), C.byref(fa))
        oracle_lib.fn["biquad_f32_from_sos"]((C.c_float * 6)(*sos), C.byref(fb))
        assert bytes(fa) == bytes(fb)
    assert fn["biquad_i32_from_sos"]((C.c_double * 6)(1, 0, 0, 1, 0, 0), 32, C.byref(_abi.BiquadI32())) == _abi.IDSP_EINVAL
    for kind in ("dec", "int"):
        for tap_set in (0, 1):
            for stages in range(1, 6):
                a, b = _abi.HbfCascadeF32(), _abi.HbfCascadeF32()
                assert fn[f"hbf_{kind}_cascade"](tap_set, stages, C.byref(a)) == 0
                assert oracle_lib.fn[f"hbf_{kind}_cascade"](tap_set, stages, C.byref(b)) == 0
                assert bytes(a) == bytes(b)
                for h in ("response_length", "state_words"):
                    assert fn[f"hbf_{kind}_{h}"](C.byref(a)) == oracle_lib.fn[f"hbf_{kind}_{h}"](C.byref(b))
        assert fn[f"hbf_{kind}_cascade"](0, 6, C.byref(_abi.HbfCascadeF32())) == _abi.IDSP_EINVAL
    cfg = _abi.LockinI32()
    cfg.order, cfg.cascade = 2, 2
    assert fn["lockin_state_words"](C.byref(cfg)) == oracle_lib.fn["lockin_state_words"](C.byref(cfg)) == 18
