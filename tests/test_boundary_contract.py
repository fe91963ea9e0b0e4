"""Boundary rules that can be checked without a GPU: the product never reaches into the
oracle, and contract violations come back as IDSP_EINVAL before anything touches the device
(the reference's debug_assert!/const-assert preconditions)."""
import ctypes as C
import os
import re

from idsp_amd import _abi
from idsp_amd._lib import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sources(*dirs, exts=(".py", ".h", ".hpp", ".hip", ".c", ".cpp")):
    for d in dirs:
        for base, _, files in os.walk(os.path.join(ROOT, d)):
            if "__pycache__" in base:
                continue
            for f in files:
                if f.endswith(exts):
                    yield os.path.join(base, f)


def test_product_never_imports_or_links_the_oracle():
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b|idsp_ref_|idsp_oracle|libidsp_oracle)", re.M)
    for path in list(_sources("idsp_amd", "include")):
        text = open(path, encoding="utf-8", errors="replace").read()
        assert not pat.search(text), f"{path} refers to the oracle"
    # bench.py may use it in the cpu_baseline leg only
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"\boracle\b", bench)]
    body = bench[bench.index("def cpu_baseline"):bench.index("def main")]
    assert bench.count("import oracle") == 1 and "import oracle" in body
    assert uses, "cpu_baseline leg missing"


def test_no_cpu_fallback_in_the_python_host_layer():
    text = open(os.path.join(ROOT, "idsp_amd", "process.py")).read()
    assert "no CPU path" in text and ".cpu()" not in text and "numpy" not in text


def test_contract_violations_return_einval_without_a_device():
    fn, _ = load()
    q = (_abi.BiquadI32 * 1)()
    q[0].frac = 30
    one = C.c_void_p(16)  # never dereferenced: validation fails first
    # bad layout
    assert fn["biquad_i32_df1"](C.cast(q, C.c_void_p), 1, one, one, one, 4, 4, 9, None) == _abi.IDSP_EINVAL
    assert b"layout" in fn["last_error"]()
    # F outside 0..31: `const { assert!(F >= 0 && F < 32) }` (src/iir/biquad.rs:448-450)
    q[0].frac = 32
    assert fn["biquad_i32_df1"](C.cast(q, C.c_void_p), 1, one, one, one, 4, 4, 0, None) == _abi.IDSP_EINVAL
    assert b"frac" in fn["last_error"]()
    q[0].frac = 30
    # NULL buffers, too many sections, cascade longer than 8
    assert fn["biquad_i32_df1"](C.cast(q, C.c_void_p), 1, None, one, one, 4, 4, 0, None) == _abi.IDSP_EINVAL
    assert fn["biquad_i32_df1"](C.cast(q, C.c_void_p), 65, one, one, one, 4, 4, 0, None) == _abi.IDSP_EINVAL
    q9 = (_abi.BiquadI32 * 9)()
    assert fn["cascade_i32_df1"](C.cast(q9, C.c_void_p), 9, one, one, one, 4, 4, 0, None) == _abi.IDSP_EINVAL
    # half-band: stage count, tap count, NULL buffer (4-byte aligned buffers are accepted: tests/test_gpu_misaligned.py)
    h = _abi.HbfCascadeF32()
    h.stages = 6
    assert fn["hbf_dec_f32"](C.byref(h), one, one, one, 1, 1, 1, None) == _abi.IDSP_EINVAL
    h.stages, h.m[0] = 1, 33
    assert fn["hbf_int_f32"](C.byref(h), one, one, one, 1, 1, 1, None) == _abi.IDSP_EINVAL
    h.m[0] = 3
    assert fn["hbf_dec_f32"](C.byref(h), one, None, one, 1, 1, 1, None) == _abi.IDSP_EINVAL
    assert fn["hbf_dec_state_words"](C.byref(h)) == 7 and fn["hbf_int_state_words"](C.byref(h)) == 5
    # lock-in: Lowpass order other than 1/2 is `unimplemented!()` (src/lowpass.rs:75)
    lk = _abi.LockinI32()
    lk.order, lk.cascade = 3, 1
    assert fn["lockin_i32_process"](C.byref(lk), one, one, one, 1, 1, 0, None) == _abi.IDSP_EINVAL
    assert fn["lockin_state_words"](C.byref(lk)) == 0
    lk.order, lk.cascade = 2, 5
    assert fn["lowpass_i32"](C.byref(lk), one, one, one, 1, 1, 0, None) == _abi.IDSP_EINVAL
    # FIR kind / tap count
    f = _abi.FirSymF32()
    f.kind, f.m = 4, 3
    assert fn["fir_sym_f32_process"](C.byref(f), one, one, one, 1, 1, 0, None) == _abi.IDSP_EINVAL
    f.kind, f.m = 0, 0
    assert fn["fir_sym_state_words"](C.byref(f)) == 0
    # zero lanes / zero frames are valid no-ops that never launch
    assert fn["biquad_i32_df1"](C.cast(q, C.c_void_p), 1, one, one, one, 0, 4, 0, None) == 0
    assert fn["cossin_i32"](None, None, 0, None) == 0
    assert fn["atan2_i32"](None, None, 0, None) == 0
    assert fn["atan2_i32"](None, one, 1, None) == _abi.IDSP_EINVAL
