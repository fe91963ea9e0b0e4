"""The half-band decimator's pair mode (two lanes per wave, streams interleaved in LDS, v_pk_add_f32 / v_pk_mul_f32;
idsp_amd/csrc/hbf_wave.h) is a diagnostic form — measured no faster than one lane per wave, so not the default — but it
stays bit-exact: the half-band parity, reference-KAT and full-tensor suites are re-run with IDSP_DIAG=1 IDSP_HBF_PAIR=1
(the switch is read once per process, hence the subprocess)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hbf_suites_with_two_lanes_per_wave(gpu):
    if os.environ.get("IDSP_HBF_PAIR"):
        pytest.skip("already inside the forced run")
    env = dict(os.environ, IDSP_DIAG="1", IDSP_HBF_PAIR="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_reference_kat.py",
                        "tests/test_gpu_full_tensor_oracle.py", "tests/test_gpu_fullsize.py", "-m", "gpu", "-x", "-q", "-k",
                        "hbf or c3_"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    # ... and the forced run did take the pair kernel
    probe = ("import ctypes as C, torch; from idsp_amd import _abi; from tests import _harness as H; e = H.engine(); o = H.oracle();"
             "c = _abi.HbfCascadeF32(); o.fn['hbf_dec_cascade'](0, 4, C.byref(c));"
             "x = torch.zeros(7 * 64 * 16, device='cuda'); y = torch.zeros(7 * 64, device='cuda'); s = torch.zeros((118, 7), dtype=torch.int32, device='cuda');"
             "assert e.cfgcall('hbf_dec_f32', c, s, x, y, 7, 64, H.LM) == 0; print(e.fn['last_kernel']().decode())")
    r = subprocess.run([sys.executable, "-c", probe], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert "2 lanes per wave" in r.stdout, r.stdout + r.stderr[-2000:]
