"""Every form of the multi-wave lock-in kernel (idsp_amd/csrc/lockin_waves.h), not only the ones default dispatch picks:
4 or 6 waves per 64 lanes x 8- or 16-frame batches x LDS-DMA or register-prefetch input.  Default dispatch takes 6 waves /
16 frames up to 16384 FrameMajor lanes, 4 waves / 16 frames for `Complex<i32>` and `norm_sqr` up to 40960 lanes, 6 waves / 8 frames for `arg`, 4 waves / 8 frames above —
6 waves with 16-frame batches is reachable through the diagnostic switches only, and so is the register-prefetch input on
aligned buffers.  Each combination re-runs the randomised lock-in suite and the lock-in parity tests in a process of its
own (the switches are read once per process) against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# IDSP_LOCKIN_NO_STAGES: the stage-wave kernel (tests/test_gpu_lockin_stages.py) would otherwise take the K = 2 shapes of these suites
_NS = dict(IDSP_LOCKIN_NO_STAGES="1")
FORMS = [dict(_NS, IDSP_LOCKIN_WAVES="6", IDSP_LOCKIN_B="16"), dict(_NS, IDSP_LOCKIN_WAVES="6", IDSP_LOCKIN_B="8"), dict(_NS, IDSP_LOCKIN_WAVES="4", IDSP_LOCKIN_B="8"),
         dict(_NS, IDSP_LOCKIN_WAVES="4", IDSP_LOCKIN_B="16"), dict(_NS, IDSP_LOCKIN_NO_DMA="1"), dict(_NS, IDSP_LOCKIN_NO_DMA="1", IDSP_LOCKIN_WAVES="6")]


@pytest.mark.parametrize("form", FORMS, ids=lambda f: ",".join(f"{k[12:]}={v}" for k, v in f.items() if k != "IDSP_LOCKIN_NO_STAGES"))
def test_lockin_suites_on_a_forced_form(gpu, form):
    if os.environ.get("IDSP_LOCKIN_WAVES") or os.environ.get("IDSP_LOCKIN_NO_DMA") or os.environ.get("IDSP_LOCKIN_NO_STAGES"):
        pytest.skip("already inside a forced run")
    env = dict(os.environ, IDSP_DIAG="1", **form)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_lockin_fuzz.py", "tests/test_gpu_parity.py", "-m", "gpu", "-x", "-q", "-k", "lockin"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
