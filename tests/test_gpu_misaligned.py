"""f32 / i32 slices are 4-byte aligned in the reference, and a host may hand any such sub-slice to the boundary.  Replay
the parity suites with every device buffer (input, output, state, coefficient planes) starting one element -- 4 or 8
bytes -- into its allocation (IDSP_TEST_MISALIGN, read by tests/_backends.py): each kernel must either take a path
without 16-byte vectors / LDS-DMA or be legal at that alignment, and give the same bits.  Modes "lds" / "lds-persistent"
replay the same suites with the LDS-DMA FrameMajor kernel forced for every processor (and with a persistent grid).  The
suites' own in-place pass (run_both) covers y == x there."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = ["tests/test_gpu_parity.py", "tests/test_gpu_bylane.py", "tests/test_gpu_cic.py", "tests/test_gpu_normal_wdf.py",
          "tests/test_fm_disc.py"]


@pytest.mark.parametrize("mode", ["misaligned", "inplace", "both", "lds", "lds-persistent"])
def test_parity_suites_on_buffers_without_16_byte_alignment(gpu, mode):
    """mode "inplace": every stream / by-lane call runs as `Inplace::inplace` (y is x, dsp-process/src/process.rs:61-65)."""
    env = dict(os.environ)
    if mode in ("misaligned", "both"):
        env["IDSP_TEST_MISALIGN"] = "1"
    if mode in ("inplace", "both"):
        env["IDSP_TEST_INPLACE"] = "1"
    if mode.startswith("lds"):
        # every FrameMajor case with whole 256-lane blocks (SHAPES holds ragged-frame ones) on the LDS-DMA kernel whatever
        # the processor's cost or the launch size: clamp / f32 DF1 / ByLane / Normal functors meet the oracle there;
        # "lds-persistent": a grid of 3 workgroups walks the 256-lane blocks (uneven shares, several blocks per workgroup)
        env.update(IDSP_DIAG="1", IDSP_LDS_MIN_WAVES="0", IDSP_LDS_COST="100000", IDSP_LDS_GRID="3" if mode == "lds-persistent" else "0",
                   IDSP_NO_FM_STAGED="1")
    r = subprocess.run([sys.executable, "-m", "pytest", *SUITES, "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
