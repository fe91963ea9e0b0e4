"""Build check (no GPU): no kernel of the built library uses scratch memory except the ones tools/check_scratch.py
lists with a reason.  Several kernels keep per-thread arrays in registers only as long as every loop over them is fully
unrolled; a failed unroll moves them to scratch silently (the Makefile passes -Wno-pass-failed), and this is the signal."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_unexpected_scratch_users():
    import check_scratch

    lib = os.path.join(ROOT, "idsp_amd", "lib", "libidsp_hip.so")
    assert os.path.exists(lib), "build the HIP engine first (make lib)"
    ks = check_scratch.kernels_of(lib)
    assert len(ks) > 500, "code objects not found in the library"
    names = check_scratch.demangle([k[0] for k in ks])
    import re

    bad = [(n, k[1]) for k, n in zip(ks, names)
           if k[1] and not any(re.search(p, n) or re.search(p, k[0]) for p, _ in check_scratch.ALLOWED)]
    assert not bad, f"kernels with unexpected scratch use: {bad[:5]}"
    # the hot kernels of the five BASELINE configurations, by name: present and spill-free
    for frag in ("stream_frame_major_lds", "stream_lane_major_staged", "hbf_dec_wave", "lockin_waves_kernel"):
        hot = [k for k, n in zip(ks, names) if (frag in n or frag in k[0]) and not any(re.search(p, n) or re.search(p, k[0]) for p, _ in check_scratch.ALLOWED)]
        assert hot and all(k[1] == 0 for k in hot), frag
