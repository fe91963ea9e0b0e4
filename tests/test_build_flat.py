"""Build check (no GPU): no kernel of the built library reaches memory through `flat_*` instructions, except the ones
tools/check_flat.py lists with a reason.  A pointer rebuilt from a wave-uniform integer base is a generic one, and its accesses
then count on lgkmcnt beside vmcnt (gfx9: a 4-bit counter shared with the wave's LDS traffic); round 6 found both staged stream
kernels and the lock-in kernels' row stores on flat accesses."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_unexpected_flat_accesses():
    import check_flat

    lib = os.path.join(ROOT, "idsp_amd", "lib", "libidsp_hip.so")
    assert os.path.exists(lib), "build the HIP engine first (make lib)"
    users = check_flat.flat_users(lib)
    names = sorted(users)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines() if names else []
    bad = [(d[:120], dict(users[n])) for n, d in zip(names, dem) if not any(re.search(rx, d) for rx, _ in check_flat.ALLOWED)]
    assert not bad, f"kernels with flat accesses: {bad[:5]}"
