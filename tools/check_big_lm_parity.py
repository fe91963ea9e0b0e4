"""Ad hoc parity of the round-4 LaneMajor kernels (lock-in line groups + stagger, fm_disc role kernel) against the oracle at shapes the
test suite is too small for: > 2000 workgroups, two rounds of staggered workgroups, long rows.  `gpurun -- python tools/check_big_lm_parity.py`"""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _harness as H
from tests._backends import OracleBackend, GpuBackend
import tests.test_fm_disc as TF
o, e = H.oracle(), H.engine()
rng = np.random.default_rng(99)
# lock-in LaneMajor: many workgroups, few frames; and 2 rounds of staggered workgroups
for lanes, frames in ((131072 + 17, 48), (70000, 2048), (100, 8192)):
    cfg = H.lockin_cfg([[1 << 22, -(1 << 27)], [1 << 21, -(1 << 26)]])
    st = rng.integers(0, 1 << 32, size=(18, lanes), dtype=np.uint64).astype(np.uint32)
    x = rng.integers(-(1 << 31), (1 << 31) - 1, size=lanes * frames, dtype=np.int64).astype(np.int32)
    so = st.copy(); sg = torch.from_numpy(st.view(np.int32).copy()).cuda()
    yo = np.empty(lanes * frames * 2, np.int32); yg = torch.full((lanes * frames * 2,), -77, dtype=torch.int32, device="cuda")
    assert o.cfgcall("lockin_i32_process", cfg, so, x, yo, lanes, frames, H.LM) == 0
    assert e.cfgcall("lockin_i32_process", cfg, sg, torch.from_numpy(x).cuda(), yg, lanes, frames, H.LM) == 0
    torch.cuda.synchronize()
    print("lockin LM", lanes, frames, e.last_kernel()[:45], np.array_equal(yg.cpu().numpy(), yo), np.array_equal(sg.cpu().numpy().view(np.uint32), so))
ob, gb = OracleBackend(), GpuBackend()
for lanes, frames in ((200000, 64), (70000, 2048), (33, 4096)):
    cfg = TF._cfg(rng)
    init = np.zeros((7, lanes), np.uint32); init[0, ::3] = 1
    init[1:] = rng.integers(0, 1 << 32, size=(6, lanes), dtype=np.uint64).astype(np.uint32)
    so, sg = init.copy(), init.copy()
    x = TF._x(rng, lanes * frames)
    rco, yo = ob.cfgcall("fm_disc_i32", cfg, so, x, (lanes * frames,), np.int32, lanes, frames, H.LM)
    rcg, yg = gb.cfgcall("fm_disc_i32", cfg, sg, x, (lanes * frames,), np.int32, lanes, frames, H.LM)
    print("fm_disc LM", lanes, frames, H.engine().last_kernel()[:30], rco, rcg, np.array_equal(yo, yg), np.array_equal(so, sg))
