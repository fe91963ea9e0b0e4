"""Debug helper (GPU): element-wise comparison of hbf_dec_f32 against the oracle on a few shapes, with the positions of the
mismatches.  usage: python tools/dbg_hbf.py [fm|lm]"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, '.')
from tests import _harness as H
from tests._backends import OracleBackend, GpuBackend
from idsp_amd import _abi
ob, gb = OracleBackend(), GpuBackend()
def cascade(ts, s):
    cfg = _abi.HbfCascadeF32(); assert H.oracle().fn["hbf_dec_cascade"](ts, s, C.byref(cfg)) == 0; return cfg
which = sys.argv[1] if len(sys.argv) > 1 else "all"
cases = []
if which in ("all", "lm"):
    cases += [(0,1,1,2048,1),(0,1,1,512,1),(0,2,1,1024,1),(0,3,1,1024,1),(0,4,1,256,1),(0,4,1,64,1),(0,5,1,128,1),(1,4,3,300,1)]
if which in ("all", "fm"):
    cases += [(0,4,16,64,0),(0,4,16,64,0),(0,4,16,256,0),(0,4,32,320,0),(1,4,16,100,0)]
for (ts, S, lanes, frames, layout) in cases:
    cfg = cascade(ts, S); R = 1 << S
    words = H.oracle().fn["hbf_dec_state_words"](C.byref(cfg))
    rng = np.random.default_rng(1)
    init = rng.standard_normal(size=(words, lanes)).astype(np.float32).view(np.uint32)
    so, sg = init.copy(), init.copy()
    x = rng.standard_normal(lanes*frames*R).astype(np.float32)
    rco, yo = ob.cfgcall("hbf_dec_f32", cfg, so, x, (lanes*frames,), np.float32, lanes, frames, layout)
    rcg, yg = gb.cfgcall("hbf_dec_f32", cfg, sg, x, (lanes*frames,), np.float32, lanes, frames, layout)
    bad = np.nonzero(yo.view(np.uint32) != yg.view(np.uint32))[0]
    sbad = np.argwhere(so != sg)
    if layout == 0:
        fr, ln = bad // lanes, bad % lanes
    else:
        ln, fr = bad // frames, bad % frames
    print((ts,S,lanes,frames,layout), H.engine().fn["last_kernel"]().decode()[:24], "bad y:", bad.size, "frames", sorted(set(fr.tolist()))[:12], "lanes", sorted(set(ln.tolist()))[:16],
          "bad state rows:", sorted(set(sbad[:,0].tolist()))[:8], "lanes", sorted(set(sbad[:,1].tolist()))[:16])
