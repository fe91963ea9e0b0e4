"""LaneMajor lock-in: input by DMA or by registers (IDSP_DIAG=1 IDSP_LOCKIN_NO_DMA=1), 4 or 6 waves per 64 lanes
(IDSP_LOCKIN_WAVES), the three outputs, 32768 / 65536 lanes.  One process per switch combination (the switches are read
once): `python tools/exp_lockin_lm.py` drives them, `python tools/exp_lockin_lm.py run` is one combination."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "run":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
    import perf_configs as P
    for lanes in (32768, 65536):
        for out in ("iq", "norm_sqr", "arg"):
            P.lockin(2, 2, lanes, 4096, 1, 20, os.environ.get("TAG", "lm"), out)
    P.lockin(1, 1, 32768, 4096, 1, 20, os.environ.get("TAG", "lm"), "iq")
else:
    for nodma in ("", "1"):
        for waves in ("", "4", "6"):
            env = dict(os.environ, IDSP_DIAG="1", TAG=f"in={'reg' if nodma else 'dma'},waves={waves or 'auto'}")
            if nodma:
                env["IDSP_LOCKIN_NO_DMA"] = "1"
            if waves:
                env["IDSP_LOCKIN_WAVES"] = waves
            r = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True)
            sys.stdout.write("".join(l + "\n" for l in r.stdout.splitlines() if l.startswith("{")))
            sys.stdout.flush()
