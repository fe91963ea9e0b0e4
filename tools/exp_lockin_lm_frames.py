import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_configs as P
for lanes in (16384, 32768, 65536):
    for frames in (1024, 2048, 4096, 16384):
        if lanes * frames <= 32768 * 16384:
            P.lockin(2, 2, lanes, frames, 1, 10, os.environ.get("TAG", "lm"))
