import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import perf_configs as P
for lanes in (8192, 16384, 24576, 32768, 49152, 65536):
    P.lockin(2, 2, lanes, 4096, 0, 20, "E1")
for lanes in (16384, 32768):
    P.lockin(1, 1, lanes, 4096, 0, 20, "E1")
    P.lockin(2, 1, lanes, 4096, 0, 20, "E1")
