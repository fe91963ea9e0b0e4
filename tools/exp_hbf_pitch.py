"""C3 experiment: does the LANE_MAJOR lane pitch (frames * 16 * 4 bytes) matter for the /16 decimator?"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import perf_configs as P
for fr in (4096, 4160, 4224, 5120, 3968):
    P.hbf("dec", 4, 16384, fr, 1, 10, "pitch")
