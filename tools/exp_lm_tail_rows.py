import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import perf_configs as P
for fr in (4, 8, 16, 28, 4096, 4100, 4104, 4112, 4124):
    P.fm_disc(65536, fr, 1, 10, "fm")
for fr in (4, 12, 4096, 4100, 4108):
    P.lockin(2, 2, 32768, fr, 1, 10, "li")
