#!/bin/bash
# 4-wave lock-in kernel at C4: alternating priority 3 (base) / arm waves 2, read-out 0 (parm)
mkdir -p gpurun_out/s
for r in 1 2; do for v in base parm; do echo "== $v"; timeout 200 build/exp_ls_$v a b c | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['N'], d['mode'], d['lanes'], d['y_mismatches'], d['ms_waves'], d['frac_waves'])"; done; done 2>&1 | tee gpurun_out/s/exp_lockin_prio_v2b.txt
