#!/bin/bash
# stage kernel with one lane group per workgroup and an input ring of 5 / 4 / 3 slots (66 / 62 / 58 KiB of LDS): do two workgroups share a CU below 64 KiB?
mkdir -p gpurun_out/s
for v in ring5 ring4 ring3; do echo "== $v"; timeout 200 build/exp_ls_$v a b c d | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['mode'], 'G', d['G'], d['lanes'], d['y_mismatches'], d['state_mismatches'], 'waves', d['ms_waves'], 'stages', d['ms_stages'], d['frac_stages'])"; done 2>&1 | tee gpurun_out/s/exp_lockin_ring.txt
