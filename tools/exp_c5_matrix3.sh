#!/bin/bash
# round-2 experiment matrix 3 for tools/exp_c5.hip: wide tiles (whole row pieces per workgroup) vs tile sequence
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/exp_c5_matrix3.jsonl; mkdir -p gpurun_out; : > $O
E=build/exp_c5
run() { timeout 120 $E "$@" >> $O; }
#   lanes frames pitch grid NB inplace yoff LPT wide
for W in 1 0; do for NB in 7 5; do
run 131072 4096 0 0 $NB 0 0 2 $W
run 262144 4096 0 0 $NB 0 0 4 $W
run 524288 4096 0 0 $NB 0 0 8 $W
run 1048576 4096 0 0 $NB 0 0 16 $W
run 1048576 4096 0 256 $NB 0 0 8 $W
done; done
# C2 shape with LPT 2 on 128 workgroups (how much does one CU deliver?) and C2 itself
run 65536 4096 0 0 7 0 0 2 1
run 65536 4096 0 0 7 0 0 1 1
# 98304 / 196608 lanes: 384 workgroups of LPT 1 / 2 vs 192 of LPT 2 / 4... (non powers of two)
run 196608 4096 0 0 7 0 0 2 1
run 196608 4096 0 256 7 0 0 2 1
run 196608 4096 0 0 7 0 0 4 1
cat $O
