#!/bin/bash
# round-2 experiment matrix 6: which blocks does a persistent workgroup own (panel-major vs adjacent)?
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/exp_c5_matrix6.jsonl; mkdir -p gpurun_out; : > $O
E=build/exp_c5
run() { timeout 120 $E "$@" >> $O; }
#   lanes frames pitch grid NB inplace yoff LPT adj
for A in 0 1; do
run 1048576 4096 0 256 7 0 0 1 $A
run 1048576 4096 0 512 7 0 0 1 $A
run 1048576 4096 0 256 5 0 0 1 $A
run 524288 4096 0 256 7 0 0 1 $A
run 262144 4096 0 256 7 0 0 1 $A
run 131072 4096 0 256 7 0 0 1 $A
run 1048576 4096 0 256 7 0 0 2 $A
run 1048576 4096 0 256 7 0 0 4 $A
done
cat $O
