#!/bin/bash
# round-2 experiment matrix 7: ring depth at large lane counts (256 persistent workgroups, column panels)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/exp_c5_matrix7.jsonl; mkdir -p gpurun_out; : > $O
E=build/exp_c5
run() { timeout 120 $E "$@" >> $O; }
for L in 1048576 524288 262144 131072; do for NB in 6 7 8 9 10 11 12 14; do run $L 4096 0 256 $NB 0 0 1; done; done
for NB in 6 8 9 10 12; do run 131072 4096 0 0 $NB 0 0 2; done
cat $O
