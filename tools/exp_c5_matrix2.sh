#!/bin/bash
# round-2 experiment matrix 2 for tools/exp_c5.hip: lanes-per-thread variants; output gpurun_out/exp_c5_matrix2.jsonl
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/exp_c5_matrix2.jsonl; mkdir -p gpurun_out; : > $O
E=build/exp_c5
run() { timeout 120 $E "$@" >> $O; }
#   lanes frames pitch grid NB inplace yoff LPT
for NB in 7 5; do
run 131072 4096 0 0 $NB 0 0 2
run 262144 4096 0 0 $NB 0 0 4
run 262144 4096 0 0 $NB 0 0 2
run 524288 4096 0 0 $NB 0 0 8
run 524288 4096 0 0 $NB 0 0 4
run 1048576 4096 0 0 $NB 0 0 16
run 1048576 4096 0 0 $NB 0 0 8
run 1048576 4096 0 256 $NB 0 0 8
run 1048576 4096 0 256 $NB 0 0 4
run 1048576 4096 0 256 $NB 0 0 2
done
# placement probes at 131072 lanes, full rows, 512 workgroups
run 131072 4096 0 512 7 1 0 1
for Y in 1024 16384 49152 1048576 3145728 7340032; do run 131072 4096 0 512 7 0 $Y 1; done
# LPT 2 in place and at offsets
run 131072 4096 0 0 7 1 0 2
for Y in 1024 49152 3145728 7340032; do run 131072 4096 0 0 7 0 $Y 2; done
# non-multiples: 196608 lanes with LPT 2 (384 workgroups) and 98304 with LPT 1 (384)
run 196608 4096 0 0 7 0 0 2
run 98304 4096 0 0 7 0 0 1
cat $O
