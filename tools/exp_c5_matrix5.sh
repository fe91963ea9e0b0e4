#!/bin/bash
# round-2 experiment matrix 5: y placement relative to x at 2^20 and 2^19 lanes (repeat each twice for noise)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/exp_c5_matrix5.jsonl; mkdir -p gpurun_out; : > $O
E=build/exp_c5
run() { timeout 120 $E "$@" >> $O; }
#   lanes frames pitch grid NB inplace yoff LPT wide
for Y in 0 65536 262144 524288 1048576 2097152 3145728 6291456; do
run 1048576 4096 0 0 7 0 $Y 16 1
run 1048576 4096 0 256 7 0 $Y 1 1
done
run 1048576 4096 0 0 7 1 0 16 1
run 1048576 4096 0 256 7 1 0 1 1
for Y in 0 262144 1048576; do run 524288 4096 0 0 7 0 $Y 8 1; run 524288 4096 0 256 7 0 $Y 1 1; done
# noise check: same config 3 times
for k in 1 2 3; do run 1048576 4096 0 256 7 0 0 1 1; done
cat $O
