#!/bin/bash
# round-2 experiment matrix 4: is the large-pitch penalty a power-of-two aliasing effect?
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/exp_c5_matrix4.jsonl; mkdir -p gpurun_out; : > $O
E=build/exp_c5
run() { timeout 120 $E "$@" >> $O; }
#   lanes frames pitch grid NB inplace yoff LPT wide
for P in 131072 132096 135168 147456 262144 263168 266240 294912 1048576 1049600 1052672 1064960 1179648; do run 65536 4096 $P 0 7 0 0 1 1; done
# whole 2^20-lane job at padded pitches: column panels (grid 256, LPT 1), LPT 16 wide, LPT 4 grid 256
for P in 1048576 1052672 1064960; do
run 1048576 4096 $P 256 7 0 0 1 1
run 1048576 4096 $P 0 7 0 0 16 1
run 1048576 4096 $P 256 7 0 0 4 1
run 1048576 4096 $P 256 7 0 0 2 0
done
# 131072-lane job (8-GPU shard) at padded pitches
for P in 131072 132096 135168; do run 131072 4096 $P 0 7 0 0 2 0; run 131072 4096 $P 256 7 0 0 1 1; done
# LPT 16 with shallower rings
for NB in 5; do run 1048576 4096 0 0 $NB 0 0 16 1; done
cat $O
