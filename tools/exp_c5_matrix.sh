#!/bin/bash
# round-2 experiment matrix for tools/exp_c5.hip (see its header); output gpurun_out/exp_c5_matrix.jsonl
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/exp_c5_matrix.jsonl; mkdir -p gpurun_out; : > $O
E=build/exp_c5
run() { timeout 120 $E "$@" >> $O; }
# 1. stride effect alone: 65536 lanes (grid 256) at various row pitches
for P in 65536 66560 69632 81920 98304 131072 196608 262144 1048576; do run 65536 4096 $P 0 7; done
# 2. ring depth x residency at 131072 lanes, full rows (grid 512) and persistent (grid 256)
for NB in 3 4 5 6 7 8; do run 131072 4096 0 512 $NB; run 131072 4096 0 256 $NB; done
# 3. non-power-of-two lane counts, persistent 256
for L in 196608 163840 327680; do run $L 4096 0 256 7; run $L 4096 0 0 7; run $L 4096 0 512 4; done
# 4. 2^20 lanes: ring depth x grid
for NB in 3 4 5 7; do for G in 256 512 0; do run 1048576 4096 0 $G $NB; done; done
# 5. ring depth at C2 itself (sanity vs round 1)
for NB in 4 5 6 7 8; do run 65536 4096 0 0 $NB; done
cat $O
