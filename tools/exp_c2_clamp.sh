#!/bin/bash
# round-2 experiment: ring depth of the plain and the clamp processor at C2 under four placements of y (adjacent to x, own
# allocation, 48 KiB further, in place), for both addressing forms of the kernel (RUN = false / true).  Build first:
#   F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fwrapv -fno-slp-vectorize -Iinclude -Iidsp_amd/csrc"
#   hipcc $F tools/exp_c5.hip -o build/exp_c5; hipcc $F -DEXP_RUN=true tools/exp_c5.hip -o build/exp_c5_run
#   hipcc $F -DEXP_CLAMP tools/exp_c5.hip -o build/exp_c5_clamp; hipcc $F -DEXP_CLAMP -DEXP_RUN=true tools/exp_c5.hip -o build/exp_c5_clamp_run
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/exp_c2_clamp.jsonl; : > $O
for E in build/exp_c5 build/exp_c5_run build/exp_c5_clamp build/exp_c5_clamp_run; do
  echo "{\"binary\": \"$E\"}" >> $O
  for NB in 4 5 6 7 8 9 10; do
    timeout 60 $E 65536 4096 0 0 $NB 0 0 1 >> $O
    timeout 60 $E 65536 4096 0 0 $NB 0 -1 1 >> $O
    timeout 60 $E 65536 4096 0 0 $NB 0 49152 1 >> $O
    timeout 60 $E 65536 4096 0 0 $NB 1 0 1 >> $O
  done
done
cat $O
