#!/bin/bash
# wave roles by SIMD in the lock-in: arms on SIMDs 0/2 | 1/3 (rot1), the same with the mixer in the read-out waves (rot1m) or without
# the alternating priority (rot1p0), arms on SIMDs 0/1 | 2/3 (rot2), roles by wave index (rot0)
mkdir -p gpurun_out/q
for v in rot1 rot1m rot1p0 rot2 rot0 rot1 rot1m rot1p0 rot2 rot0; do echo "== $v"; build/exp_lockin_$v | grep "^C4\|^one"; done 2>&1 | tee gpurun_out/q/exp_lockin_rot_b.txt
