#!/bin/bash
# the 4-wave lock-in kernel with roles by SIMD (r1) or wave index (r0), mixer on the read-out (m1) or arm (m0) waves, at lane counts of
# two and more workgroups per CU
mkdir -p gpurun_out/r
for v in r0m0 r0m1 r1m0 r1m1; do echo "== $v"; timeout 200 build/exp_ls_$v a b | cut -c1-20,60-140,190-260; done 2>&1 | tee gpurun_out/r/exp_lockin_rot_mixr.txt
