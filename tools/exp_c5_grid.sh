#!/bin/bash
# C5 experiment (round 2): how should the LDS-DMA FrameMajor kernel be launched beyond 65536 lanes?
# Sweeps the persistent-grid cap (IDSP_DIAG=1 IDSP_LDS_GRID) over lane counts 2^16..2^20; "old" = round-1
# dispatch (register-window kernel above 2048 waves).  Output: gpurun_out/exp_c5_grid.jsonl
set -u
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/exp_c5_grid.jsonl
mkdir -p gpurun_out; : > $O
for G in 0 256 512 768 1024; do
  echo "{\"grid_cap\": $G}" >> $O
  IDSP_DIAG=1 IDSP_LDS_GRID=$G timeout 600 python tools/perf_configs.py --only c5sweep --iters 12 >> $O 2>gpurun_out/exp_c5_grid.err
done
echo '{"grid_cap": "old: LDS kernel only up to 2048 waves"}' >> $O
IDSP_DIAG=1 IDSP_LDS_MAX_WAVES=2048 timeout 600 python tools/perf_configs.py --only c5sweep --iters 12 >> $O 2>>gpurun_out/exp_c5_grid.err
cat $O
