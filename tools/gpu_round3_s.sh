#!/bin/bash
# (a) 4-wave lock-in kernel with 8- / 16-frame batches against the stage kernel at 49152 ... 196608 lanes; (b) the one-thread-per-lane DDS on
# the 16-byte table (IDSP_DDS_ONE_CIRCLE) against the 512-byte table at 65536 / 131072 lanes
mkdir -p gpurun_out/s
timeout 300 build/exp_lockin_stages a b | tee gpurun_out/s/exp_lockin_batch_v2.jsonl | cut -c1-75,100-117,161-250
for v in 0 1; do
  if [ $v = 1 ]; then export IDSP_DIAG=1 IDSP_DDS_ONE_CIRCLE=1; fi
  python tools/perf_configs.py --only c4 2>&1 | grep '"dds' | sed "s/^/one_circle=$v /"
done | tee gpurun_out/s/perf_dds_one_circle.jsonl
