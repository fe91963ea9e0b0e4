#!/bin/bash
# lock-in at one workgroup per CU or fewer: default dispatch against the 6-wave form with 16- and 8-frame batches
mkdir -p gpurun_out/s
for f in "default" "IDSP_LOCKIN_WAVES=6 IDSP_LOCKIN_B=16 IDSP_LOCKIN_NO_STAGES=1" "IDSP_LOCKIN_WAVES=6 IDSP_LOCKIN_B=8 IDSP_LOCKIN_NO_STAGES=1" "IDSP_LOCKIN_WAVES=4 IDSP_LOCKIN_NO_STAGES=1"; do
  if [ "$f" = default ]; then env python tools/perf_configs.py --only c4small 2>&1 | grep C4s | sed "s/^/[$f] /"; else env IDSP_DIAG=1 $f python tools/perf_configs.py --only c4small 2>&1 | grep C4s | sed "s/^/[$f] /"; fi
done | tee gpurun_out/s/perf_c4small.jsonl | cut -c1-200
