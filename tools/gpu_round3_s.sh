#!/bin/bash
# lock-in forms for [Lowpass<N>; K], K != 2, at 32768 lanes: default against 6 waves x 16 frames (66 KiB of LDS with the 16-byte table: do two share a CU now?)
mkdir -p gpurun_out/s
for f in "default" "IDSP_LOCKIN_WAVES=6 IDSP_LOCKIN_B=16" "IDSP_LOCKIN_WAVES=4 IDSP_LOCKIN_B=16"; do
  if [ "$f" = default ]; then env python tools/perf_configs.py --only c4small 2>&1 | grep "32768" | sed "s/^/[$f] /"; else env IDSP_DIAG=1 $f python tools/perf_configs.py --only c4small 2>&1 | grep "32768" | sed "s/^/[$f] /"; fi
done | tee gpurun_out/s/perf_c4small_32768.jsonl
