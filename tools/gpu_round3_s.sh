#!/bin/bash
# DDS FrameMajor with the full-circle-table cossin: parity of the DDS / cossin suites, then the dds lines of the survey with and without it
mkdir -p gpurun_out/s
python -m pytest tests -m gpu -x -q -k "dds or cossin or accu or kat or last_kernel" > gpurun_out/s/pytest_dds.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s/pytest_dds.log
tail -3 gpurun_out/s/pytest_dds.log
for v in 0 1; do
  if [ $v = 1 ]; then export IDSP_DIAG=1 IDSP_DDS_NO_CIRCLE=1; fi
  python tools/perf_configs.py --only c4 2>&1 | grep '"dds' | sed "s/^/no_circle=$v /"
done | tee gpurun_out/s/perf_dds_circle.jsonl
