#!/bin/bash
# LaneMajor lock-in with whole-line output stores: parity of the lock-in suites, the C4 survey, and the HW_ID dump of the 4-wave kernel
# with roles by wave index (where the waves of a workgroup land)
mkdir -p gpurun_out/s gpurun_out/q
python -m pytest tests -m gpu -x -q -k "lockin or c4 or full_tensor" > gpurun_out/s/pytest_lm.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s/pytest_lm.log
tail -3 gpurun_out/s/pytest_lm.log
python tools/perf_configs.py --only c4 2>&1 | grep "C4" | tee gpurun_out/s/perf_c4_lm_lines.jsonl
LW_DUMP=1 build/exp_lockin_trace_rot0 > gpurun_out/q/exp_lockin_trace_rot0.txt 2>&1; grep -c "wg " gpurun_out/q/exp_lockin_trace_rot0.txt
