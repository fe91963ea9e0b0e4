#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r03_gpu_tests_b.log 2>&1; echo "suite rc=$?" >> $O/r03_gpu_tests_b.log
python tools/perf_configs.py --only c4 > $O/r03_perf_c4_b.jsonl 2>&1
bash tools/collect_profiles_r03.sh r03 c4 > $O/r03_collect_c4.log 2>&1
tail -3 $O/r03_gpu_tests_b.log; cut -c1-170 $O/r03_perf_c4_b.jsonl; tail -12 $O/r03_collect_c4.log
