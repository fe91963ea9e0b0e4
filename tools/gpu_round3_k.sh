#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests/test_gpu_round_split.py tests/test_gpu_ragged_lds_block.py tests/test_gpu_pitch.py tests/test_gpu_dispatch_fuzz.py tests/test_gpu_bylane.py tests/test_gpu_multi.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/r03_tests_split.log 2>&1; echo "rc=$?" >> $O/r03_tests_split.log
python tools/perf_configs.py --only ragged,lanesweep 2>&1 | grep -v libdrm > $O/r03_perf_split.jsonl
tail -12 $O/r03_tests_split.log; cut -c1-150 $O/r03_perf_split.jsonl
