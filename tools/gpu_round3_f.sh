#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests/test_gpu_lockin_fuzz.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_misaligned.py -m gpu -x -q -k "lockin or lowpass or c4" > $O/r03_lockin_tests_b.log 2>&1; echo "rc=$?" >> $O/r03_lockin_tests_b.log
python tools/perf_configs.py --only c4 2>&1 | grep -v libdrm > $O/r03_perf_c4_c.jsonl
tail -3 $O/r03_lockin_tests_b.log; cut -c1-170 $O/r03_perf_c4_c.jsonl | head -12
