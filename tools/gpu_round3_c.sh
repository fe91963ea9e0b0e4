#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests/test_gpu_lockin_fuzz.py tests/test_gpu_parity.py -m gpu -x -q -k "lockin or lowpass" > $O/r03_lockin_tests.log 2>&1; echo "rc=$?" >> $O/r03_lockin_tests.log
python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c4" >> $O/r03_lockin_tests.log 2>&1; echo "rc=$?" >> $O/r03_lockin_tests.log
python tools/perf_configs.py --only c4 > $O/r03_perf_c4_a.jsonl 2>&1
tail -4 $O/r03_lockin_tests.log; cut -c1-170 $O/r03_perf_c4_a.jsonl
