#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_reference_kat.py tests/test_gpu_misaligned.py tests/test_gpu_full_tensor_oracle.py -m gpu -x -q -k "hbf or c3 or c4 or c5 or every_output" > $O/r03_hbf_tests.log 2>&1; echo "rc=$?" >> $O/r03_hbf_tests.log
echo "# pair mode (default)" > $O/r03_perf_c3_a.jsonl
python tools/perf_configs.py --only c3,hbfvar 2>&1 | grep -v libdrm >> $O/r03_perf_c3_a.jsonl
echo "# IDSP_HBF_NO_PAIR=1" >> $O/r03_perf_c3_a.jsonl
IDSP_DIAG=1 IDSP_HBF_NO_PAIR=1 python tools/perf_configs.py --only c3 2>&1 | grep -v libdrm >> $O/r03_perf_c3_a.jsonl
echo "# IDSP_HBF_FM_PAIR=1" >> $O/r03_perf_c3_a.jsonl
IDSP_DIAG=1 IDSP_HBF_FM_PAIR=1 python tools/perf_configs.py --only c3 2>&1 | grep -v libdrm >> $O/r03_perf_c3_a.jsonl
tail -5 $O/r03_hbf_tests.log; cut -c1-160 $O/r03_perf_c3_a.jsonl
