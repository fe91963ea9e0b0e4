#!/bin/bash
# table cossin + roles by SIMD + mixer on the read-out waves in the multi-wave lock-in, stage-wave kernel for small lane counts and the
# arg read-out: parity of everything that touches the lock-in (every forced form), then the C4 survey
mkdir -p gpurun_out/s
python -m pytest tests/test_gpu_lockin_stages.py tests/test_gpu_lockin_forms_forced.py tests/test_gpu_lds_path_forced.py tests/test_gpu_last_kernel.py -m gpu -x -q > gpurun_out/s/pytest_forms.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s/pytest_forms.log
tail -4 gpurun_out/s/pytest_forms.log
python tools/perf_configs.py --only c4 2>&1 | grep -v "^/opt" | tee gpurun_out/s/perf_c4_b.jsonl
