#!/bin/bash
# Wdf chains on the two-wave kernel: parity + timing with and without
mkdir -p gpurun_out/l
python -m pytest tests/test_gpu_duo.py tests/test_gpu_normal_wdf.py -m gpu -x -q > gpurun_out/l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/l/pytest.log
tail -5 gpurun_out/l/pytest.log
python tools/perf_configs.py --only nw > gpurun_out/l/perf_duo.log 2>&1
IDSP_DIAG=1 IDSP_NO_DUO=1 python tools/perf_configs.py --only nw > gpurun_out/l/perf_noduo.log 2>&1
grep wdf gpurun_out/l/perf_duo.log gpurun_out/l/perf_noduo.log
