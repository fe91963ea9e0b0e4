#!/bin/bash
# rows off the 16-byte grid: parity + timing against round 2's rule
mkdir -p gpurun_out/m
python -m pytest tests/test_gpu_rows_off_16_byte_grid.py tests/test_gpu_ragged_lds_block.py tests/test_gpu_frame_major_staged.py tests/test_gpu_round_split.py tests/test_gpu_misaligned.py tests/test_gpu_pitch.py tests/test_gpu_dispatch_fuzz.py -m gpu -x -q > gpurun_out/m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/m/pytest.log
tail -15 gpurun_out/m/pytest.log
python tools/perf_configs.py --only ragged > gpurun_out/m/perf_ragged.log 2>&1
IDSP_DIAG=1 IDSP_ALIGN16_ONLY=1 python tools/perf_configs.py --only ragged > gpurun_out/m/perf_ragged_align16.log 2>&1
grep -h ragged gpurun_out/m/perf_ragged.log | cut -c1-200
echo ---
grep -h ragged gpurun_out/m/perf_ragged_align16.log | cut -c1-200
