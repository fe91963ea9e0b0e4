#!/bin/bash
# whole GPU suite + perf after the rows-off-the-16-byte-grid dispatch
mkdir -p gpurun_out/n
python -m pytest tests -m gpu -q -x > gpurun_out/n/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/n/pytest.log
tail -15 gpurun_out/n/pytest.log
python tools/exp_fm_unaligned_small.py 2>&1 | grep lanes > gpurun_out/n/exp_fm_unaligned_small.jsonl
IDSP_DIAG=1 IDSP_ALIGN16_ONLY=1 python tools/exp_fm_unaligned_small.py 2>&1 | grep lanes >> gpurun_out/n/exp_fm_unaligned_small.jsonl
python tools/perf_configs.py --only ragged 2>&1 | grep ragged > gpurun_out/n/perf_ragged.jsonl
IDSP_DIAG=1 IDSP_ALIGN16_ONLY=1 python tools/perf_configs.py --only ragged 2>&1 | grep ragged > gpurun_out/n/perf_ragged_align16.jsonl
cut -c1-120 gpurun_out/n/exp_fm_unaligned_small.jsonl | head -16
