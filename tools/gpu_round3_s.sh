#!/bin/bash
# LaneMajor lock-in: DMA wait at even intervals when the read-out waves mix; parity, then the C4 lines
mkdir -p gpurun_out/s
python -m pytest tests -m gpu -x -q -k "lockin or c4 or full_tensor" > gpurun_out/s/pytest_lmw.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s/pytest_lmw.log
tail -3 gpurun_out/s/pytest_lmw.log
python tools/perf_configs.py --only c4 2>&1 | grep "C4" | tee gpurun_out/s/perf_c4_lmwait.jsonl
