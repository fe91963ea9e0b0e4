#!/bin/bash
# round-3 first GPU pass: toolchain probe, new GPU tests, full GPU suite, the bench at N = 1 and N = 2 (self-spawned), cliff rows
mkdir -p gpurun_out; O=gpurun_out
{ echo "cargo: $(which cargo 2>&1)"; echo "rustc: $(which rustc 2>&1)"; echo "nproc: $(nproc)"; lscpu | grep -E "Model name|Socket|NUMA node\(s\)"; free -g | head -2; rocm-smi --showproductname 2>/dev/null | head -8; } > $O/r03_toolchain.txt 2>&1
python -m pytest tests/test_gpu_ragged_lds_block.py tests/test_gpu_multi.py tests/test_gpu_frame_major_staged.py -m gpu -x -q > $O/r03_new_tests.log 2>&1; echo "new tests rc=$?" >> $O/r03_new_tests.log
python -m pytest tests -m gpu -x -q > $O/r03_gpu_tests.log 2>&1; echo "suite rc=$?" >> $O/r03_gpu_tests.log
python bench.py --steps 20 --warmup 5 > $O/r03_bench_driverflags.json 2> $O/r03_bench_driverflags.err
python bench.py --gpus 2 --steps 20 --warmup 5 > $O/r03_bench_gpus2_selfspawn.json 2> $O/r03_bench_gpus2_selfspawn.err
python tools/perf_configs.py --only ragged > $O/r03_perf_ragged.jsonl 2>&1
python tools/perf_configs.py --only lanesweep > $O/r03_lanesweep_default.jsonl 2>&1
IDSP_DIAG=1 IDSP_LDS_GRID=0 python tools/perf_configs.py --only lanesweep > $O/r03_lanesweep_grid0.jsonl 2>&1
tail -3 $O/r03_new_tests.log $O/r03_gpu_tests.log; cat $O/r03_toolchain.txt; head -c 1500 $O/r03_bench_driverflags.json; echo; tail -5 $O/r03_bench_gpus2_selfspawn.err; head -c 600 $O/r03_bench_gpus2_selfspawn.json
