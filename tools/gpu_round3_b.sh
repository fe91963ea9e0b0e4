#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out
build/ubench_cossin > $O/r03_ubench_cossin.txt 2>&1
python tools/cpu_scaling_probe.py > $O/r03_cpu_scaling_probe.txt 2>&1
# lane counts between one and two workgroups per CU: the staged single-wave kernel (64 lanes per wave) against the default
IDSP_DIAG=1 IDSP_FM_LANES_PER_WAVE=64 python tools/perf_configs.py --only lanesweep > $O/r03_lanesweep_staged64.jsonl 2>&1
cat $O/r03_ubench_cossin.txt; cat $O/r03_cpu_scaling_probe.txt; cut -c1-140 $O/r03_lanesweep_staged64.jsonl
