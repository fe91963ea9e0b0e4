#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out
python bench.py --steps 20 --warmup 5 > $O/r03_bench_driverflags_b.json 2> $O/r03_bench_driverflags_b.err
python bench.py --steps 20 --warmup 5 --layout lane --no-c5 --no-cpu > $O/r03_bench_lane_b.json 2>> $O/r03_bench_driverflags_b.err
python tools/perf_configs.py --only c5,ragged 2>&1 | grep -v libdrm > $O/r03_perf_c5_b.jsonl
echo "# IDSP_LOCKIN_NO_DMA=1" > $O/r03_perf_c4_lm.jsonl
IDSP_DIAG=1 IDSP_LOCKIN_NO_DMA=1 python tools/perf_configs.py --only c4 2>&1 | grep "LM" >> $O/r03_perf_c4_lm.jsonl
python -m pytest tests/test_gpu_hbf_pair_mode.py tests/test_gpu_ragged_lds_block.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/r03_tests_c.log 2>&1; echo "rc=$?" >> $O/r03_tests_c.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_bench_driverflags_b.json"))
print("C2", d["value"], d["roofline"]["frac"], d["integrity"]["match"], d["integrity"].get("oracle_live_match"))
print("C5", d["c5"]["value"], d["c5"]["ms_per_step"], d["c5"]["roofline"]["frac"], d["c5"]["integrity"]["match"])
cb=d["cpu_baseline"]; print("cpu", cb["value"], cb["cores"], cb["cgroup_cpu_quota"], cb["single_thread_value"], cb["parallel_efficiency"], cb["by_layout"])
l=json.load(open("gpurun_out/r03_bench_lane_b.json")); print("C2 lane", l["value"], l["roofline"]["frac"], l["roofline"]["kernel"][:40])
PY
cut -c1-165 $O/r03_perf_c5_b.jsonl; cat $O/r03_perf_c4_lm.jsonl | cut -c1-165; tail -3 $O/r03_tests_c.log
