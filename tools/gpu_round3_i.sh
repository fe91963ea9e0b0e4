#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out
bash tools/collect_profiles_r03.sh r03 all > $O/r03_collect_all.log 2>&1
python bench.py --gpus 2 --steps 20 --warmup 5 > $O/r03_bench_gpus2_selfspawn.json 2> $O/r03_bench_gpus2_selfspawn.err
tail -40 $O/r03_collect_all.log | cut -c1-200; head -c 400 $O/r03_bench_gpus2_selfspawn.json
