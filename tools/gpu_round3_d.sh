#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out
echo "# default" > $O/r03_exp_lockin_a.jsonl
python tools/exp_lockin_lanes.py >> $O/r03_exp_lockin_a.jsonl 2>&1
for W in 4 6; do for B in 8 16; do
echo "# IDSP_LOCKIN_WAVES=$W IDSP_LOCKIN_B=$B" >> $O/r03_exp_lockin_a.jsonl
IDSP_DIAG=1 IDSP_LOCKIN_WAVES=$W IDSP_LOCKIN_B=$B python tools/perf_configs.py --only c4 2>&1 | grep -E "C4:|C4v|arg" >> $O/r03_exp_lockin_a.jsonl
done; done
cut -c1-150 $O/r03_exp_lockin_a.jsonl
