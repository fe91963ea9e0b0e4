#!/bin/bash
# C5 dispatch policy check through the ABI with plain torch allocations (tools/perf_configs.py --only c5sweep)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/exp_c5_policy.jsonl; mkdir -p gpurun_out; : > $O
echo '{"policy": "default"}' >> $O
timeout 600 python tools/perf_configs.py --only c5sweep --iters 12 >> $O 2>gpurun_out/exp_c5_policy.err
echo '{"policy": "default, second process"}' >> $O
timeout 600 python tools/perf_configs.py --only c5sweep --iters 12 >> $O 2>>gpurun_out/exp_c5_policy.err
echo '{"policy": "IDSP_LDS_GRID=0 (no persistence)"}' >> $O
IDSP_DIAG=1 IDSP_LDS_GRID=0 timeout 600 python tools/perf_configs.py --only c5sweep --iters 12 >> $O 2>>gpurun_out/exp_c5_policy.err
echo '{"policy": "IDSP_LDS_LPT=1 IDSP_LDS_GRID=256"}' >> $O
IDSP_DIAG=1 IDSP_LDS_LPT=1 IDSP_LDS_GRID=256 timeout 600 python tools/perf_configs.py --only c5sweep --iters 12 >> $O 2>>gpurun_out/exp_c5_policy.err
echo '{"policy": "IDSP_LDS_LPT=2 IDSP_LDS_GRID=256"}' >> $O
IDSP_DIAG=1 IDSP_LDS_LPT=2 IDSP_LDS_GRID=256 timeout 600 python tools/perf_configs.py --only c5sweep --iters 12 >> $O 2>>gpurun_out/exp_c5_policy.err
cat $O
