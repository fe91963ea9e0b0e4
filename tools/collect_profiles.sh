#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun):
#   kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
# (they do not fit one pass: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2), then SQ counters.
# Output: gpurun_out/prof_<tag>/ (scratch); summaries are produced by tools/rocpd_summary.py.
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python bench.py --no-cpu > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o bench -- python bench.py --no-cpu --steps 5 --warmup 2 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o bench -- python bench.py --no-cpu --steps 5 --warmup 2 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmc_sq -o bench -- python bench.py --no-cpu --steps 5 --warmup 2 > $O/pmc_sq.log 2>&1
python tools/rocpd_summary.py stats $O/trace/bench_results.db > $O/kernel_stats.csv
python tools/rocpd_summary.py pmc $O/pmc_fetch/bench_results.db $O/pmc_write/bench_results.db $O/pmc_sq/bench_results.db > $O/pmc.csv
grep -h "^{" $O/trace.log | tail -1 > $O/bench_under_rocprof.json
head -3 $O/kernel_stats.csv | cut -c1-200
cat $O/pmc.csv | cut -c1-200
