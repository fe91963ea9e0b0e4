#!/bin/bash
# Round-2 addendum to collect_profiles_r02.sh: the LaneMajor kernels that changed late in the round
# (stream_lane_major_staged, lock-in LaneMajor by DMA).  Same passes, same rules (PMC passes apart from trace domains).
#   bash tools/collect_profiles_r02b.sh
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02b
rm -rf $O; mkdir -p $O
cd $R
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"
run() {  # name, command...
  local n=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/$n/trace -o p -- "$@" > $O/$n.trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/$n/fetch -o p -- "$@" > $O/$n.fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/$n/write -o p -- "$@" > $O/$n.write.log 2>&1
  rocprofv3 --pmc $SQ -d $O/$n/sq -o p -- "$@" > $O/$n.sq.log 2>&1
  rocprofv3 --pmc $SQ2 -d $O/$n/sq2 -o p -- "$@" > $O/$n.sq2.log 2>&1
  python tools/rocpd_summary.py stats $(find $O/$n/trace -name '*results.db' | head -1) > $O/${n}_kernel_stats.csv
  python tools/rocpd_summary.py pmc $(find $O/$n/fetch $O/$n/write $O/$n/sq $O/$n/sq2 -name '*results.db') > $O/${n}_pmc.csv
  grep -h "^{" $O/$n.trace.log | tail -3 > $O/${n}_lines_under_rocprof.jsonl
  rm -rf $O/$n
}
run c2_lanemajor python bench.py --layout lane --no-cpu --steps 100 --warmup 5
run c4 python tools/perf_configs.py --only c4 --iters 10
for n in c2_lanemajor c4; do echo "== $n"; head -4 $O/${n}_kernel_stats.csv | cut -c1-220; grep -E "FETCH_SIZE|WRITE_SIZE" $O/${n}_pmc.csv | head -6 | cut -c1-220; done
