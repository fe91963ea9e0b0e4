#!/bin/bash
# Round-6 rocprofv3 evidence (passes of rounds 2-5 + one TCC pass: L2 busy cycles = the clock the launch ran at, EA read requests in flight = memory latency;
# every rocprofv3 under `timeout`: a pass that aborts must not hang the box), collected on the GPU box through gpurun:
#   bash tools/collect_profiles_r06.sh [tag] [what]
# For bench.py (C2, and C5 on one GPU) and for the C3 / C4 lines of tools/perf_configs.py:
#   pass 1  --kernel-trace --stats                      -> kernel durations
#   pass 2  --pmc FETCH_SIZE                            -> HBM read bytes  (x2 on gfx950, MI355X_MICROARCH.md)
#   pass 3  --pmc WRITE_SIZE                            -> HBM write bytes (FETCH_SIZE takes 3 of 4 TCC slots: separate passes)
#   pass 4  --pmc SQ_* (8 slots) + GRBM_GUI_ACTIVE      -> where the wave cycles go; kernel cycles (the issue roof of bench.py: busiest of VALU / LDS)
#   pass 6  --pmc TCC_BUSY_avr TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum -> L2 clock (busy cycles / duration), EA read latency
# PMC passes never share a run with a trace domain other than the kernel trace.
# Output: gpurun_out/prof_<tag>/{c2,c5,c3,c4}_{kernel_stats,pmc}.csv (+ the bench lines printed under the profiler).
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd $R
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
TCC="TCC_BUSY_avr TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum"
T="timeout ${PASS_TIMEOUT:-240}"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"
run() {  # name, command...
  local n=$1; shift
  has() { echo " ${PASSES:-trace fetch write sq sq2 tcc} " | grep -q " $1 "; }  # PASSES="sq tcc": a subset (diagnosis runs)
  has trace && $T rocprofv3 --kernel-trace --stats -d $O/$n/trace -o p -- "$@" > $O/$n.trace.log 2>&1
  has fetch && $T rocprofv3 --pmc FETCH_SIZE -d $O/$n/fetch -o p -- "$@" > $O/$n.fetch.log 2>&1
  has write && $T rocprofv3 --pmc WRITE_SIZE -d $O/$n/write -o p -- "$@" > $O/$n.write.log 2>&1
  has sq && $T rocprofv3 --pmc $SQ -d $O/$n/sq -o p -- "$@" > $O/$n.sq.log 2>&1
  has sq2 && $T rocprofv3 --pmc $SQ2 -d $O/$n/sq2 -o p -- "$@" > $O/$n.sq2.log 2>&1
  has tcc && $T rocprofv3 --pmc $TCC -d $O/$n/tcc -o p -- "$@" > $O/$n.tcc.log 2>&1
  has trace && python tools/rocpd_summary.py stats $(find $O/$n/trace -name '*results.db' | head -1) > $O/${n}_kernel_stats.csv
  python tools/rocpd_summary.py pmc $(find $O/$n -name '*results.db' -not -path '*/trace/*') > $O/${n}_pmc.csv
  has trace && grep -h "^{" $O/$n.trace.log | tail -12 > $O/${n}_lines_under_rocprof.jsonl
  rm -rf $O/$n  # the sqlite databases are large; the summaries are what gets committed
}
WHAT=${2:-all}
# usage: collect_profiles_r06.sh [tag] [all | comma list of c2,c2df,c2ip,c2lm,c5,c5ip,c5s8,c3,c3lm,c4,c4lm,cic]
want() { [ "$WHAT" = all ] || echo ",$WHAT," | grep -q ",$1,"; }
want c2 && run c2 python bench.py --no-cpu --no-c5 --no-c3 --no-c4 --no-lane-major --no-inplace --no-guard --steps 100 --warmup 5
want c2df && run c2_driverflags python bench.py --no-cpu --no-c5 --no-c3 --no-c4 --no-lane-major --no-inplace --no-guard --steps 20 --warmup 5
want c2lm && run c2_lanemajor python bench.py --no-cpu --no-c5 --no-c3 --no-c4 --no-lane-major --no-inplace --no-guard --layout lane --steps 100 --warmup 5
want c2ip && run c2_inplace python bench.py --no-cpu --no-c5 --no-c3 --no-c4 --no-lane-major --no-inplace --no-guard --inplace --steps 20 --warmup 5
want c5 && run c5 python bench.py --config c5 --no-cpu --steps 20 --warmup 5
want c5ip && run c5_inplace python bench.py --config c5 --inplace --no-cpu --steps 20 --warmup 5
want c5s8 && run c5_shard8 python bench.py --config c5 --lanes 131072 --no-cpu --steps 50 --warmup 5
# C3 / C4 from bench.py alone: one kernel name = one shape (FRAME_MAJOR, the layout of the driver's line), then LANE_MAJOR
want c3 && run c3 python bench.py --config c3 --no-cpu --steps 20 --warmup 5
want c3lm && run c3_lanemajor python bench.py --config c3 --layout lane --no-cpu --steps 20 --warmup 5
want c4 && run c4 python bench.py --config c4 --no-cpu --steps 20 --warmup 5
want c4lm && run c4_lanemajor python bench.py --config c4 --layout lane --no-cpu --steps 20 --warmup 5
# the Cic kernels at 16384 lanes x 4096 chunks of 16 (tools/perf_configs.py --only cic: one shape per kernel name)
want cic && run cic python tools/perf_configs.py --only cic --iters 20
# SURVEY 8(f) surfaces far below the HBM roof: issue-roof passes (tools/issue_roof.py)
want readouts && run readouts python tools/perf_configs.py --only readouts --iters 10
# ... and the small lane counts of the biquads
want small && run small python tools/perf_configs.py --only ragged --iters 10
# the LaneMajor fm_disc role kernel at 65536 lanes x 4096 frames
want fmlm && run fm_disc_lanemajor python tools/perf_configs.py --only fmlm --iters 20
for n in c2 c2_driverflags c2_inplace c2_lanemajor c5 c5_inplace c5_shard8 c3 c3_lanemajor c4 c4_lanemajor cic fm_disc_lanemajor readouts; do [ -f $O/${n}_kernel_stats.csv ] || continue; echo "== $n"; head -4 $O/${n}_kernel_stats.csv | cut -c1-220; grep -E "FETCH_SIZE|WRITE_SIZE" $O/${n}_pmc.csv | head -6 | cut -c1-220; done
