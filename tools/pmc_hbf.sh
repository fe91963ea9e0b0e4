#!/bin/bash
# SQ / LDS counters of the C3 half-band decimator (product and the timing variants of tools/exp_hbf_ring.sh) under rocprofv3.
#   gpurun -- 'bash tools/pmc_hbf.sh "product NOLOAD NOSTAGES"'
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_hbf
mkdir -p $O
cd $R
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"
SQ3="SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
for v in ${1:-product}; do
  if [ $v = product ]; then unset IDSP_HBF_LIB; else export IDSP_HBF_LIB=$R/build/exp_hbf_ring/libidsp_hip_$v.so; fi
  i=0
  for set in "$SQ1" "$SQ2" "$SQ3"; do
    i=$((i+1))
    rocprofv3 --pmc $set -d $O/$v/sq$i -o p -- python tools/perf_configs.py --only c3dec --iters 5 > $O/$v.sq$i.log 2>&1
  done
  python tools/rocpd_summary.py pmc $(find $O/$v -name '*results.db') > $O/${v}_pmc.csv
  rm -rf $O/$v
  echo "== $v"; grep hbf_dec $O/${v}_pmc.csv | cut -d, -f1-4,7 | sed 's/"hbfr::hbf_dec_ring_//; s/Lay<0, 4>//'
done
