#!/bin/bash
# Memory-side counters of the C3 half-band decimator (product and variants of tools/exp_hbf_blk.sh) under rocprofv3: EA read
# latency (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ), DRAM credit stalls, texture-addresser busy / stalled, VMEM instruction latency.
#   gpurun -- 'bash tools/pmc_hbf_mem.sh "product NOSTAGES NOSTORE RING" lm'
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_hbf_mem
mkdir -p $O
cd $R
LAYOUT=${2:-lm}
P1="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_avr"
P2="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
P3="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES"
P4="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum"
for v in ${1:-product}; do
  i=0
  for set in "$P1" "$P3" "$P4"; do  # P2 (TA_*) aborts rocprofv3 on this image and then hangs: left out
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $set -d $O/$v/p$i -o p -- python tools/exp_hbf_ab.py --layout $LAYOUT --rounds 1 --iters 5 $v > $O/$v.p$i.log 2>&1
  done
  python tools/rocpd_summary.py pmc $(find $O/$v -name '*results.db') > $O/${v}_${LAYOUT}_pmc.csv
  rm -rf $O/$v
  echo "== $v"; grep -i "hbf_dec" $O/${v}_${LAYOUT}_pmc.csv | cut -c1-400
done
