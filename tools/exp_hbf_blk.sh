#!/bin/bash
# C3 experiments on the register-blocked half-band decimator (hbf_blk.h): builds variants of hbf_blk_dec.o — timing-only ones
# (no arithmetic / no input requests: results WRONG by construction) and geometry ones (IDSP_HBF_BLK_PM: which stages run at
# four outputs per thread) — links each into a small half-band-only library and times C3 with tools/perf_configs.py next to the
# product, the ring kernels (IDSP_HBF_NO_BLK) and the LDS-padding occupancy sweep.
#   bash tools/exp_hbf_blk.sh build ; gpurun -- 'bash tools/exp_hbf_blk.sh run'
set -u
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
VARIANTS=${VARIANTS:-"NOSTAGES:-DIDSP_EXP_HBF_NOSTAGES NOLOAD:-DIDSP_EXP_HBF_NOLOAD NOSTORE:-DIDSP_EXP_HBF_NOSTORE PHASES:-DIDSP_EXP_HBF_PHASES PM0a:-DIDSP_HBF_BLK_PM=0x0a PM02:-DIDSP_HBF_BLK_PM=0x02 WAVE:-DIDSP_HBF_BLK_OFF"}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -fwrapv -Wall -Wno-unused-function -Wno-pass-failed -Iinclude"
D=build/exp_hbf_blk
if [ "${1:-run}" = build ]; then
  mkdir -p $D
  for v in $VARIANTS; do
    n=${v%%:*}; d=${v#*:}
    /opt/rocm/bin/hipcc $FLAGS ${d//,/ } -c idsp_amd/csrc/hbf_blk_dec.hip -o $D/hbf_blk_dec_$n.o &
  done
  wait
  for v in $VARIANTS; do
    n=${v%%:*}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o $D/libidsp_hip_$n.so idsp_amd/csrc/hbf.o \
      idsp_amd/csrc/hbf_wave_dec.o idsp_amd/csrc/hbf_wave_int.o idsp_amd/csrc/hbf_ring_dec.o idsp_amd/csrc/api_util.o $D/hbf_blk_dec_$n.o
  done
  ls -la $D/*.so
else
  O=gpurun_out/${OUT:-exp_hbf_blk.jsonl}; mkdir -p $(dirname $O); : > $O
  echo '{"variant": "product"}' >> $O
  python tools/perf_configs.py --only c3dec --iters ${ITERS:-10} 2>/dev/null | grep hbf_dec >> $O
  echo '{"variant": "ring kernels of round 4 (IDSP_HBF_NO_BLK)"}' >> $O
  IDSP_DIAG=1 IDSP_HBF_NO_BLK=1 python tools/perf_configs.py --only c3dec --iters ${ITERS:-10} 2>/dev/null | grep hbf_dec >> $O
  for pad in ${PADS:-1024 2048 4096}; do
    echo "{\"variant\": \"product, LaneMajor with $pad more bytes of LDS per wave\"}" >> $O
    IDSP_DIAG=1 IDSP_HBF_LDS_PAD=$pad python tools/perf_configs.py --only c3dec --iters ${ITERS:-10} 2>/dev/null | grep "hbf_dec.*LM" >> $O
  done
  for v in $VARIANTS; do
    n=${v%%:*}
    echo "{\"variant\": \"$n\"}" >> $O
    IDSP_HBF_LIB=$PWD/$D/libidsp_hip_$n.so python tools/perf_configs.py --only c3dec --iters ${ITERS:-10} 2>/dev/null | grep hbf_dec >> $O
  done
  cat $O
fi
