#!/bin/bash
# Whole-engine variants with one object (UNIT, default biquad_i32_df1: the LaneMajor biquad without its loads / stores) rebuilt with -D switches
#   bash tools/exp_lm_ablate.sh build ; gpurun -- 'bash tools/exp_lm_ablate.sh run'
set -u
UNIT=${UNIT:-biquad_i32_df1}
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
VARIANTS=${VARIANTS:-"NOSTORE:-DIDSP_EXP_LM_NOSTORE NOLOAD:-DIDSP_EXP_LM_NOLOAD COMPUTE:-DIDSP_EXP_LM_NOSTORE,-DIDSP_EXP_LM_NOLOAD"}
FLAGS="--offload-arch=gfx950 --offload-compress -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -fwrapv -Wall -Wno-unused-function -Wno-pass-failed -Iinclude"
if [ "${1:-run}" = build ]; then
  mkdir -p build/exp_lm
  for v in $VARIANTS; do
    n=${v%%:*}; d=${v#*:}
    /opt/rocm/bin/hipcc $FLAGS ${d//,/ } -c idsp_amd/csrc/${UNIT}.hip -o build/exp_lm/${UNIT}_$n.o &
  done
  wait
  for v in $VARIANTS; do
    n=${v%%:*}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -shared -fPIC -o build/exp_lm/full_$n.so $(ls idsp_amd/csrc/*.o | grep -v ${UNIT}.o) build/exp_lm/${UNIT}_$n.o
  done
  ls -la build/exp_lm/*.so
else
  O=gpurun_out/${OUT:-exp_lm_ablate.jsonl}; mkdir -p gpurun_out; : > $O
  echo '{"variant": "product"}' >> $O
  python tools/perf_configs.py --only ${ONLY:-c2} --iters ${ITERS:-10} 2>/dev/null | grep '^{' >> $O
  for v in $VARIANTS; do
    n=${v%%:*}
    echo "{\"variant\": \"$n\"}" >> $O
    IDSP_HIP_LIB=$PWD/build/exp_lm/full_$n.so python tools/perf_configs.py --only ${ONLY:-c2} --iters ${ITERS:-10} 2>/dev/null | grep '^{' >> $O
  done
  cut -c1-150 $O
fi
