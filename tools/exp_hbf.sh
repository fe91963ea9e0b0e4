#!/bin/bash
# C3 experiment: where does the /16 decimator's time go?  Builds variants of hbf_wave_dec.o (no arithmetic / no input
# loads / single stages skipped), links each into its own library and times C3 with tools/perf_configs.py.
# Results are WRONG by construction in these variants: timing only.  Run on the build host first (compiles), then
# through gpurun:  bash tools/exp_hbf.sh build ; gpurun -- 'bash tools/exp_hbf.sh run'
set -u
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
VARIANTS=${VARIANTS:-"CH2048:-DIDSP_HBF_CH=2048 CH512:-DIDSP_HBF_CH=512 NOSTAGES:-DIDSP_EXP_HBF_NOSTAGES NOLOAD:-DIDSP_EXP_HBF_NOLOAD SKIP0:-DIDSP_EXP_HBF_SKIP=1 SKIP1:-DIDSP_EXP_HBF_SKIP=2 SKIP2:-DIDSP_EXP_HBF_SKIP=4 SKIP3:-DIDSP_EXP_HBF_SKIP=8 SKIP123:-DIDSP_EXP_HBF_SKIP=14"}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -fwrapv -Wall -Wno-unused-function -Iinclude"
if [ "${1:-run}" = build ]; then
  mkdir -p build/exp_hbf
  for v in $VARIANTS; do
    n=${v%%:*}; d=${v#*:}
    /opt/rocm/bin/hipcc $FLAGS ${d//,/ } -c idsp_amd/csrc/hbf_wave_dec.hip -o build/exp_hbf/hbf_wave_dec_$n.o &
  done
  wait
  for v in $VARIANTS; do
    n=${v%%:*}
    objs=$(ls idsp_amd/csrc/*.o | grep -v hbf_wave_dec.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/exp_hbf/libidsp_hip_$n.so $objs build/exp_hbf/hbf_wave_dec_$n.o
  done
  ls -la build/exp_hbf/*.so
else
  O=gpurun_out/exp_hbf.jsonl; mkdir -p gpurun_out; : > $O
  echo '{"variant": "product"}' >> $O
  python tools/perf_configs.py --only c3 --iters 10 2>/dev/null | grep hbf_dec >> $O
  for v in $VARIANTS; do
    n=${v%%:*}
    echo "{\"variant\": \"$n\"}" >> $O
    IDSP_HIP_LIB=$PWD/build/exp_hbf/libidsp_hip_$n.so python tools/perf_configs.py --only c3 --iters 10 2>/dev/null | grep hbf_dec >> $O
  done
  cat $O
fi
