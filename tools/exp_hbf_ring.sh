#!/bin/bash
# C3 experiment on the LDS-DMA ring decimator (hbf_ring.h): builds timing-only variants of hbf_ring_dec.o (no arithmetic /
# no input requests), links each into its own library and times C3 with tools/perf_configs.py, next to the product and
# to the round-3 wave kernels (IDSP_HBF_NO_RING).  Variant results are WRONG by construction: timing only.
#   bash tools/exp_hbf_ring.sh build ; gpurun -- 'bash tools/exp_hbf_ring.sh run'
set -u
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
VARIANTS=${VARIANTS:-"NOSTAGES:-DIDSP_EXP_HBF_NOSTAGES NOLOAD:-DIDSP_EXP_HBF_NOLOAD"}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -fwrapv -Wall -Wno-unused-function -Wno-pass-failed -Iinclude"
if [ "${1:-run}" = build ]; then
  mkdir -p build/exp_hbf_ring
  for v in $VARIANTS; do
    n=${v%%:*}; d=${v#*:}
    /opt/rocm/bin/hipcc $FLAGS ${d//,/ } -c idsp_amd/csrc/hbf_ring_dec.hip -o build/exp_hbf_ring/hbf_ring_dec_$n.o &
  done
  wait
  for v in $VARIANTS; do
    n=${v%%:*}
    # a small library with the half-band objects only (1.4 MB instead of 90): perf_configs.py takes idsp_hbf_*_f32 from IDSP_HBF_LIB
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o build/exp_hbf_ring/libidsp_hip_$n.so idsp_amd/csrc/hbf.o \
      idsp_amd/csrc/hbf_wave_dec.o idsp_amd/csrc/hbf_wave_int.o idsp_amd/csrc/api_util.o build/exp_hbf_ring/hbf_ring_dec_$n.o
  done
  ls -la build/exp_hbf_ring/*.so
elif [ "$1" = full ]; then
  # whole engine with the variant object in place of hbf_ring_dec.o (for IDSP_HIP_LIB: parity runs of a variant that is meant to be correct)
  n=$2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-compress -shared -fPIC -o build/exp_hbf_ring/full_$n.so \
    $(ls idsp_amd/csrc/*.o | grep -v hbf_ring_dec.o) build/exp_hbf_ring/hbf_ring_dec_$n.o
  ls -la build/exp_hbf_ring/full_$n.so
else
  O=gpurun_out/${OUT:-exp_hbf_ring.jsonl}; mkdir -p gpurun_out; : > $O
  echo '{"variant": "product"}' >> $O
  python tools/perf_configs.py --only c3 --iters ${ITERS:-10} 2>/dev/null | grep hbf_dec >> $O
  echo '{"variant": "round-3 wave kernels (IDSP_HBF_NO_RING)"}' >> $O
  IDSP_DIAG=1 IDSP_HBF_NO_RING=1 python tools/perf_configs.py --only c3 --iters ${ITERS:-10} 2>/dev/null | grep hbf_dec >> $O
  for v in $VARIANTS; do
    n=${v%%:*}
    echo "{\"variant\": \"$n\"}" >> $O
    IDSP_HBF_LIB=$PWD/build/exp_hbf_ring/libidsp_hip_$n.so python tools/perf_configs.py --only c3 --iters ${ITERS:-10} 2>/dev/null | grep hbf_dec >> $O
  done
  cat $O
fi
