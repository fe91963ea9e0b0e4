#!/bin/bash
# builds (here, no GPU needed) or runs (on the GPU box) the variants of tools/exp_lockin_ablate.hip
#   bash tools/exp_lockin_ablate.sh build ; gpurun -- 'bash tools/exp_lockin_ablate.sh run'
set -u
cd "$(dirname "$0")/.."
mkdir -p build/exp_lockin_ablate gpurun_out
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fwrapv -ffp-contract=off -fno-slp-vectorize -Iinclude -Iidsp_amd/csrc"
declare -A V=( [product]="" [nostore]="-DIDSP_LW_ABL_NOSTORE" [noload]="-DIDSP_LW_ABL_NOLOAD" [compute]="-DIDSP_LW_ABL_NOSTORE -DIDSP_LW_ABL_NOLOAD" )
if [ "${1:-build}" = build ]; then
  for v in "${!V[@]}"; do
    ( hipcc $FLAGS ${V[$v]} ${EXTRA:-} -DVARIANT="\"$v${TAG:-}\"" tools/exp_lockin_ablate.hip -o build/exp_lockin_ablate/$v${TAG:-} || echo "build of $v failed" ) &
  done
  wait
else
  for v in product nostore noload compute; do
    for t in ${TAGS:-""}; do ./build/exp_lockin_ablate/$v$t; done
  done | tee -a gpurun_out/exp_lockin_ablate.jsonl
fi
