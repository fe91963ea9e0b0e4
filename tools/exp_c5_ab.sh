#!/bin/bash
# A/B over separate processes (fresh allocations each): lanes-per-thread form vs one lane per thread on a persistent grid
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/exp_c5_ab.jsonl; : > $O
for rep in 1 2 3 4 5; do
  for L in 131072 262144; do
    for V in "default:" "lpt1:IDSP_DIAG=1 IDSP_LDS_LPT=1"; do
      n=${V%%:*}; e=${V#*:}
      line=$(env $e python bench.py --config c5 --lanes $L --no-cpu --steps 40 --warmup 5 2>/dev/null | tail -1)
      echo "{\"variant\": \"$n\", \"lanes\": $L, \"rep\": $rep, \"line\": $line}" >> $O
    done
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/exp_c5_ab.jsonl"):
    j = json.loads(l); r = j["line"]["roofline"]
    print(j["variant"], j["lanes"], j["rep"], r["frac"], r["kernel_ms"], r["kernel"][:45])
PY
