# fm_disc LaneMajor: start-up stagger patterns ((b >> shift) % mod) * ticks of 10 ns; CFGS="ticks shift mod;..."
IFS=';' read -ra L <<< "${CFGS:-0 4 4;600 4 4;1200 4 4;2400 4 4;1200 3 4;1200 5 4;1200 4 8;2400 5 4;1200 6 4;1200 8 4;600 4 8;2400 3 8;4800 4 4}"
for cfg in "${L[@]}"; do set -- $cfg; echo "{\"skew\": [$1, $2, $3]}"; IDSP_DIAG=1 IDSP_FMD_SKEW=$1 IDSP_FMD_SKEW_SHIFT=$2 IDSP_FMD_SKEW_MOD=$3 python tools/perf_configs.py --only fm 2>/dev/null | grep "LM 65536\|LM 32768" | cut -c1-110; done
