# C5 shapes on the persistent LDS-DMA kernel with a start-up stagger (-DIDSP_EXP_LDS_SKEW=ticks ...): three processes per variant
for rep in 1 2 3; do
  for v in product ${VLIST:-S1200 S5000 S20000 S5000M16}; do
    if [ $v = product ]; then unset IDSP_HIP_LIB; else export IDSP_HIP_LIB=$PWD/build/exp_lm/full_$v.so; fi
    python tools/perf_configs.py --only c5sweep 2>/dev/null | grep "^{" | grep "df2t" | sed "s/C5s:/$v:/" | cut -c1-120
  done
done
