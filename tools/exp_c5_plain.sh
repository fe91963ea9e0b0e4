# C5 shapes (f32 DF2T, 2^17 .. 2^20 lanes) on the LDS-DMA kernel with plain instead of nontemporal stores / requests: does the read / write
# interference (NOTES, "without its memory traffic") change?  Three processes per variant: the result depends on where the buffers land.
for rep in 1 2 3; do
  for v in product PLAINST PLAINLD PLAINBOTH; do
    if [ $v = product ]; then unset IDSP_HIP_LIB; else export IDSP_HIP_LIB=$PWD/build/exp_lm/full_$v.so; fi
    python tools/perf_configs.py --only c5sweep 2>/dev/null | grep "^{" | grep "df2t" | sed "s/C5s:/$v:/" | cut -c1-120
  done
done
