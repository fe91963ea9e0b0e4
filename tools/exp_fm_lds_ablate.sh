echo '{"variant": "product"}'; python tools/perf_configs.py --only c2,c5 2>/dev/null | grep "^{" | grep " FM " | cut -c1-125
for v in NOSTORE NOLOAD COMPUTE; do echo "{\"variant\": \"$v\"}"
IDSP_HIP_LIB=$PWD/build/exp_lm_biquad_i32_df1/full_$v.so python tools/perf_configs.py --only c2 2>/dev/null | grep "^{" | grep " FM " | cut -c1-125
IDSP_HIP_LIB=$PWD/build/exp_lm_biquad_f32_df2t/full_$v.so python tools/perf_configs.py --only c5 2>/dev/null | grep "^{" | grep " FM " | cut -c1-125
done
