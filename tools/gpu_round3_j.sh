#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out
python -m pytest tests/test_gpu_ragged_lds_block.py tests/test_gpu_pitch.py tests/test_gpu_dispatch_fuzz.py tests/test_gpu_misaligned.py tests/test_gpu_lds_path_forced.py tests/test_gpu_frame_major_staged.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/r03_tests_rot.log 2>&1; echo "rc=$?" >> $O/r03_tests_rot.log
(cd tools && python exp_fm_pitch.py 2>&1 | grep -v libdrm > ../$O/r03_exp_fm_pitch_xcdc.jsonl)
python tools/perf_configs.py --only ragged 2>&1 | grep -v libdrm > $O/r03_perf_ragged_xcdc.jsonl
build/exp_fm_mis >> $O/r03_exp_fm_misaligned.jsonl
tail -6 $O/r03_tests_rot.log; cut -c1-200 $O/r03_exp_fm_pitch_xcdc.jsonl; cut -c1-160 $O/r03_perf_ragged_xcdc.jsonl
