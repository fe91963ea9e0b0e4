#!/bin/bash
# fm_disc role waves + branch-free atan2: parity of everything that calls atan2_dev, timing per form
mkdir -p gpurun_out/p
python -m pytest tests/test_fm_disc.py tests/test_gpu_host_mirror.py tests/test_gpu_parity.py tests/test_gpu_lockin_fuzz.py tests/test_gpu_full_tensor_oracle.py tests/test_gpu_kat.py -m gpu -x -q -k "fm_disc or atan2 or lockin or arg or polar or c4 or kat" > gpurun_out/p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/p/pytest.log
tail -5 gpurun_out/p/pytest.log
for w in 4 0; do IDSP_DIAG=1 IDSP_FM_DISC_WAVES=$w python tools/perf_configs.py --only fm 2>&1 | grep fm_disc | sed "s/^/waves=$w /"; done | tee gpurun_out/p/perf_fm_disc.jsonl
python tools/perf_configs.py --only c4 2>&1 | grep -v "^/opt" | tee gpurun_out/p/perf_c4.jsonl
