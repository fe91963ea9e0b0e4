# round 5: the sweep kernel on narrow blocks and on rows off the 64-byte grid, by schedule (diagnostic switches of fm_sweep.h)
mkdir -p gpurun_out/r05
S="61440:61440 61440:61444 122880:122880 122880:122884 100000:100000 99840:99840 57344:57344 200000:200000 65540:65540 131076:131076 300016:300016"
echo "== default"; timeout 300 python tools/exp_fm_pitch.py $S 2>/dev/null
echo "== IDSP_SWEEP_PACE=1 (narrow blocks paced like full ones)"; IDSP_DIAG=1 IDSP_SWEEP_PACE=1 timeout 300 python tools/exp_fm_pitch.py $S 2>/dev/null
echo "== IDSP_SWEEP_PACE=2 (and rows off the grid)"; IDSP_DIAG=1 IDSP_SWEEP_PACE=2 timeout 300 python tools/exp_fm_pitch.py $S 2>/dev/null
