# round 5: the sweep kernel's FULL blocks on rows off the 64-byte grid (row pitch = lanes + 4 or + 1), by blocks per workgroup, block order and schedule
S="65536:65536 65536:65540 65536:65537 65536:65544 131072:131076 262144:262148 524288:524292 1048576:1048580"
echo "== sweep from any lane count off the grid (requests as 60 + 4 lanes)"; IDSP_DIAG=1 IDSP_SWEEP_OFFGRID_MIN_LANES=1 timeout 300 python tools/exp_fm_pitch.py $S 2>/dev/null
echo "== same, one instruction per request"; IDSP_DIAG=1 IDSP_SWEEP_OFFGRID_MIN_LANES=1 IDSP_SWEEP_NO_SPLIT_REQUESTS=1 timeout 300 python tools/exp_fm_pitch.py $S 2>/dev/null
echo "== same, plain block order"; IDSP_DIAG=1 IDSP_SWEEP_OFFGRID_MIN_LANES=1 IDSP_SWEEP_NO_XCDC=1 timeout 300 python tools/exp_fm_pitch.py $S 2>/dev/null
echo "== default dispatch"; timeout 300 python tools/exp_fm_pitch.py 65536:65540 65536:65537 65537:65537 65540:65540 131073:131073 131076:131076 2>/dev/null
