#!/usr/bin/env python3
"""Issue roof of the kernels that sit far below the HBM roof (SURVEY 8(f) read-outs, cascades, ...): from a PMC summary
(tools/rocpd_summary.py pmc, passes SQ + SQ2 + TCC of tools/collect_profiles_r06.sh) one line per kernel —
VALU busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / kernel cycles, LDS busy = SQ_ACTIVE_INST_LDS x 4 / 256 CUs / kernel cycles,
kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the eight XCDs: C2 = 6.60 M for a 342 us launch at 2.41 GHz),
VALU instructions per wave, the clock (TCC_BUSY_avr / duration) and the wave-parked share.
    python tools/issue_roof.py gpurun_out/prof_r06/readouts_pmc.csv > profiles/r06_issue_roof_readouts.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = {}
for r in rows:
    by.setdefault(r["kernel"], {})[r["counter"]] = (float(r["avg_value"]), float(r["avg_dispatch_us"]), int(r["dispatches"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatch_us", "clock_ghz", "valu_busy", "lds_busy", "wave_parked", "valu_per_wave", "lds_per_wave", "waves", "lds_conflict_per_active"])
for k, d in sorted(by.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", (0, 0, 0))[0] * kv[1].get("SQ_WAVE_CYCLES", (0, 0, 1))[2]):
    need = ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_WAVE_CYCLES")
    if not all(c in d for c in need) or d["SQ_WAVES"][0] < 64:
        continue
    cyc = d["GRBM_GUI_ACTIVE"][0] / 8.0
    us = d["GRBM_GUI_ACTIVE"][1]
    clock = d["TCC_BUSY_avr"][0] / (d["TCC_BUSY_avr"][1] * 1e3) if "TCC_BUSY_avr" in d else cyc / (us * 1e3)
    waves = d["SQ_WAVES"][0]
    w.writerow([k[:110], f"{us:.1f}", f"{clock:.2f}", f"{d['SQ_ACTIVE_INST_VALU'][0] * 4 / 1024 / cyc:.3f}", f"{d['SQ_ACTIVE_INST_LDS'][0] * 4 / 256 / cyc:.3f}",
                f"{d['SQ_WAIT_ANY'][0] / d['SQ_WAVE_CYCLES'][0]:.2f}" if "SQ_WAIT_ANY" in d else "",
                f"{d['SQ_INSTS_VALU'][0] / waves:.0f}" if "SQ_INSTS_VALU" in d else "", f"{d['SQ_INSTS_LDS'][0] / waves:.0f}" if "SQ_INSTS_LDS" in d else "",
                f"{waves:.0f}", f"{d['SQ_LDS_BANK_CONFLICT'][0] / max(d['SQ_ACTIVE_INST_LDS'][0], 1):.2f}" if "SQ_LDS_BANK_CONFLICT" in d else ""])
