#!/usr/bin/env python3
"""profiles/bench_<config>_traffic.json (what bench.py quotes as `roofline.traffic`) from the round's PMC summaries
(profiles/rNN_<run>_pmc.csv, tools/collect_profiles_rNN.sh): HBM bytes per launch of the dominant kernel = FETCH_SIZE x 2
(gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE, both in KiB.  Round 6: also the second roof
bench.py prints as `roofline.issue` — how busy the busiest issue unit is, SQ_ACTIVE_INST_{VALU, LDS} (quad-cycles summed over the
waves) x 4 / (1024 SIMDs resp. 256 CUs) / (GRBM_GUI_ACTIVE / 8: the counter is summed over the XCDs) — and the clock the launch ran at (TCC_BUSY_avr / dispatch duration:
the L2 runs on the shader clock, and these launches are power-bound in clock, profiles/NOTES.md round 6).
    python tools/make_traffic_json.py r06"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
# run name -> (bench config key, name prefix idsp_last_kernel() reports for it, bench command)
RUNS = {
    "c2_driverflags": ("c2", "stream_frame_major_sweep[", "python bench.py --no-cpu --no-c5 --no-c3 --no-c4 --no-lane-major --no-inplace --steps 20 --warmup 5"),
    "c2_inplace": ("c2_inplace", "stream_frame_major_sweep[", "python bench.py ... --inplace --steps 20 --warmup 5"),
    "c2_lanemajor": ("c2_lane", "stream_lane_major_staged", "python bench.py ... --layout lane --steps 100 --warmup 5"),
    "c5": ("c5", "stream_frame_major_sweep[", "python bench.py --config c5 --no-cpu --steps 20 --warmup 5"),
    "c5_inplace": ("c5_inplace", "stream_frame_major_sweep[", "python bench.py --config c5 --inplace --no-cpu --steps 20 --warmup 5"),
    "c3": ("c3", "hbf_dec_ring[FrameMajor]", "python bench.py --config c3 --no-cpu --steps 20 --warmup 5"),
    "c3_lanemajor": ("c3_lane", "hbf_dec_blk[LaneMajor]", "python bench.py --config c3 --layout lane --no-cpu --steps 20 --warmup 5"),
    "c4": ("c4", "lockin_waves_kernel", "python bench.py --config c4 --no-cpu --steps 20 --warmup 5"),
    "c4_lanemajor": ("c4_lane", "lockin_waves_kernel", "python bench.py --config c4 --layout lane --no-cpu --steps 20 --warmup 5"),
}
for run, (key, prefix, cmd) in RUNS.items():
    path = os.path.join(ROOT, "profiles", f"{tag}_{run}_pmc.csv")
    if not os.path.exists(path):
        continue
    allrows = [r for r in csv.DictReader(open(path)) if not r["kernel"].startswith("at::")]
    rows = [r for r in allrows if r["counter"] in ("FETCH_SIZE", "WRITE_SIZE")]
    # the dominant kernel = the one with the largest FETCH_SIZE x dispatches
    fetch = max((r for r in rows if r["counter"] == "FETCH_SIZE"), key=lambda r: float(r["avg_value"]) * int(r["dispatches"]))
    write = next(r for r in rows if r["counter"] == "WRITE_SIZE" and r["kernel"] == fetch["kernel"])
    f, w = float(fetch["avg_value"]), float(write["avg_value"])
    out = {
        "source": f"profiles/{tag}_{run}_pmc.csv (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `{cmd}`, tools/collect_profiles_{tag}.sh)",
        "kernel": fetch["kernel"], "kernel_prefix": prefix,
        "fetch_size_kib_raw": round(f, 4), "fetch_correction": 2.0,
        "fetch_correction_note": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 B (MI355X_MICROARCH.md, HBM section): raw x2",
        "write_size_kib": round(w, 4),
        "traffic_bytes_per_launch": int(round((2.0 * f + w) * 1024)),
    }
    c = {r["counter"]: (float(r["avg_value"]), float(r["avg_dispatch_us"])) for r in allrows if r["kernel"] == fetch["kernel"]}
    if all(k in c for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE")):
        cyc = c["GRBM_GUI_ACTIVE"][0] / 8.0  # the counter comes summed over the eight XCDs (C2: 6.61 M for a 342 us launch = 8 x 2.42 GHz)
        valu, lds = c["SQ_ACTIVE_INST_VALU"][0] * 4 / 1024 / cyc, c["SQ_ACTIVE_INST_LDS"][0] * 4 / 256 / cyc
        issue = {
            "unit": "busiest of VALU issue (per SIMD) / LDS issue (per CU)", "busy_frac": round(max(valu, lds), 4),
            "valu_busy_frac": round(valu, 4), "lds_busy_frac": round(lds, 4), "kernel_cycles": int(cyc),
            "formula": "SQ_ACTIVE_INST_{VALU,LDS} x 4 / (1024 SIMDs | 256 CUs) / (GRBM_GUI_ACTIVE / 8 XCDs)",
            "source": f"profiles/{tag}_{run}_pmc.csv",
        }
        if "TCC_BUSY_avr" in c:
            issue["l2_clock_ghz"] = round(c["TCC_BUSY_avr"][0] / (c["TCC_BUSY_avr"][1] * 1e3), 3)
            issue["l2_clock_note"] = ("TCC_BUSY_avr / dispatch duration: the shader clock the launch ran at; 2.1-2.4 GHz when nothing throttles it — a launch "
                                      "far below that is bound by the power it draws (fewer instructions and LDS bytes per sample buy clock)")
        if "TCC_EA0_RDREQ_LEVEL_sum" in c and "TCC_EA0_RDREQ_sum" in c and c["TCC_EA0_RDREQ_sum"][0] > 0:
            issue["ea_read_latency_cycles"] = round(c["TCC_EA0_RDREQ_LEVEL_sum"][0] / c["TCC_EA0_RDREQ_sum"][0], 1)
        out["issue"] = issue
    with open(os.path.join(ROOT, "profiles", f"bench_{key}_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(key, out["kernel"][:70], out["traffic_bytes_per_launch"])
