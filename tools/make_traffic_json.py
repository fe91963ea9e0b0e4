#!/usr/bin/env python3
"""profiles/bench_<config>_traffic.json (what bench.py quotes as `roofline.traffic`) from the round's PMC summaries
(profiles/rNN_<run>_pmc.csv, tools/collect_profiles_rNN.sh): HBM bytes per launch of the dominant kernel = FETCH_SIZE x 2
(gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE, both in KiB.
    python tools/make_traffic_json.py r05"""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
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
    rows = [r for r in csv.DictReader(open(path)) if r["counter"] in ("FETCH_SIZE", "WRITE_SIZE") and not r["kernel"].startswith("at::")]
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
    with open(os.path.join(ROOT, "profiles", f"bench_{key}_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(key, out["kernel"][:70], out["traffic_bytes_per_launch"])
