#!/usr/bin/env python3
"""Copy the summaries `run_profile.sh <tag>` left under gpurun_out/ into profiles/<tag>/ (tracked) and refresh
sweep_pmc_latest.json, the HBM-traffic figure bench.py quotes in roofline.traffic.

  python3 profiles/collect.py r01_coef [kernel-name-substring]
"""
import json
import os
import shutil
import sys

tag = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "pair_sweep_kernel"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "profiles_" + tag)
raw = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles", tag)
os.makedirs(dst, exist_ok=True)
for f in ("kernel_stats.csv", "pmc_summary.json"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f))
if os.path.exists(os.path.join(raw, "bench_trace.json")):
    shutil.copy(os.path.join(raw, "bench_trace.json"), os.path.join(dst, "bench_under_rocprof.json"))
pmc = json.load(open(os.path.join(src, "pmc_summary.json")))


def per_launch(counter):
    for k, v in pmc.get(counter, {}).items():
        if kernel in k:
            return k, v["mean_value_per_launch"]
    return None, None


def rocprof_mean_ms():
    import csv

    for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))):
        if kernel in r.get("Name", ""):
            return float(r["AverageNs"]) * 1e-6, int(r["Calls"])
    return None, 0


name, fetch_kb = per_launch("FETCH_SIZE")
_, write_kb = per_launch("WRITE_SIZE")
mean_ms, calls = rocprof_mean_ms()
# the same bench command's own HIP-event figure, side by side with the profiler's (so the two can be diffed)
cross = {"kernel": kernel, "rocprof_avg_launch_ms": mean_ms, "rocprof_calls": calls}
bj = os.path.join(dst, "bench_under_rocprof.json")
if os.path.exists(bj):
    try:
        line = [ln for ln in open(bj).read().splitlines() if ln.startswith("{")][-1]
        rf = json.loads(line)["roofline"]
        cross.update(bench_avg_launch_ms=rf["avg_launch_ms"], bench_event_pair_ms=rf.get("event_pair_ms"),
                     algorithmic_bytes_per_launch=rf["algorithmic_bytes_per_launch"],
                     frac_from_bench_events=rf["frac"],
                     frac_from_rocprof_mean=(rf["algorithmic_bytes_per_launch"] / (mean_ms * 1e-3) / 8e12
                                             if mean_ms else None))
    except Exception as e:  # keep the rocprof half
        cross["bench_error"] = repr(e)
json.dump(cross, open(os.path.join(dst, "roofline_crosscheck.json"), "w"), indent=1)
print(json.dumps(cross, indent=1))
if name is not None and kernel == "pair_sweep_kernel":  # (the figure bench.py quotes belongs to the headline workload)
    rec = {
        "kernel": name.replace("void ", ""),
        "workload": "pcn61_4096",
        "FETCH_SIZE_KB_per_launch": fetch_kb,
        "WRITE_SIZE_KB_per_launch": write_kb,
        # MI355X_MICROARCH.md: gfx950 tallies a 128-byte request as 64 B in FETCH_SIZE -> double it; KB = 1024 B
        "hbm_bytes_per_launch": 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0,
        "rocprof_avg_launch_ms": mean_ms,
        "rocprof_calls": calls,
        "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (profiles/run_profile.sh); "
                "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests as 64 B)",
        "source": "profiles/%s/pmc_summary.json" % tag,
    }
    json.dump(rec, open(os.path.join(root, "profiles", "sweep_pmc_latest.json"), "w"), indent=1)
    print(json.dumps(rec, indent=1))
