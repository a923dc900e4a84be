#!/usr/bin/env python3
"""Condense rocprofv3 csv output into the small summaries that get committed under profiles/."""
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
dst = os.path.join("gpurun_out", "profiles_" + tag)
os.makedirs(dst, exist_ok=True)


def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return r[0] if r else None


st = find("trace", "*kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    with open(os.path.join(dst, "kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)

summary = {}
for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    cc = find(sub, "*counter_collection.csv")
    if not cc:
        continue
    agg = {}
    for r in csv.DictReader(open(cc)):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    summary[counter] = {k: {"launches": v[0], "mean_value_per_launch": v[1] / v[0]} for k, v in agg.items()}
json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(summary, indent=1)[:3000])
