#!/usr/bin/env python3
"""The per-size table of DESIGN.md section 5 from a round's bench lines:
   python3 profiles/size_table.py profiles/r03_jacobi"""
import json
import os
import sys

d = sys.argv[1]
rows = []
for f in ("bench.json", "bench_other_workloads.jsonl", "bench_uvt.json", "bench_pcn61_21183_uvt.json"):
    p = os.path.join(d, f)
    if not os.path.exists(p):
        continue
    for ln in open(p):
        if ln.startswith("{"):
            rows.append(json.loads(ln))
print("| workload | atoms (polarizable) | steps/s | dominant kernel, launch | frac of 8 TB/s (bytes of this design) | "
      "rebuilt from scratch, steps/s | CPU port, 1 pinned core, steps/s | GPU / CPU |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    c, rf, cb = r["config"], r["roofline"], r.get("cpu_baseline")
    name = c["workload"].split(";")[0]
    kern = rf["kernel"].split(" (")[0]
    frac = ("%.3f" % rf["frac"]) if rf["avg_launch_ms"] > 0 else "—"
    launch = ("`%s` %.1f µs" % (kern, 1e3 * rf["avg_launch_ms"])) if rf["avg_launch_ms"] > 0 else "— (no dipole sweep)"
    if "resident" in kern or "folded" in kern:
        frac += " (effective: one launch per solve)"
    cpu = ("%.3g (%s)" % (cb["value"], cb["cpu_model"].replace(" Processor", ""))) if cb else "— (see text)"
    ratio = ("%.0f×" % (r["value"] / cb["value"])) if cb else "—"
    print("| %s | %d (%d) | %.0f | %s | %s | %.0f | %s | %s |" % (name, c.get("n_atoms_final", c["n_atoms"]), c.get("n_polarizable_final", c["n_polarizable"]),
                                                                r["value"], launch, frac, r["full_rebuild_steps_per_s"], cpu, ratio))
