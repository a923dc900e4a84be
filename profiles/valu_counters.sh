#!/bin/bash
# SQ counters under the two VALU-bound kernels on a full (non-incremental) pass -- VERDICT r02 item 7.
#   bash profiles/valu_counters.sh <tag>     -> gpurun_out/profiles_<tag>/valu_counters.json
# Two --pmc passes (8 SQ slots each), --kernel-trace only beside them (gpurun refuses other trace domains with --pmc).
set -e
TAG=${1:-rXX}
OUT=gpurun_out/prof_${TAG}_valu
mkdir -p $OUT gpurun_out/profiles_$TAG
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d $OUT/a -- python3 tools/full_pass.py 12 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/b -- python3 tools/full_pass.py 12 > $OUT/b.log 2>&1
python3 - $OUT gpurun_out/profiles_$TAG/valu_counters.json <<'PY'
import csv, glob, json, os, sys
out, dst = sys.argv[1], sys.argv[2]
want = ("pair_rd_es_kernel", "static_field_kernel", "field_coef_kernel", "pair_recip_kernel")
agg = {}
dur = {}
for sub in ("a", "b"):
    for cc in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(cc)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not any(w in k for w in want):
                continue
            # only the FULL grids (a full pass over all upper-triangle tiles): the incremental launches of the first step are few
            a = agg.setdefault(k, {}).setdefault(r["Counter_Name"], [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for kt in glob.glob(os.path.join(out, sub, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(kt)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if any(w in k for w in want):
                d = dur.setdefault(k, [0, 0.0])
                d[0] += 1
                d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
res = {}
for k, cs in agg.items():
    m = {c: v[1] / v[0] for c, v in cs.items()}
    m["launches"] = max(v[0] for v in cs.values())
    if k in dur:
        m["mean_us_under_pmc"] = dur[k][1] / dur[k][0]
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md)
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if c in m:
                m["frac_of_wave_cycles_" + c] = m[c] / wc
    if m.get("SQ_WAVES") and m.get("SQ_INSTS_VALU"):
        m["valu_insts_per_wave"] = m["SQ_INSTS_VALU"] / m["SQ_WAVES"]
    if m.get("SQ_LDS_IDX_ACTIVE"):
        m["lds_bank_conflict_share"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
    res[k] = m
json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1)[:4000])
PY
