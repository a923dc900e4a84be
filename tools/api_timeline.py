"""Host-side time line of one steady-state MC step: HIP API calls (rocprofv3 --hip-trace) merged with the kernels they start
(--kernel-trace), to see where the device waits for the host.
    rocprofv3 --hip-trace --kernel-trace --output-format csv -d gpurun_out/api -- python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline
    python tools/api_timeline.py gpurun_out/api [step]"""
import csv, glob, os, re, sys

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 200
kern = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("mpmc::", "")))
api = []
for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "A " + r["Function"]))
kern.sort()
firsts = [i for i, k in enumerate(kern) if "update_coef_moves_kernel" in k[2] or "apply_moves_kernel" in k[2] and (i == 0 or "update_coef_moves" not in kern[i - 1][2])]
firsts = [i for i, k in enumerate(kern) if "update_coef_moves_kernel" in k[2]]
t0 = kern[firsts[which]][0]
t1 = kern[firsts[which + 1]][0]
rows = [r for r in kern + api if t0 - 60000 <= r[0] < t1]
rows.sort()
for s, e, name in rows:
    print("%9.2f  %8.2f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, name[:70]))
