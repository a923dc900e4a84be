"""Time line of one steady-state MC step from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline
    python tools/step_timeline.py gpurun_out/tl [step]
Prints every kernel of the step (start relative to the step's first kernel, duration, queue) and the device-idle gaps."""
import csv, glob, os, re, sys

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 200
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
ends = [i for i, r in enumerate(rows) if "publish_result_kernel" in r[2]]
if len(ends) < 10:  # the last finish launch publishes: a step ends with the finish launch that is followed by a new step's
    # first kernel (update_coef_moves / apply_moves / apply_edits)
    firsts = ("update_coef_moves_kernel", "apply_moves_kernel", "apply_edits_kernel")
    ends = [i - 1 for i, r in enumerate(rows) if i > 0 and any(f in r[2] for f in firsts) and "pair_finish_kernel" in rows[i - 1][2]]
lo, hi = ends[which - 1] + 1, ends[which]
t0 = rows[lo][0]
prev_end = rows[lo][0]
busy_until = t0
print("step %d: %d kernels, %.1f us from first start to publish end" % (which, hi - lo + 1, (rows[hi][1] - t0) / 1e3))
for s, e, name, q in rows[lo:hi + 1]:
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("mpmc::", "")
    gap = (s - busy_until) / 1e3
    print("%8.2f  %7.2f us  q%-3s %-40s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short[:40], ("idle %.2f" % gap) if gap > 0.3 else ""))
    busy_until = max(busy_until, e)
# step-to-step period
per = [(rows[ends[k + 1]][1] - rows[ends[k]][1]) / 1e3 for k in range(max(1, which - 50), min(len(ends) - 1, which + 50))]
print("mean step period around it: %.1f us" % (sum(per) / len(per)))
chain = [(rows[ends[k]][1] - rows[ends[k - 1] + 1][0]) / 1e3 for k in range(max(2, which - 100), min(len(ends), which + 100))]
chain.sort()
print("device chain (first kernel start -> publish end) over %d steps: median %.1f us, mean %.1f us, kernels per step %.1f" %
      (len(chain), chain[len(chain) // 2], sum(chain) / len(chain), (ends[-1] - ends[0]) / max(1, len(ends) - 1)))
