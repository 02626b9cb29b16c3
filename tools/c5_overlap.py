"""Reads a rocprofv3 kernel trace of tools/c5_time.py and reports, for the last plan run, how
the wall time splits into: trailing GEMM running (alone / beside panel kernels), panel kernels
only, nothing running."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"])
        for r in csv.DictReader(open(f))]
rows.sort()
# the last run starts at the last assemble kernel group
starts = [i for i, r in enumerate(rows) if "assemble" in r[2]]
first = starts[-1]
while first > 0 and "assemble" in rows[first - 1][2]:
    first -= 1
run = rows[first:]
t0, t1 = run[0][0], max(r[1] for r in run)
ev = []
for s, e, n in run:
    k = "gemm" if "gemm_lds" in n else "panel"
    ev.append((s, 1, k))
    ev.append((e, -1, k))
ev.sort()
cnt = {"gemm": 0, "panel": 0}
acc = {"gemm only": 0, "gemm+panel": 0, "panel only": 0, "idle": 0}
prev = t0
for t, d, k in ev:
    dt = t - prev
    if cnt["gemm"] and cnt["panel"]:
        acc["gemm+panel"] += dt
    elif cnt["gemm"]:
        acc["gemm only"] += dt
    elif cnt["panel"]:
        acc["panel only"] += dt
    else:
        acc["idle"] += dt
    cnt[k] += d
    prev = t
print("span ms", (t1 - t0) / 1e6)
for k, v in acc.items():
    print("%-12s %.3f ms" % (k, v / 1e6))
by = {}
for s, e, n in run:
    key = n.split("(")[0][:40]
    by.setdefault(key, [0, 0])
    by[key][0] += 1
    by[key][1] += e - s
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("%-42s %4d %8.3f ms" % (k, v[0], v[1] / 1e6))
