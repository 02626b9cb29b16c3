"""Summarises the look-ahead timeline of one N=16384 potrf from a rocprofv3 kernel trace of
tools/trailing_bench.py (third factorisation = first look-ahead run):
    python tools/la_timeline.py gpurun_out/kt/kt_kernel_trace.csv"""
import sys

import pandas as pd

t = pd.read_csv(sys.argv[1]).sort_values("Start_Timestamp").reset_index(drop=True)
t["short"] = t.Kernel_Name.str.extract(r"(\w+_kernel(?:<[^>]*>)?)")[0].fillna(t.Kernel_Name)
starts = t.index[t.short.str.startswith("gram_sym")].tolist()
run = t.iloc[starts[2] + 1: starts[3]]
t0 = run.Start_Timestamp.min()
print("kernels", len(run), "span ms %.3f" % ((run.End_Timestamp.max() - t0) / 1e6))
big = run[(run.short == "gemm_lds_kernel")]
prev_end, rows = None, []
for _, r in big.iterrows():
    gap = (r.Start_Timestamp - prev_end) / 1e3 if prev_end else 0.0
    rows.append(((r.Start_Timestamp - t0) / 1e6, (r.End_Timestamp - r.Start_Timestamp) / 1e3, gap,
                 r.Grid_Size_X // 256))
    prev_end = r.End_Timestamp
for x in rows[::4]:
    print("start %.3f ms dur %.1f us gap %.1f us wgs %d" % x)
print("sum bulk ms %.2f, sum gaps ms %.2f, first start %.3f, after last bulk ms %.2f" % (
    sum(x[1] for x in rows) / 1e3, sum(x[2] for x in rows) / 1e3, rows[0][0],
    (run.End_Timestamp.max() - t0) / 1e6 - rows[-1][0] - rows[-1][1] / 1e3))
g = run.assign(dur=(run.End_Timestamp - run.Start_Timestamp) / 1e3).groupby(["Stream_Id", "short"]).dur
print(pd.DataFrame({"n": g.size(), "avg_us": g.mean().round(1), "sum_ms": (g.sum() / 1e3).round(2)}))
