"""Timeline of one N=16384 factorisation at the engine's own block size.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/potrf_timeline.py run
    python tools/potrf_timeline.py DIR
The analysis takes the last factorisation (after the last Gram launch): when the first bulk
update starts, the bulk updates' durations and the gaps between them, what follows the last."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import numpy as np
    from bayesian_quadrature_amd import Engine, _lib as L_, workloads as wl
    e = Engine(0)
    n = int(os.environ.get("TL_N", "16384"))
    c4 = wl.c4(n)
    w4 = np.ascontiguousarray(c4["w"])
    xd, Kd, info = e.alloc(8 * n), e.alloc(8 * n * n), e.alloc(64)
    e.upload(xd, np.ascontiguousarray(c4["x"]))
    for rep in range(3):
        e._check(e._lib.bq_gram_gauss_dev(e._ctx, xd, 1, n, c4["h"], L_.dptr(w4), c4["s"], Kd, n))
        e.sync()
        e.timer_start()
        e._check(e._lib.bq_potrf_dev(e._ctx, Kd, n, n, info))
        print("potrf ms %.3f" % e.timer_stop_ms(), flush=True)
    e.close()
    sys.exit(0)

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", ""))
        for r in csv.DictReader(open(f))]
rows.sort()
last_gram = max(i for i, r in enumerate(rows) if "gram_" in r[2])
run = rows[last_gram + 1:]
t0 = rows[last_gram][1]
t1 = max(r[1] for r in run)
print("kernels %d, span %.3f ms" % (len(run), (t1 - t0) / 1e6))
big = [r for r in run if "gemm_lds" in r[2]]
print("first bulk update starts at %.3f ms; %d bulk updates, sum %.3f ms" % (
    (big[0][0] - t0) / 1e6, len(big), sum(r[1] - r[0] for r in big) / 1e6))
gaps = [(b[0] - a[1]) / 1e3 for a, b in zip(big, big[1:])]
print("gaps between bulk updates: sum %.3f ms, max %.1f us, median %.1f us" % (
    sum(gaps) / 1e3, max(gaps), sorted(gaps)[len(gaps) // 2]))
print("after the last bulk update: %.3f ms" % ((t1 - big[-1][1]) / 1e6))
for i, r in enumerate(big):
    print("  bulk %2d start %.3f dur %7.1f us gap before %6.1f us" % (
        i, (r[0] - t0) / 1e6, (r[1] - r[0]) / 1e3, gaps[i - 1] if i else 0.0))
by = {}
for s, e_, n, st in run:
    key = (st, n.split("(")[0][:44])
    by.setdefault(key, [0, 0])
    by[key][0] += 1
    by[key][1] += e_ - s
for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  stream %-4s %-46s %4d %8.3f ms" % (k[0], k[1], v[0], v[1] / 1e6))
