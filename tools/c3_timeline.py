"""The C3 grid (400 points, N = 4096, chunks of 100) through Engine.logml_grid: wall time of seven
passes; with an argument also the launch profiler's totals per stream and class, with two the
launches of the first chunk.  python tools/c3_timeline.py [q | a b]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine, workloads as wl
e = Engine(0)
if os.environ.get("C3_NB"):
    e.set_block(int(os.environ["C3_NB"]))
c3 = wl.c3()
ts = []
for rep in range(7):
    t0 = time.perf_counter()
    lm = e.logml_grid(c3["x"], c3["y"], c3["h"], c3["w"], c3["s"], chunk=100)
    ts.append((time.perf_counter() - t0) * 1e3)
print("c3 grid ms:", " ".join("%.1f" % t for t in ts), flush=True)
rows = e.timeline(lambda: e.logml_grid(c3["x"], c3["y"], c3["h"], c3["w"], c3["s"], chunk=100))
agg = {}
for cls, st, t0, t1, w in rows:
    a = agg.setdefault((st, cls), [0, 0.0, 0.0]); a[0] += 1; a[1] += t1 - t0; a[2] += w
if len(sys.argv) == 2 and sys.argv[1] == "q":
    sys.exit(0)
print("span %.1f" % max(r[3] for r in rows))
for k in sorted(agg):
    n, t, w = agg[k]
    print("%d %-20s n=%4d %8.2f ms %6.1f TF" % (k[0], k[1], n, t, w / max(t, 1e-9) / 1e9))
if len(sys.argv) > 2:
    for cls, st, t0, t1, w in rows:
        if t0 < 50.0:
            print("%d %-20s %8.3f %8.1f us %6.1f TF" % (st, cls, t0, (t1 - t0) * 1e3, w / max(t1 - t0, 1e-9) / 1e9))
e.close()
