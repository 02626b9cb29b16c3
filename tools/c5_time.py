"""Times one C5 shard (64 problems, N=2048, M=256) through a resident plan, several runs; also
256 copies of C2 and one C3 chunk (100 grid points) when asked: python tools/c5_time.py [all]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

e = Engine(0)
if os.environ.get("TB_NB"):
    e.set_block(int(os.environ["TB_NB"]))
c5 = wl.c5(range(64))
plan = e.plan(64, 1, 2048, 256)
plan.set_inputs(c5["x"], c5["y"], c5["xo"], c5["h"], c5["w"], c5["s"])
ts = []
for rep in range(8):
    e.sync()
    e.timer_start()
    plan.run()
    ts.append(e.timer_stop_ms())
print("c5 shard ms:", " ".join("%.3f" % t for t in ts), flush=True)
res = plan.results()
print("failed", int((res[3] != 0).sum()), "logml[0] %.12e" % res[2][0])
plan.close()
if len(sys.argv) > 1:
    c2 = wl.c2()
    B = 256
    plan = e.plan(B, 1, 1024, 256)
    plan.set_inputs(np.repeat(c2["x"][None], B, 0), np.repeat(c2["y"][None], B, 0),
                    np.repeat(c2["xo"][None], B, 0), c2["h"], c2["w"], c2["s"])
    ts = []
    for rep in range(6):
        e.sync()
        e.timer_start()
        plan.run()
        ts.append(e.timer_stop_ms())
    print("256 x c2 ms:", " ".join("%.3f" % t for t in ts), flush=True)
    plan.close()
    c3 = wl.c3()
    for rep in range(3):
        t0 = time.perf_counter()
        lm = e.logml_grid(c3["x"], c3["y"], c3["h"], c3["w"], c3["s"], chunk=100)
        print("c3 grid %.1f ms" % ((time.perf_counter() - t0) * 1e3), "n_inf", int(np.isinf(lm).sum()),
              "lm[0] %.12e" % lm[0], flush=True)
e.close()
