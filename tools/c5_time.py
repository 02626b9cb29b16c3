"""Times one C5 shard (64 problems, N=2048, M=256) through a resident plan, several runs."""
import os
import sys

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
for rep in range(4):
    e.sync()
    e.timer_start()
    plan.run()
    print("rep", rep, "%.3f ms" % e.timer_stop_ms(), flush=True)
res = plan.results()
print("failed", int((res[3] != 0).sum()))
plan.close()
e.close()
