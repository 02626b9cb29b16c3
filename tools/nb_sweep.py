"""Outer block of the factorisation for SMALL batches: ms of a resident plan pass (B systems of N
points, 64 prediction points) with the engine's choice and with blocks 64 / 128 / 256 forced --
the table behind auto_nb (potrf.hip).  python tools/nb_sweep.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine, workloads as wl
e = Engine(0)
rng = np.random.RandomState(3)
print("%6s %4s | %8s %8s %8s %8s" % ("N", "B", "auto", "64", "128", "256"))
for n in (512, 768, 1024, 1536, 2048, 3072):
    for B in (3, 5, 8, 12, 16, 24, 32):
        if 8.0 * n * n * B > 1.2e9: continue
        x = np.sort(rng.uniform(-5, 5, (B, n, 1)), axis=1)
        y = rng.randn(B, n)
        xo = rng.uniform(-5, 5, (B, 64, 1))
        res = []
        for nb in (0, 64, 128, 256):
            e.set_block(nb)
            plan = e.plan(B, 1, n, 64)
            plan.set_inputs(x, y, xo, 1.0, np.array([10.0 / n * 1.3]), 1e-3)
            plan.run(); e.sync()
            ts = []
            for rep in range(5):
                e.sync(); e.timer_start(); plan.run(); ts.append(e.timer_stop_ms())
            res.append(sorted(ts)[1])
            plan.close()
        best = min(res[1:])
        print("%6d %4d | %8.3f %8.3f %8.3f %8.3f %s" % (n, B, res[0], res[1], res[2], res[3], "<<<" if res[0] > 1.03 * best else ""), flush=True)
e.close()
