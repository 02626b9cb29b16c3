"""Times the batched BASELINE configs on one GPU: C5 shard (64 x N=2048, M=256) and
C3 (20x20 hyper-grid log-ML at N=4096, d=2).  Prints JSON."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402


def main():
    e = Engine(0)
    if os.environ.get('BQ_NB'):
        e.set_block(int(os.environ['BQ_NB']))
    out = {'nb': os.environ.get('BQ_NB', 'auto')}
    # ---- C5 shard -------------------------------------------------------------
    B = int(os.environ.get("C5_BATCH", "64"))
    c = wl.c5(list(range(B)))
    plan = e.plan(B, 1, 2048, 256)
    plan.set_inputs(c["x"], c["y"], c["xo"], c["h"], c["w"], c["s"])
    plan.run(); e.sync()
    e.timer_start()
    reps = 3
    for _ in range(reps):
        plan.run()
    ms = e.timer_stop_ms() / reps
    mean, var, logml, status = plan.results()
    e.profile(True); e.profile_reset(); plan.run(); prof = e.profile_read(); e.profile(False)
    out["c5_shard"] = {"batch": B, "ms_per_batch": ms, "problems_per_s": B / ms * 1e3,
                       "failed": int((status != 0).sum()), "class_ms": {k: v["ms"] for k, v in prof.items()}, "class_tflops_or_gbs": {k: (v["work"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0) for k, v in prof.items()},
                       "class_launches": {k: v["launches"] for k, v in prof.items()},
                       "plan_GB": plan.nbytes() / 1e9}
    plan.close()
    # ---- C3 grid ----------------------------------------------------------------
    c3 = wl.c3()
    t0 = time.perf_counter()
    lm = e.logml_grid(c3["x"], c3["y"], c3["h"], c3["w"], c3["s"], chunk=int(os.environ.get("C3_CHUNK", "100")))
    t1 = time.perf_counter()
    out["c3_grid"] = {"points": len(lm), "wall_ms": (t1 - t0) * 1e3, "ms_per_point": (t1 - t0) * 1e3 / len(lm),
                      "n_minus_inf": int(np.isinf(lm).sum()), "logml_min": float(np.min(lm[np.isfinite(lm)])),
                      "logml_max": float(np.max(lm))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
