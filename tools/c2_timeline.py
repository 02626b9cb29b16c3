"""Where a one-launch slab step's time goes: in-kernel s_memtime stamps of workgroup 0
(the critical path) for every step of a C2 pass (bq_probe_c2_timeline)."""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from bayesian_quadrature_amd import Engine, _lib as L  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402

NAMES = ["entry", "frags_loaded", "rows_solved", "tile_loaded", "tile_updated", "potf2_entry",
         "potf2_loaded", "potf2_chain", "potf2_blocks", "potf2_end"]

if __name__ == "__main__":
    e = Engine(0)
    c = wl.c2()
    plan = e.plan(1, 1, 1024, 256)
    plan.set_inputs(c["x"][None], c["y"][None], c["xo"][None], c["h"], c["w"], c["s"])
    for _ in range(5):
        plan.run()
    e.sync()
    nsteps = 16
    st = (C.c_int64 * (160 * nsteps))()
    e._check(e._lib.bq_probe_c2_timeline(e._ctx, plan._h, st, nsteps))
    t = np.array(list(st), dtype=np.int64).reshape(nsteps, 160)[:, :10]
    d = np.diff(t, axis=1)                       # phase lengths per step
    gap = t[1:, 0] - t[:-1, 9]                   # end of a step's factor -> next step's entry
    out = {"library": L.LIB_PATH, "unit": "s_memtime ticks",
           "phases_mean_steps_1_14": {NAMES[i + 1]: float(d[1:15, i].mean()) for i in range(9)},
           "phases_step_5": {NAMES[i + 1]: int(d[5, i]) for i in range(9)},
           "step_total_mean": float((t[1:15, 9] - t[1:15, 0]).mean()),
           "gap_end_to_next_entry_mean": float(gap[:14].mean()),
           "gap_all": gap.tolist(),
           "pass_total": int(t[15, 4] - t[0, 0])}
    print(json.dumps(out, indent=1))
    plan.close()
    e.close()
