"""Socket power and shader clock while the batched panel solve runs for seconds (tools/power_probe.py's
sampler): python tools/panel_solve_power.py [MODE ...]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402

e = Engine(0)


def smi():
    import bench
    v = bench._hwmon_sample()
    return (int(v[1]), v[0]) if v else (-1, -1.0)


modes = [int(v) for v in sys.argv[1:]] or [2, 18]
m, kb, batch = (int(v) for v in os.environ.get("SHAPE", "2048,448,64").split(","))
rs = np.random.RandomState(0)
# unit diagonal + a small strictly lower part: thousands of solves of the solve's own output stay
# finite, of order one and with random mantissas (a well-conditioned random factor divides by
# ~30 per solve and the operands are zeros after 200 repetitions -- a chip that multiplies zeros
# draws far less power and holds 2.39 GHz)
Lf = np.eye(kb) + 1e-4 * np.tril(rs.standard_normal((kb, kb)), -1)
Ls = np.repeat(Lf[None], batch, 0)
Xs = rs.standard_normal((batch, m, kb))
print("shape", m, kb, batch, "idle", smi(), flush=True)
for mode in modes:
    res = {}
    t = threading.Thread(target=lambda: res.setdefault("v", e.probe_panel_solve(Ls, Xs, mode, reps=int(os.environ.get("REPS", str(int(2.2e12 / (m * kb * batch))))))))
    t.start()
    time.sleep(0.5)
    seen = []
    while t.is_alive():
        seen.append(smi())
        time.sleep(0.2)
    t.join()
    seen = seen[:-1] or seen
    clk = sum(s[0] for s in seen) / max(len(seen), 1)
    pw = sum(s[1] for s in seen) / max(len(seen), 1)
    r = m * kb * kb * batch / res["v"] / 1e9
    print("mode %d: %.3f ms %.1f TFLOP/s  sclk %4.0f MHz  %5.0f W (%d samples) -> %.1f%% of the MFMA rate at "
          "that clock" % (mode, res["v"], r, clk, pw, len(seen), 100.0 * r / (78.6 * clk / 2400.0)), flush=True)
e.close()
