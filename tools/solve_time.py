"""The solve / predict rooflines of bench.py alone."""
import json
import sys

sys.path.insert(0, ".")
import bench  # noqa: E402
from bayesian_quadrature_amd import Engine  # noqa: E402

e = Engine(0)
r = bench.solve_predict_rooflines(e)
for k, v in r.items():
    print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()
              if kk in ("achieved", "unit", "frac", "ms_kernels", "ms_call_host_buffers", "class_ms")})
e.close()
