"""us per launch and phase cycles of the eight-wave factor for several builds (BQHIP_LIBRARY is read
at import: one child process per library).  python tools/potf2_variants.py lib1.so lib2.so ..."""
import json
import os
import subprocess
import sys

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from bayesian_quadrature_amd import Engine
import potf2_probe as pp
e = Engine(0)
rs = np.random.RandomState(0)
R = rs.rand(64, 64); S = R + R.T + 64 * np.eye(64)
for fl in (2, 3):
    Lo, dv, info, us, st = pp.probe(e, S, fl, reps=400)
    print("  fl=%d us=%.3f phases=%s total=%d" % (fl, us, (st[1:5]-st[:4]).tolist(), st[4]-st[0]))
e.close()
'''
for lib in sys.argv[1:]:
    env = dict(os.environ, BQHIP_LIBRARY=os.path.abspath(lib))
    print(lib, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
