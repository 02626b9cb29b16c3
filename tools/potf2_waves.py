"""Arrival of the eight waves at the sixteen panel barriers of the 64 x 64 factor (bq_probe_potf2,
flag 6 | from_lds): cycles after the chain's start, and which wave came last.
python tools/potf2_waves.py [lib.so ...]"""
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
Lo, dv, info, us, st = pp.probe(e, S, 7, reps=50)
a = st[8:136].reshape(16, 8) - st[1]
print("  us=%.2f total=%d" % (us, st[4] - st[0]))
prev = 0
for P in range(16):
    last = int(np.argmax(a[P]))
    print("  barrier %2d: last wave %d at %6d (+%4d)  arrivals %s" % (P, last, a[P].max(), a[P].max() - prev, (a[P] - a[P].max()).tolist()))
    prev = a[P].max()
e.close()
'''
for lib in (sys.argv[1:] or [None]):
    env = dict(os.environ)
    if lib:
        env["BQHIP_LIBRARY"] = os.path.abspath(lib)
    print(lib or "default", flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
