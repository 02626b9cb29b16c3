import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine, _lib as L_
from bayesian_quadrature_amd import workloads as wl
e = Engine(0)
lib, ctx = e._lib, e._ctx
c3 = wl.c3(); pts = np.asfortranarray(c3["x"]); w3 = np.ascontiguousarray(c3["w"][200])
xd = e.alloc(8*2*4096); Kd = e.alloc(8*4096*4096); e.upload(xd, pts)
res = []
for rep in range(5):
    for _ in range(3):
        e._check(lib.bq_gram_gauss_dev(ctx, xd, 2, 4096, float(c3["h"][200]), L_.dptr(w3), c3["s"], Kd, 4096))
    e.sync(); e.timer_start()
    for _ in range(50):
        e._check(lib.bq_gram_gauss_dev(ctx, xd, 2, 4096, float(c3["h"][200]), L_.dptr(w3), c3["s"], Kd, 4096))
    res.append(e.timer_stop_ms()/50*1e3)
print("us per launch", [round(r,2) for r in res], "GB/s", round(134.28e6/min(res)/1e3,1))
