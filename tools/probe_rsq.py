import sys; sys.path.insert(0,'/root/repo')
import numpy as np, ctypes as C
from bayesian_quadrature_amd import Engine, _lib as L
e=Engine(0)
rs=np.random.RandomState(0)
x=np.exp(rs.uniform(-40,40,200000)); out=np.empty(3*x.size)
e._check(e._lib.bq_probe_rsq(e._ctx, L.dptr(x), x.size, L.dptr(out)))
o=out.reshape(-1,3); print("max rel err raw/1 newton/2 newton:", o.max(axis=0), "log2", np.log2(o.max(axis=0)+1e-300))
