"""Derives the lane maps of v_mfma_f64_4x4x4_4b_f64 from the hardware."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine, _lib as L
e = Engine(0)
out = np.empty(4096, dtype=np.int32)
e._check(e._lib.bq_probe_mfma444_layout(e._ctx, out.ctypes.data_as(L._i32p)))
T = out.reshape(64, 64)   # T[la, lb] = D lane or -1
print("pairs that contribute:", int((T >= 0).sum()), "(expect 4 blocks x 4 k x 4 i x 4 j = 256)")
# group A lanes by which B lanes they pair with (same block and same k)
for la in range(64):
    lbs = np.nonzero(T[la] >= 0)[0]
    print("A lane %2d pairs with B lanes %s -> D lanes %s" % (la, lbs.tolist(), T[la, lbs].tolist()))
