"""Derives the lane maps of v_mfma_f64_4x4x4_4b_f64 from the hardware, with and without
the CBSZ / ABID A-operand broadcast."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine, _lib as L
e = Engine(0)


def table(cbsz, abid):
    out = np.empty(8192, dtype=np.int32)
    e._check(e._lib.bq_probe_mfma444_layout(e._ctx, cbsz, abid, out.ctypes.data_as(L._i32p)))
    o = out.astype(np.int64).reshape(64, 64, 2)
    return (o[..., 0] & 0xffffffff) | ((o[..., 1] & 0xffffffff) << 32)


for cbsz, abid in ((0, 0), (2, 0), (2, 1), (2, 3), (1, 0), (1, 1)):
    T = table(cbsz, abid)
    pairs = [(la, lb, [l for l in range(64) if (int(T[la, lb]) >> l) & 1]) for la in range(64)
             for lb in range(64) if T[la, lb]]
    print("cbsz", cbsz, "abid", abid, ": contributing (A lane, B lane) pairs:", len(pairs))
    # which A lanes are used at all, and an example of the D lanes per pair
    used = sorted(set(p[0] for p in pairs))
    print("   A lanes used:", used)
    for la, lb, dl in pairs[:6]:
        print("   A %2d x B %2d -> D %s" % (la, lb, dl))
