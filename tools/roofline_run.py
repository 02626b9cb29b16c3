"""A short, fixed sequence of the two north-star kernels for profiler passes:
  * gram_sym_kernel<2> on the C3 points (N=4096, d=2), 10 launches
  * gram_sym_kernel<1> at N=16384, 3 launches
  * one sequential (no look-ahead) potrf at N=16384, nb=256 -> 63 trailing updates
  * 20 passes of the C2 problem (the headline workload's slab_step_kernel)
  * a resident N=16384 fit and 4 single-vector solves (32 + 32 trsv step launches each)
Used under `rocprofv3 --kernel-trace --stats` and under `rocprofv3 --pmc ...`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine, _lib as L_  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402


def main():
    e = Engine(0)
    # 20 passes of the headline problem (C2: 16 slab_step_kernel launches each)
    c2 = wl.c2()
    plan = e.plan(1, 1, 1024, 256)
    plan.set_inputs(c2["x"][None], c2["y"][None], c2["xo"][None], c2["h"], c2["w"], c2["s"])
    for _ in range(20):
        plan.run()
    e.sync()
    plan.close()
    e.set_lookahead(False)
    e.set_block(256)
    lib, ctx = e._lib, e._ctx
    c3 = wl.c3()
    pts = np.asfortranarray(c3["x"])
    w3 = np.ascontiguousarray(c3["w"][200])
    xd = e.alloc(8 * 2 * 4096)
    Kd = e.alloc(8 * 4096 * 4096)
    e.upload(xd, pts)
    for _ in range(10):
        e._check(lib.bq_gram_gauss_dev(ctx, xd, 2, 4096, float(c3["h"][200]), L_.dptr(w3), c3["s"], Kd, 4096))
    e.sync()
    e.free(xd), e.free(Kd)
    n = 16384
    c4 = wl.c4(n)
    w4 = np.ascontiguousarray(c4["w"])
    xd = e.alloc(8 * n)
    Kd = e.alloc(8 * n * n)
    info = e.alloc(64)
    e.upload(xd, np.ascontiguousarray(c4["x"]))
    for _ in range(3):
        e._check(lib.bq_gram_gauss_dev(ctx, xd, 1, n, c4["h"], L_.dptr(w4), c4["s"], Kd, n))
    e._check(lib.bq_potrf_dev(ctx, Kd, n, n, info))
    e.sync()
    h = np.zeros(1, dtype=np.int32)
    e.download(h, info)
    print("potrf info", h[0])
    e.free(xd), e.free(Kd), e.free(info)
    # the GEMV sweeps of one right-hand side over a resident N=16384 factor
    e.set_lookahead(True)
    e.set_block(0)
    fit = e.gp_fit(c4["x"], wl.norm_logpdf(c4["x"]), c4["h"], c4["w"], c4["s"])
    b = np.random.RandomState(5).randn(n)
    for _ in range(4):
        fit.solve(b)
    e.sync()
    fit.close()
    e.close()


if __name__ == "__main__":
    main()
