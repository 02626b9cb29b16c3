"""Times the N=16384 trailing-update class in a sequential (no look-ahead) potrf, twice, and
prints TFLOP/s.  Used to A/B experimental builds: BQHIP_LIBRARY=<path to .so> python
tools/trailing_bench.py [label].  Results of diagnostic (ablated) builds are NOT checked."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine, _lib as L_  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402


def main():
    label = sys.argv[1] if len(sys.argv) > 1 else ""
    e = Engine(0)
    lib, ctx = e._lib, e._ctx
    if os.environ.get("TB_NB"):
        e.set_block(int(os.environ["TB_NB"]))
    n = 16384
    c4 = wl.c4(n)
    # TB_WSCALE=1e-6 makes the Gram matrix numerically diagonal: the trailing updates then
    # multiply exact zeros (same instructions, no operand toggling) -- a DVFS diagnostic
    w4 = np.ascontiguousarray(c4["w"]) * float(os.environ.get("TB_WSCALE", "1"))
    ld = n + int(os.environ.get("TB_LDPAD", "0"))
    xd, Kd, info = e.alloc(8 * n), e.alloc(8 * n * ld), e.alloc(64)
    e.upload(xd, np.ascontiguousarray(c4["x"]))
    res = []
    for la in (False, False, True, True):
        e.set_lookahead(la)
        e._check(lib.bq_gram_gauss_dev(ctx, xd, 1, n, c4["h"], L_.dptr(w4), c4["s"], Kd, ld))
        e.sync()
        if la:
            e.timer_start()
            e._check(lib.bq_potrf_dev(ctx, Kd, n, ld, info))
            res.append("potrf_la %.2f ms" % e.timer_stop_ms())
        else:
            e.profile(True)
            e.profile_reset()
            e._check(lib.bq_potrf_dev(ctx, Kd, n, ld, info))
            pr = e.profile_read()
            e.profile(False)
            sy = pr["syrk_trailing"]
            res.append("trailing %.2f TF (%.3f ms/launch)" % (
                sy["work"] / (sy["ms"] * 1e-3) / 1e12, sy["ms"] / max(1, sy["launches"])))
            if os.environ.get("TB_CLASSES"):
                res.append(" ".join("%s=%.2f/%d" % (k[:10], v["ms"], v["launches"])
                                    for k, v in pr.items() if v["launches"]))
    h = np.zeros(1, dtype=np.int32)
    e.download(h, info)
    print(label, "|", " | ".join(res), "| info", h[0])
    e.close()


if __name__ == "__main__":
    main()
