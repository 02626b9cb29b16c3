"""BQ.fit_hypers(['h', 'w']) with the central-difference gradient from one batched device pass per gradient
(bq_pair_llh) against scipy's own sequential differencing of the same objective, on the
reference's fixture size and at ns = 1024; and one evaluation of the objective."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bayesian_quadrature_amd as bqa  # noqa: E402
from bayesian_quadrature_amd import util, workloads as wl  # noqa: E402

params = ["h", "w"]
for n in (9, 64, 1024):
    def make():
        np.random.seed(8728)
        x = np.linspace(-5, 5, n)
        b = bqa.BQ(x, np.exp(wl.norm_logpdf(x)), n_candidate=10, x_mean=0.0, x_var=10.0,
                   candidate_thresh=0.5, kernel=bqa.GaussianKernel, optim_method="L-BFGS-B")
        dx = 10.0 / (n - 1)
        if n == 9:
            b.init(params_tl=(15.0, 2.0, 0.0), params_l=(0.2, 1.3, 0.0))
        else:
            b.init(params_tl=(15.0, 1.3 * dx, 1e-3), params_l=(0.2, 1.3 * dx, 1e-4))
        return b
    b = make()
    f = b._make_llh_params(params)
    p0 = b._current_params(params)
    f(p0)
    t0 = time.perf_counter()
    for it in range(20):
        f(p0 * (1.0 + 1e-4 * (it + 1)))
    t_eval = (time.perf_counter() - t0) / 20 * 1e3
    fb = b._make_llh_batch(params)
    fb(util.cd_points(p0)[0])
    t0 = time.perf_counter()
    for it in range(20):
        fb(util.cd_points(p0 * (1.0 + 1e-4 * (it + 1)))[0])
    t_grad = (time.perf_counter() - t0) / 20 * 1e3
    res = {}
    for mode in ("sequential", "batched"):
        b = make()
        f = b._make_llh_params(params)
        t0 = time.perf_counter()
        p = util.find_good_parameters(f, b._current_params(params), "L-BFGS-B",
                                      logpdf_batch=b._make_llh_batch(params) if mode == "batched" else None)
        res[mode] = (time.perf_counter() - t0, f(p), dict(util.LAST_OPT))
    print("n=%d (nc=%d): objective %.3f ms, value + central gradient (9 points) in one pass %.3f ms; "
          "fit_hypers sequential %.1f ms (llh %.9f, %d iterations, %d evaluations), batched "
          "gradient %.1f ms (llh %.9f, %d iterations, %d passes): %.2fx wall, %.3f / %.3f ms per "
          "iteration"
          % (n, b.nc, t_eval, t_grad, res["sequential"][0] * 1e3, res["sequential"][1],
             res["sequential"][2]["nit"], res["sequential"][2]["nfev"],
             res["batched"][0] * 1e3, res["batched"][1], res["batched"][2]["nit"],
             res["batched"][2]["nfev"], res["sequential"][0] / res["batched"][0],
             res["sequential"][0] * 1e3 / max(1, res["sequential"][2]["nit"]),
             res["batched"][0] * 1e3 / max(1, res["batched"][2]["nit"])),
          flush=True)
