"""The reference's 9-point test fixture (bayesian_quadrature/tests/util.py:12-59
and bq.py:132-171,967-991) rebuilt from its recipe: data, not reference code."""
import json
import os

import numpy as np
from scipy.stats import norm

HERE = os.path.dirname(os.path.abspath(__file__))


def known_answers():
    with open(os.path.join(HERE, "golden", "known_answers.json")) as f:
        return json.load(f)


def build_chain(fit, predict_mean, filter_candidates):
    """fit(x, y, h, w, s) -> (L, alpha, logml); predict_mean(x, h, w, L, alpha, xo);
    filter_candidates(xc, xs, thresh) in place.  Returns the dict of arrays."""
    fx = known_answers()["fixture"]
    np.random.seed(fx["seed"])
    xs = np.linspace(fx["xmin"], fx["xmax"], fx["n"])
    ls = norm.pdf(xs, 0, 1)
    h1, w1, s1 = fx["params_tl"]
    h2, w2, s2 = fx["params_l"]
    xc = np.random.uniform(xs.min() - w1, xs.max() + w1, fx["n_candidate"])
    filter_candidates(xc, xs, fx["candidate_thresh"])
    xc = np.sort(xc[~np.isnan(xc)])
    L1, a1, lm1 = fit(xs, np.log(ls), h1, w1, s1)
    lc = np.exp(predict_mean(xs, h1, w1, L1, a1, xc))
    xsc = np.concatenate([xs, xc])
    lsc = np.concatenate([ls, lc])
    L2, a2, lm2 = fit(xsc, lsc, h2, w2, s2)
    return dict(xs=xs, ls=ls, xc=xc, lc=lc, xsc=xsc, lsc=lsc, L1=L1, a1=a1, L2=L2, a2=a2,
                lm1=lm1, lm2=lm2, h1=h1, w1=w1, h2=h2, w2=w2, mu=np.array([fx["x_mean"]]),
                cov=np.array([[fx["x_var"]]]))
