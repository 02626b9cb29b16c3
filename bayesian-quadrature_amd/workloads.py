"""Synthetic inputs of the BASELINE.json configurations.

All fp64, seeded, Gaussian integrands.  Used by bench.py and by the parity
tests so that both run exactly the same problems.
"""
import numpy as np

LOG_2PI = float(np.log(2.0 * np.pi))


def norm_logpdf(x, mu=0.0, sd=1.0):
    z = (np.asarray(x, dtype=np.float64) - mu) / sd
    return -0.5 * z * z - np.log(sd) - 0.5 * LOG_2PI


def c2(n=1024, m=256):
    """C2: d=1, N=1024 on linspace(-5,5), y = log N(x|0,1), h=1, w=dx, s=1e-3,
    M=256 prediction points offset by dx/3."""
    x = np.linspace(-5.0, 5.0, n)
    dx = 10.0 / (n - 1)
    y = norm_logpdf(x)
    xo = np.linspace(-5.0, 5.0, m) + dx / 3.0
    return {"x": x, "y": y, "xo": xo, "h": 1.0, "w": np.array([dx]), "s": 1e-3, "d": 1}


def c3(side=64, seed=3, gh=20, gw=20):
    """C3: d=2, N=side^2 jittered grid on [-5,5]^2, y = log N(x|0,I); hyper grid
    h in logspace(-1,1,gh) x w in linspace(0.5 dx, 2 dx, gw) (isotropic), s=0.1."""
    rs = np.random.RandomState(seed)
    g = np.linspace(-5.0, 5.0, side)
    dx = 10.0 / (side - 1)
    X, Y = np.meshgrid(g, g, indexing="ij")
    pts = np.stack([X.ravel(), Y.ravel()], axis=0)  # 2 x N
    pts = pts + rs.uniform(-dx / 4, dx / 4, size=pts.shape)
    y = norm_logpdf(pts[0]) + norm_logpdf(pts[1])
    hs = np.logspace(-1, 1, gh)
    ws = np.linspace(0.5 * dx, 2.0 * dx, gw)
    H, W = np.meshgrid(hs, ws, indexing="ij")
    return {"x": pts, "y": y, "h": H.ravel(), "w": np.repeat(W.ravel()[:, None], 2, axis=1),
            "s": 0.1, "d": 2, "dx": dx}


def c4(n=16384):
    """C4: d=1, N=16384 on linspace, w=dx, s=1e-3 (the MFMA roofline run)."""
    x = np.linspace(-5.0, 5.0, n)
    dx = 10.0 / (n - 1)
    return {"x": x, "h": 1.0, "w": np.array([dx]), "s": 1e-3, "d": 1}


def c4_dense(n=16384):
    """C4's matrix size with DENSE operands: the same points, a length scale of 200 spacings and
    unit noise (well conditioned through the noise floor).  C4 itself (w = dx) gives a banded Gram
    -- exp underflows to exactly 0 beyond ~38 neighbours -- so the panels fed to its trailing
    updates are > 97 % zeros, which draw less power and clock higher: the kernel's roofline figure
    is quoted on this variant."""
    x = np.linspace(-5.0, 5.0, n)
    dx = 10.0 / (n - 1)
    return {"x": x, "h": 1.0, "w": np.array([200.0 * dx]), "s": 1.0, "d": 1}


def c5_problem(p, n=2048, m=256):
    """Problem p of C5: x = sort(U(-5,5,n)) seed 1000+p, y = log N(x|mu_p,1)."""
    rs = np.random.RandomState(1000 + p)
    x = np.sort(rs.uniform(-5.0, 5.0, n))
    mu = rs.uniform(-1.0, 1.0)
    y = norm_logpdf(x, mu, 1.0)
    xo = np.linspace(-5.0, 5.0, m)
    return x, y, xo


def c5(problems, n=2048, m=256):
    """C5 shard: the listed problem indices; w = mean spacing, s=1e-2."""
    xs, ys, xos = zip(*[c5_problem(p, n, m) for p in problems])
    dx = 10.0 / (n - 1)
    return {"x": np.stack(xs), "y": np.stack(ys), "xo": np.stack(xos), "h": 1.0,
            "w": np.array([dx]), "s": 1e-2, "d": 1}


def shard(nitems, rank, world):
    """Contiguous block partition of range(nitems) over ranks (no exchange)."""
    base, rem = divmod(nitems, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return list(range(lo, hi))


def potrf_flops(n):
    return n ** 3 / 3.0


def trailing_flops(n, nb):
    """sum_k m_k^2 nb with m_k = n - (k+1) nb: the flops the trailing MFMA
    update actually performs on full square tiles' lower halves, counted as
    the algorithmic m^2 nb."""
    t, k = 0.0, 0
    while (k + 1) * nb < n:
        m = n - (k + 1) * nb
        t += float(m) * m * nb
        k += 1
    return t
