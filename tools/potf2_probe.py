"""Times and checks the 64 x 64 diagonal factor alone (bq_probe_potf2): microseconds per
launch (HIP events), the in-kernel s_memtime phases, and the factor / block inverses /
reciprocal pivots / info against numpy.  BQHIP_LIBRARY selects an experimental build."""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from bayesian_quadrature_amd import Engine, _lib as L  # noqa: E402


def probe(e, A, from_lds, reps=200):
    A = np.asfortranarray(A, dtype=np.float64)
    Lo = np.zeros((64, 64), order="F")
    dv = np.zeros(64 + 4 * 256)
    info = C.c_int32(0)
    us = C.c_double(0)
    st = (C.c_int64 * 136)()
    e._check(e._lib.bq_probe_potf2(e._ctx, L.dptr(A), int(from_lds), reps, L.dptr(Lo), L.dptr(dv),
                                   C.byref(info), C.cast(C.byref(us), L._dp), st))
    return Lo, dv, info.value, us.value, np.array(list(st), dtype=np.int64)


def check(A, Lo, dv):
    Lr = np.linalg.cholesky(A)
    errL = np.max(np.abs(np.tril(Lo) - Lr)) / np.max(np.abs(Lr))
    errd = np.max(np.abs(dv[:64] * np.diag(Lr) - 1.0))
    errW = 0.0
    for b in range(4):
        W = dv[64 + 256 * b:64 + 256 * (b + 1)].reshape(16, 16, order="F")
        Lb = Lr[16 * b:16 * b + 16, 16 * b:16 * b + 16]
        errW = max(errW, np.max(np.abs(np.tril(W).dot(Lb) - np.eye(16))))
    return errL, errd, errW


if __name__ == "__main__":
    e = Engine(0)
    rs = np.random.RandomState(0)
    out = {"library": L.LIB_PATH}
    # a C2-like diagonal block (Gaussian kernel, w = dx, s = 1e-3) and a random SPD block
    x = np.linspace(-5, 5, 1024)[:64]
    dx = 10.0 / 1023
    K = np.exp(-0.5 * (x[:, None] - x[None]) ** 2 / dx ** 2) / (np.sqrt(2 * np.pi) * dx)
    K += 1e-6 * np.eye(64)
    R = rs.rand(64, 64)
    S = R + R.T + 64 * np.eye(64)
    for name, A in (("gauss", K), ("random_spd", S)):
        # bit 0: the block through LDS; bit 1: the eight-wave form; bit 2: per-wave barrier
        # stamps as well (four waves only; they cost the chain ~2,500 cycles)
        for fl in (0, 1, 2, 3, 5):
            Lo, dv, info, us, st = probe(e, A, fl)
            errL, errd, errW = check(A, Lo, dv)
            ph = (st[1:5] - st[:4]).tolist()
            w = st[8:136].reshape(16, 4, 2) - st[1]   # [panel, wave, arrive/release]
            out["%s_from_lds%d" % (name, fl)] = {
                "us_per_launch": us, "info": info, "errL": errL, "err_dinv": errd, "err_W": errW,
                "stamp_ticks_load_chain_blocks_tail": ph,
                "cycles_in_kernel_total": int(st[4] - st[0]),
                "barrier_arrive_by_wave": w[:, :, 0].tolist() if fl & 4 else None,
                "barrier_release_by_wave": w[:, :, 1].tolist() if fl & 4 else None}
            assert info == 0 and errL < 1e-13 and errd < 1e-13 and errW < 1e-12, out
    # failure report: first non-positive pivot at column 37 (1-based 38)
    B = S.copy()
    Lr = np.linalg.cholesky(S)
    B[37, 37] = Lr[37, :37].dot(Lr[37, :37]) - 1e-3
    for fl in (1, 3):
        Lo, dv, info, us, st = probe(e, B, fl, reps=3)
        out["not_pd_info_%d" % fl] = info
        assert info == 38, info
        assert np.max(np.abs(np.tril(Lo)[:, :37] - Lr[:, :37])) < 1e-12
    print(json.dumps(out, indent=1))
    e.close()
