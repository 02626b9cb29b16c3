"""Engine: one device context of libbqhip.so behind numpy-in / numpy-out calls.

This is the whole seam between the Python host code (``gp.py``, ``linalg.py``,
``bq.py``) and the HIP kernels.  Status codes are mapped to the exceptions the
reference raises at the same places (linalg_c.pyx:49-53,88-91): not positive
definite -> numpy.linalg.LinAlgError, bad argument -> ValueError, HIP failure ->
RuntimeError.
"""
import ctypes as C

import numpy as np

from . import _lib as L

_dp = L._dp


def _pts(x):
    """Points as d x n, column-major (gauss_c.pyx:116-117)."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x[None, :]
    if x.ndim != 2:
        raise ValueError("points must be a vector or a d x n matrix")
    return np.asfortranarray(x)


def _wvec(w, d):
    w = np.atleast_1d(np.asarray(w, dtype=np.float64)).ravel().copy()
    if w.shape[0] != d:
        raise ValueError("w has invalid shape")
    return w


class Engine(object):
    """A device context.  Not thread-safe; use one per thread / per GPU."""

    def __init__(self, device=0, stream=None):
        self._lib = L.load_library()
        self._ctx = C.c_void_p()
        n = C.c_int(0)
        self._lib.bq_device_count(C.byref(n))
        if n.value <= 0:
            raise RuntimeError("no HIP device visible: the MI355X engine cannot run "
                               "(there is no CPU fallback)")
        if device >= n.value:
            raise ValueError("device %d out of range (%d visible)" % (device, n.value))
        if stream is None:
            st = self._lib.bq_ctx_create(int(device), C.byref(self._ctx))
        else:
            st = self._lib.bq_ctx_create_on_stream(int(device), C.c_void_p(stream),
                                                   C.byref(self._ctx))
        if st != L.BQ_OK:
            raise RuntimeError("bq_ctx_create failed with status %d" % st)
        self.device = int(device)

    # -- plumbing ---------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.bq_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st == L.BQ_OK:
            return
        msg = self._lib.bq_last_error(self._ctx)
        msg = msg.decode() if msg else "status %d" % st
        if st == L.BQ_ERR_NOT_PD:
            raise np.linalg.LinAlgError(msg)
        if st == L.BQ_ERR_BAD_ARG:
            raise ValueError(msg)
        if st == L.BQ_ERR_NOMEM:
            raise MemoryError(msg)
        raise RuntimeError(msg)

    def sync(self):
        self._check(self._lib.bq_ctx_sync(self._ctx))

    def device_count(self):
        n = C.c_int(0)
        self._lib.bq_device_count(C.byref(n))
        return n.value

    def info(self):
        name = C.create_string_buffer(64)
        cus, clk, mem = C.c_int(), C.c_int(), C.c_size_t()
        self._check(self._lib.bq_device_info(self._ctx, name, C.byref(cus), C.byref(mem),
                                             C.byref(clk)))
        return {"name": name.value.decode(), "cus": cus.value, "hbm_bytes": mem.value,
                "clock_khz": clk.value}

    def set_block(self, nb):
        self._check(self._lib.bq_set_block(self._ctx, int(nb)))

    def set_guard(self, on):
        """Test aid: device buffers allocated from now on carry a 4 KiB sentinel band
        (Plan.check_guards reads them back).  Process-wide."""
        self._check(self._lib.bq_set_guard(1 if on else 0))

    def set_lookahead(self, on, min_rows=None):
        """Look-ahead on / off; min_rows: rows of the bulk update below which the sweep goes
        sequential (None keeps the current setting, 0 = look-ahead to the end)."""
        self._check(self._lib.bq_set_lookahead(self._ctx, 1 if on else 0))
        if min_rows is not None:
            self._check(self._lib.bq_set_lookahead_rows(self._ctx, int(min_rows)))

    def config(self):
        """(outer block override, look-ahead on, look-ahead min_rows) as they stand."""
        nb, la, mr = C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.bq_get_config(self._ctx, C.byref(nb), C.byref(la), C.byref(mr)))
        return nb.value, bool(la.value), mr.value

    def stats(self):
        """Counters of the context: ``flow_fallbacks`` = single-vector solves re-issued on the
        per-block sweeps after a hand-off of the one-launch sweeps timed out."""
        v = (C.c_int64 * 4)()
        self._check(self._lib.bq_ctx_stats(self._ctx, v, 4))
        return {"flow_fallbacks": int(v[0])}

    def trim(self):
        """Release the workspace the batched calls keep between calls."""
        self._check(self._lib.bq_ctx_trim(self._ctx))

    # -- raw device memory -------------------------------------------------

    def alloc(self, nbytes):
        p = C.c_void_p()
        self._check(self._lib.bq_dev_alloc(self._ctx, int(nbytes), C.byref(p)))
        return p

    def free(self, p):
        self._check(self._lib.bq_dev_free(self._ctx, p))

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr) if not arr.flags.f_contiguous else arr
        self._check(self._lib.bq_upload(self._ctx, dptr, arr.ctypes.data_as(C.c_void_p),
                                        arr.nbytes))

    def download(self, arr, dptr):
        self._check(self._lib.bq_download(self._ctx, arr.ctypes.data_as(C.c_void_p), dptr,
                                          arr.nbytes))

    def timer_start(self):
        self._check(self._lib.bq_timer_start(self._ctx))

    def timer_stop_ms(self):
        ms = C.c_float()
        self._check(self._lib.bq_timer_stop_ms(self._ctx, C.byref(ms)))
        return float(ms.value)

    def profile(self, on):
        self._check(self._lib.bq_profile_enable(self._ctx, 1 if on else 0))

    def profile_reset(self):
        self._check(self._lib.bq_profile_reset(self._ctx))

    def timeline(self, fn):
        """Runs fn() under the launch profiler and returns its launches as rows
        (class name, stream, start ms, end ms, work), times since the first launch."""
        self.profile(True)
        self.profile_reset()
        self._check(self._lib.bq_profile_timeline(self._ctx, 1, None, 0, None))
        fn()
        n = C.c_int64(0)
        self._check(self._lib.bq_profile_timeline(self._ctx, 0, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 5))
        self._check(self._lib.bq_profile_timeline(self._ctx, 0, L.dptr(out), n.value, C.byref(n)))
        self.profile(False)
        return [(L.K_CLASSES[int(r[0])], int(r[1]), r[2], r[3], r[4]) for r in out[:n.value]]

    def profile_read(self):
        ms = np.zeros(len(L.K_CLASSES))
        work = np.zeros(len(L.K_CLASSES))
        cnt = np.zeros(len(L.K_CLASSES), dtype=np.int64)
        self._check(self._lib.bq_profile_read(self._ctx, L.dptr(ms),
                                              cnt.ctypes.data_as(L._i64p), L.dptr(work)))
        return {k: {"ms": float(ms[i]), "launches": int(cnt[i]), "work": float(work[i])}
                for i, k in enumerate(L.K_CLASSES)}

    # -- linalg_c drop-ins (host arrays) ------------------------------------
    def cho_factor(self, Cm, Lm):
        """linalg_c.pyx:55-93 semantics; Cm, Lm are F-contiguous n x n."""
        n = Cm.shape[0]
        info = C.c_int64(0)
        self._check(self._lib.bq_cho_factor(self._ctx, L.dptr(Cm), L.dptr(Lm), n,
                                            C.byref(info)))

    def cho_solve(self, Lm, B, X, nrhs):
        self._check(self._lib.bq_cho_solve(self._ctx, L.dptr(Lm), L.dptr(B), L.dptr(X),
                                           Lm.shape[0], int(nrhs)))

    def logdet(self, Lm):
        out = C.c_double()
        self._check(self._lib.bq_logdet(self._ctx, L.dptr(Lm), Lm.shape[0],
                                        C.cast(C.byref(out), _dp)))
        return float(out.value)

    # -- Gram -----------------------------------------------------------------
    def gram(self, x, h, w, s=0.0):
        x = _pts(x)
        d, n = x.shape
        w = _wvec(w, d)
        K = np.empty((n, n), order="F")
        self._check(self._lib.bq_gram_gauss(self._ctx, L.dptr(x), d, n, float(h), L.dptr(w),
                                            float(s), L.dptr(K)))
        return K

    def gram_cross(self, x1, x2, h, w):
        x1, x2 = _pts(x1), _pts(x2)
        d = x1.shape[0]
        if x2.shape[0] != d:
            raise ValueError("dimension mismatch")
        w = _wvec(w, d)
        K = np.empty((x1.shape[1], x2.shape[1]), order="F")
        self._check(self._lib.bq_gram_gauss_cross(self._ctx, L.dptr(x1), x1.shape[1], L.dptr(x2),
                                                  x2.shape[1], d, float(h), L.dptr(w), L.dptr(K)))
        return K

    # -- closed-form integrals (gauss_c), host arrays -------------------------
    @staticmethod
    def _mc(d, mu, cov):
        mu = np.ascontiguousarray(np.atleast_1d(mu), dtype=np.float64)
        cov = np.asfortranarray(np.atleast_2d(cov), dtype=np.float64)
        if mu.shape != (d,):
            raise ValueError("mu has invalid shape")
        if cov.shape != (d, d):
            raise ValueError("cov has invalid shape")
        return mu, cov

    def int_K(self, x, h, w, mu, cov):
        x = _pts(x)
        d, n = x.shape
        w = _wvec(w, d)
        mu, cov = self._mc(d, mu, cov)
        out = np.empty(n)
        self._check(self._lib.bq_int_K(self._ctx, L.dptr(x), d, n, float(h), L.dptr(w), L.dptr(mu),
                                       L.dptr(cov), L.dptr(out)))
        return out

    def int_K1_K2(self, x1, x2, h1, w1, h2, w2, mu, cov):
        x1, x2 = _pts(x1), _pts(x2)
        d = x1.shape[0]
        if x2.shape[0] != d:
            raise ValueError("x2 has invalid shape")
        w1, w2 = _wvec(w1, d), _wvec(w2, d)
        mu, cov = self._mc(d, mu, cov)
        out = np.empty((x1.shape[1], x2.shape[1]), order="F")
        self._check(self._lib.bq_int_K1_K2(self._ctx, L.dptr(x1), x1.shape[1], L.dptr(x2),
                                           x2.shape[1], d, float(h1), L.dptr(w1), float(h2),
                                           L.dptr(w2), L.dptr(mu), L.dptr(cov), L.dptr(out)))
        return out

    def int_int_K1_K2_K1(self, x, h1, w1, h2, w2, mu, cov):
        x = _pts(x)
        d, n = x.shape
        w1, w2 = _wvec(w1, d), _wvec(w2, d)
        mu, cov = self._mc(d, mu, cov)
        out = np.empty((n, n), order="F")
        self._check(self._lib.bq_int_int_K1_K2_K1(self._ctx, L.dptr(x), d, n, float(h1), L.dptr(w1),
                                                  float(h2), L.dptr(w2), L.dptr(mu), L.dptr(cov),
                                                  L.dptr(out)))
        return out

    def int_int_K1_K2(self, x, h1, w1, h2, w2, mu, cov):
        x = _pts(x)
        d, n = x.shape
        w1, w2 = _wvec(w1, d), _wvec(w2, d)
        mu, cov = self._mc(d, mu, cov)
        out = np.empty(n)
        self._check(self._lib.bq_int_int_K1_K2(self._ctx, L.dptr(x), d, n, float(h1), L.dptr(w1),
                                               float(h2), L.dptr(w2), L.dptr(mu), L.dptr(cov),
                                               L.dptr(out)))
        return out

    # -- BQ moments on resident fits (bq_c) -------------------------------------
    def Z_mean(self, fit_l, mu, cov):
        mu, cov = self._mc(fit_l.d, mu, cov)
        out = C.c_double()
        self._check(self._lib.bq_bq_Z_mean(self._ctx, fit_l._handle(), L.dptr(mu), L.dptr(cov),
                                           C.cast(C.byref(out), _dp)))
        return float(out.value)

    def Z_var(self, fit_tl, fit_l, mu, cov):
        mu, cov = self._mc(fit_l.d, mu, cov)
        out = C.c_double()
        self._check(self._lib.bq_bq_Z_var(self._ctx, fit_tl._handle(), fit_l._handle(), L.dptr(mu), L.dptr(cov),
                                          C.cast(C.byref(out), _dp)))
        return float(out.value)

    def esm_batch(self, x_sc, l_sc, ns, x_a, h, w, thresh, mu, cov):
        """(A_a, A_sc_l, status) for every candidate of x_a (1-D); see bq_esm_batch."""
        x_sc = np.ascontiguousarray(x_sc, dtype=np.float64)
        l_sc = np.ascontiguousarray(l_sc, dtype=np.float64)
        x_a = np.ascontiguousarray(x_a, dtype=np.float64)
        mu, cov = self._mc(1, mu, cov)
        M = x_a.shape[0]
        A_a, A_sc_l = np.empty(M), np.empty(M)
        status = np.zeros(M, dtype=np.int32)
        self._check(self._lib.bq_esm_batch(
            self._ctx, L.dptr(x_sc), L.dptr(l_sc), int(ns), x_sc.shape[0], L.dptr(x_a), M,
            float(h), float(w), float(thresh), L.dptr(mu), L.dptr(cov), L.dptr(A_a),
            L.dptr(A_sc_l), status.ctypes.data_as(L._i32p)))
        return A_a, A_sc_l, status

    def esm_border(self, fit_l, ns, x_a, thresh, mu, cov):
        """(A_a, A_sc_l, status) for every candidate of x_a from the RESIDENT factor of
        gp_l (a Fit over (x_sc, l_sc), s = 0): one multi-right-hand-side solve instead of
        one factorisation per candidate; see bq_esm_border."""
        x_a = np.ascontiguousarray(x_a, dtype=np.float64)
        mu, cov = self._mc(1, mu, cov)
        M = x_a.shape[0]
        A_a, A_sc_l = np.empty(M), np.empty(M)
        status = np.zeros(M, dtype=np.int32)
        self._check(self._lib.bq_esm_border(
            self._ctx, fit_l._handle(), int(ns), L.dptr(x_a), M, float(thresh), L.dptr(mu),
            L.dptr(cov), L.dptr(A_a), L.dptr(A_sc_l), status.ctypes.data_as(L._i32p)))
        return A_a, A_sc_l, status

    # -- GP fits --------------------------------------------------------------
    def gp_fit(self, x, y, h, w, s=0.0):
        return Fit(self, x, y, h, w, s)

    def fit_predict(self, x, y, h, w, s, xo):
        """One bordered-Cholesky pass: (mean, var, logml)."""
        x, xo = _pts(x), _pts(xo)
        d, n = x.shape
        M = xo.shape[1]
        if xo.shape[0] != d:
            raise ValueError("dimension mismatch")
        w = _wvec(w, d)
        y = np.ascontiguousarray(y, dtype=np.float64)
        if y.shape != (n,):
            raise ValueError("y has invalid shape")
        mean, var = np.empty(M), np.empty(M)
        logml = C.c_double()
        self._check(self._lib.bq_fit_predict(self._ctx, L.dptr(x), L.dptr(y), d, n, float(h),
                                             L.dptr(w), float(s), L.dptr(xo), M, L.dptr(mean),
                                             L.dptr(var), C.cast(C.byref(logml), _dp)))
        return mean, var, float(logml.value)

    def logml_grid(self, x, y, h, w, s=0.0, chunk=0):
        """log-ML at G hyper-parameter points: h (G,), w (G,) or (G, d)."""
        x = _pts(x)
        d, n = x.shape
        y = np.ascontiguousarray(y, dtype=np.float64)
        h = np.ascontiguousarray(h, dtype=np.float64).ravel()
        G = h.shape[0]
        w = np.ascontiguousarray(np.asarray(w, dtype=np.float64).reshape(G, -1))
        if w.shape[1] == 1 and d > 1:
            w = np.ascontiguousarray(np.repeat(w, d, axis=1))
        if w.shape != (G, d):
            raise ValueError("w has invalid shape")
        out = np.empty(G)
        self._check(self._lib.bq_gp_logml_grid(self._ctx, L.dptr(x), L.dptr(y), d, n, L.dptr(h),
                                               L.dptr(w), float(s), G, L.dptr(out), int(chunk)))
        return out

    def batch_fit_predict(self, x, y, h, w, s, xo):
        """nprob independent problems.  x: (P, n) or (P, d, n); y: (P, n);
        xo: (P, M) or (P, d, M).  Returns mean (P, M), var (P, M), logml (P,),
        status (P,)."""
        x = np.asarray(x, dtype=np.float64)
        xo = np.asarray(xo, dtype=np.float64)
        if x.ndim == 2:
            x = x[:, None, :]
        if xo.ndim == 2:
            xo = xo[:, None, :]
        P, d, n = x.shape
        M = xo.shape[2]
        # each problem's points are d x n column-major = (n, d) C-order
        xb = np.ascontiguousarray(np.transpose(x, (0, 2, 1)))
        xob = np.ascontiguousarray(np.transpose(xo, (0, 2, 1)))
        y = np.ascontiguousarray(y, dtype=np.float64)
        if y.shape != (P, n) or xo.shape[0] != P or xo.shape[1] != d:
            raise ValueError("shape mismatch")
        w = _wvec(w, d)
        mean, var = np.empty((P, M)), np.empty((P, M))
        logml = np.empty(P)
        status = np.zeros(P, dtype=np.int32)
        self._check(self._lib.bq_batch_fit_predict(
            self._ctx, P, L.dptr(xb), L.dptr(y), d, n, float(h), L.dptr(w), float(s),
            L.dptr(xob), M, L.dptr(mean), L.dptr(var), L.dptr(logml),
            status.ctypes.data_as(L._i32p)))
        return mean, var, logml, status

    def plan(self, nprob, d, n, M):
        return Plan(self, nprob, d, n, M)

    def pair(self, x_s, tl_s, l_s, x_c, x_a, S):
        """The stacked pair of GPs of a BQ object, resident for S hyper-parameter sets
        (bq_pair): ``llh`` without acquisition points, ``esm`` with them."""
        return Pair(self, x_s, tl_s, l_s, x_c, x_a, S)

    # -- probes ----------------------------------------------------------------
    def probe_mfma_f64(self):
        v = C.c_double()
        self._check(self._lib.bq_probe_mfma_f64(self._ctx, C.cast(C.byref(v), _dp)))
        return float(v.value)

    def probe_mfma_variant(self, kind, nacc=8, waves_per_simd=2):
        """TFLOP/s of back-to-back independent MFMAs: kind 0 = v_mfma_f64_16x16x4_f64,
        1 = v_mfma_f64_4x4x4_4b_f64."""
        v = C.c_double()
        self._check(self._lib.bq_probe_mfma_variant(self._ctx, int(kind), int(nacc),
                                                    int(waves_per_simd), C.cast(C.byref(v), _dp)))
        return v.value

    def probe_fma_f64(self):
        v = C.c_double()
        self._check(self._lib.bq_probe_fma_f64(self._ctx, C.cast(C.byref(v), _dp)))
        return float(v.value)

    def probe_hbm(self, nbytes=1 << 30):
        a, b = C.c_double(), C.c_double()
        self._check(self._lib.bq_probe_hbm(self._ctx, int(nbytes), C.cast(C.byref(a), _dp),
                                           C.cast(C.byref(b), _dp)))
        return float(a.value), float(b.value)

    def probe_hbm_read8(self, nbytes=1 << 30, reps=4):
        """GB/s of a read-only pass with the single-vector sweeps' access pattern (8 B per
        lane); under ``rocprofv3 --pmc FETCH_SIZE`` a known byte count for that counter."""
        v = C.c_double()
        self._check(self._lib.bq_probe_hbm_read8(self._ctx, int(nbytes), int(reps),
                                                 C.cast(C.byref(v), _dp)))
        return float(v.value)

    def probe_gemm(self, m, n, k, lower=0, batch=1, qt=False, reps=20):
        """ms per launch of C (m x n) -= P Q^T (k columns) through the engine's kernel selection."""
        v = C.c_double()
        self._check(self._lib.bq_probe_gemm(self._ctx, int(m), int(n), int(k), int(lower),
                                            int(batch), 1 if qt else 0, int(reps),
                                            C.cast(C.byref(v), _dp)))
        return float(v.value)

    def probe_panel_solve(self, Lfac, X, mode=0, reps=0):
        """X L^-T for a batch of lower-triangular kb x kb factors L (batch, kb, kb) and row blocks
        X (batch, m, kb) through the batched factorisation's panel-solve launches."""
        L_ = np.asarray(Lfac, dtype=np.float64)
        X_ = np.asarray(X, dtype=np.float64)
        batch, kb, _ = L_.shape
        m = X_.shape[1]
        # column-major per problem
        Lf = np.ascontiguousarray(np.transpose(L_, (0, 2, 1)))
        Xf = np.ascontiguousarray(np.transpose(X_, (0, 2, 1)))
        ms = C.c_double()
        self._check(self._lib.bq_probe_panel_solve(self._ctx, int(m), int(kb), int(batch),
                                                   L.dptr(Lf), L.dptr(Xf), int(mode), int(reps),
                                                   C.cast(C.byref(ms), _dp)))
        if reps > 0:
            return float(ms.value)  # ms per call
        return np.transpose(Xf, (0, 2, 1)).copy()

    def probe_xcd_hop(self, mode, iters=2000, kib=1):
        """(ns per hand-off, XCC ids of the 16 workgroups, stale payload words) -- see
        bq_probe_xcd_hop in include/bqhip.h."""
        ns, bad = C.c_double(), C.c_int64()
        xcc = np.zeros(16, dtype=np.int32)
        self._check(self._lib.bq_probe_xcd_hop(self._ctx, int(mode), int(iters), int(kib),
                                               C.byref(ns),
                                               xcc.ctypes.data_as(C.POINTER(C.c_int32)),
                                               C.byref(bad)))
        return float(ns.value), xcc, int(bad.value)

    def probe_launch(self, n=2000):
        v = C.c_double()
        self._check(self._lib.bq_probe_launch(self._ctx, int(n), C.cast(C.byref(v), _dp)))
        return float(v.value)

    def probe_mfma_layout(self):
        out = np.empty(256)
        self._check(self._lib.bq_probe_mfma_layout(self._ctx, L.dptr(out)))
        return out.reshape(64, 4)


class Fit(object):
    """Device-resident GP fit (bq_fit): L, z = L^-1 y, log-ML; alpha on demand."""

    def __init__(self, eng, x, y, h, w, s):
        self._eng = eng
        self._h = C.c_void_p()
        x = _pts(x)
        self.d, self.n = x.shape
        w = _wvec(w, self.d)
        y = np.ascontiguousarray(y, dtype=np.float64)
        if y.shape != (self.n,):
            raise ValueError("y has invalid shape")
        eng._check(eng._lib.bq_gp_fit(eng._ctx, L.dptr(x), L.dptr(y), self.d, self.n, float(h),
                                      L.dptr(w), float(s), C.byref(self._h)))

    def close(self):
        if self._h is not None and self._h.value and self._eng._ctx.value:
            self._eng._lib.bq_fit_destroy(self._eng._ctx, self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _handle(self):
        if self._h is None or not self._h.value:
            raise ValueError("fit is closed")
        return self._h

    def refit(self, h, w, s):
        w = _wvec(w, self.d)
        e = self._eng
        e._check(e._lib.bq_gp_refit(e._ctx, self._handle(), float(h), L.dptr(w), float(s)))

    def set_y(self, y):
        """New targets for the same points; the fit must be refitted before its next use."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        if y.shape != (self.n,):
            raise ValueError("y has invalid shape")
        e = self._eng
        e._check(e._lib.bq_gp_set_y(e._ctx, self._handle(), L.dptr(y)))

    def refit_predict(self, h, w, s, xo):
        """New hyper-parameters and the posterior mean / marginal variance at xo in one sweep
        (bq_gp_refit_predict: the hyper-parameter loop's body)."""
        w = _wvec(w, self.d)
        xo = _pts(xo)
        if xo.shape[0] != self.d:
            raise ValueError("dimension mismatch")
        M = xo.shape[1]
        mean, var = np.empty(M), np.empty(M)
        e = self._eng
        e._check(e._lib.bq_gp_refit_predict(e._ctx, self._handle(), float(h), L.dptr(w), float(s),
                                            L.dptr(xo), M, L.dptr(mean), L.dptr(var)))
        return mean, var

    @property
    def logml(self):
        v = C.c_double()
        e = self._eng
        e._check(e._lib.bq_gp_logml(e._ctx, self._handle(), C.cast(C.byref(v), _dp)))
        return float(v.value)

    def _get(self, which, shape):
        out = np.empty(shape, order="F")
        e = self._eng
        e._check(e._lib.bq_gp_get(e._ctx, self._handle(), which, L.dptr(out)))
        return out

    def L(self):
        return self._get(0, (self.n, self.n))

    def alpha(self):
        return self._get(1, (self.n,))

    def z(self):
        return self._get(2, (self.n,))

    def K(self):
        return self._get(3, (self.n, self.n))

    def solve(self, B):
        """Kxx^-1 B with the resident factor; B is (n,) or (n, nrhs)."""
        B = np.asfortranarray(B, dtype=np.float64)
        nrhs = 1 if B.ndim == 1 else B.shape[1]
        if B.shape[0] != self.n:
            raise ValueError("b has invalid size")
        X = np.empty_like(B, order="F")
        e = self._eng
        e._check(e._lib.bq_gp_solve(e._ctx, self._handle(), L.dptr(B), nrhs, L.dptr(X)))
        return X

    def predict(self, xo, want_mean=True, want_var=True, want_cov=False):
        xo = _pts(xo)
        if xo.shape[0] != self.d:
            raise ValueError("dimension mismatch")
        M = xo.shape[1]
        mean = np.empty(M) if want_mean else None
        var = np.empty(M) if want_var else None
        cov = np.empty((M, M), order="F") if want_cov else None
        e = self._eng
        e._check(e._lib.bq_gp_predict(e._ctx, self._handle(), L.dptr(xo), M, L.dptr(mean), L.dptr(var),
                                      L.dptr(cov)))
        return mean, var, cov


class Pair(object):
    """GP1 over log l at the samples and GP2 over [l_s, exp(mean of GP1 at x_c)] at samples +
    candidates, evaluated at S hyper-parameter sets in one batched pass (bq_pair_*): the
    hyper-parameter objective of bq.py:536-550 and the acquisition under sampled
    hyper-parameters of bq.py:604-662."""

    def __init__(self, eng, x_s, tl_s, l_s, x_c, x_a, S):
        self._eng = eng
        self._h = C.c_void_p()
        def c(a):
            if a is None:
                return np.empty(0)
            return np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)

        x_s, tl_s, l_s, x_c, x_a = c(x_s), c(tl_s), c(l_s), c(x_c), c(x_a)
        self.ns, self.nc, self.ma, self.S = x_s.shape[0], x_c.shape[0], x_a.shape[0], int(S)
        if tl_s.shape != x_s.shape or l_s.shape != x_s.shape:
            raise ValueError("shape mismatch")
        eng._check(eng._lib.bq_pair_create(eng._ctx, L.dptr(x_s), L.dptr(tl_s), L.dptr(l_s),
                                           self.ns, L.dptr(x_c) if self.nc else None, self.nc,
                                           L.dptr(x_a) if self.ma else None, self.ma, self.S,
                                           C.byref(self._h)))

    def close(self):
        if self._h is not None and self._h.value and self._eng._ctx.value:
            self._eng._lib.bq_pair_destroy(self._eng._ctx, self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _params(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        if p.shape != (self.S, 3):
            raise ValueError("parameters must be S x 3: (h, w, s) per set")
        return p

    def llh(self, p_tl, p_l):
        """(llh[S], l_c[S, nc], status[S]); llh = -inf where status != 0."""
        p_tl, p_l = self._params(p_tl), self._params(p_l)
        llh = np.empty(self.S)
        l_c = np.empty((self.S, self.nc))
        status = np.zeros(self.S, dtype=np.int32)
        e = self._eng
        e._check(e._lib.bq_pair_llh(e._ctx, self._h, L.dptr(p_tl), L.dptr(p_l), L.dptr(llh),
                                    L.dptr(l_c) if self.nc else None,
                                    status.ctypes.data_as(L._i32p)))
        return llh, l_c, status

    def esm(self, p_tl, p_l, thresh, mu, cov):
        """The acquisition's ingredients for every set and acquisition point: dict with A_a,
        A_sc_l, status, tm_a, tC_a (all S x ma), l_c (S x nc), sstatus (S)."""
        p_tl, p_l = self._params(p_tl), self._params(p_l)
        mu, cov = Engine._mc(1, mu, cov)
        S, ma = self.S, self.ma
        out = {k: np.empty((S, ma)) for k in ("A_a", "A_sc_l", "tm_a", "tC_a")}
        out["status"] = np.zeros((S, ma), dtype=np.int32)
        out["l_c"] = np.empty((S, self.nc))
        out["sstatus"] = np.zeros(S, dtype=np.int32)
        e = self._eng
        e._check(e._lib.bq_pair_esm(
            e._ctx, self._h, L.dptr(p_tl), L.dptr(p_l), float(thresh), L.dptr(mu), L.dptr(cov),
            L.dptr(out["A_a"]), L.dptr(out["A_sc_l"]), out["status"].ctypes.data_as(L._i32p),
            L.dptr(out["tm_a"]), L.dptr(out["tC_a"]), L.dptr(out["l_c"]) if self.nc else None,
            out["sstatus"].ctypes.data_as(L._i32p)))
        return out


class Plan(object):
    """Resident batched fit+posterior pipeline (bq_plan); what bench.py times."""

    def __init__(self, eng, nprob, d, n, M):
        self._eng = eng
        self._h = C.c_void_p()
        self.nprob, self.d, self.n, self.M = int(nprob), int(d), int(n), int(M)
        eng._check(eng._lib.bq_plan_create(eng._ctx, self.nprob, self.d, self.n, self.M,
                                           C.byref(self._h)))

    def close(self):
        if self._h is not None and self._h.value and self._eng._ctx.value:
            self._eng._lib.bq_plan_destroy(self._eng._ctx, self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _handle(self):
        if self._h is None or not self._h.value:
            raise ValueError("plan is closed")
        return self._h

    def nbytes(self):
        v = C.c_size_t()
        self._eng._check(self._eng._lib.bq_plan_bytes(self._handle(), C.byref(v)))
        return int(v.value)

    def check_guards(self):
        """(buffers that carry a sentinel band, sentinel bytes overwritten) -- see
        Engine.set_guard; synchronises."""
        g, bad = C.c_int64(), C.c_int64()
        self._eng._check(self._eng._lib.bq_plan_check_guards(self._eng._ctx, self._handle(),
                                                             C.byref(g), C.byref(bad)))
        return int(g.value), int(bad.value)

    def set_inputs(self, x, y, xo, h, w, s):
        """x: (P, d, n) or (P, n); y: (P, n); xo: (P, d, M) or (P, M);
        h, s: (P,) or scalars; w: (P, d), (P,) or (d,)."""
        P, d, n, M = self.nprob, self.d, self.n, self.M
        x = np.asarray(x, dtype=np.float64).reshape(P, d, n)
        xb = np.ascontiguousarray(np.transpose(x, (0, 2, 1)))
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(P, n))
        if M:
            xo = np.asarray(xo, dtype=np.float64).reshape(P, d, M)
            xob = np.ascontiguousarray(np.transpose(xo, (0, 2, 1)))
        else:
            xob = None
        h = np.ascontiguousarray(np.broadcast_to(np.asarray(h, dtype=np.float64), (P,)))
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(s, dtype=np.float64), (P,)))
        w = np.asarray(w, dtype=np.float64)
        if w.ndim <= 1 and w.size == d:
            w = np.broadcast_to(w.reshape(1, d), (P, d))
        elif w.size == P and d == 1:
            w = w.reshape(P, 1)
        elif w.size == P and d > 1:
            w = np.repeat(w.reshape(P, 1), d, axis=1)
        w = np.ascontiguousarray(w.reshape(P, d))
        e = self._eng
        e._check(e._lib.bq_plan_set_inputs(e._ctx, self._handle(), L.dptr(xb), L.dptr(y), L.dptr(xob),
                                           L.dptr(h), L.dptr(w), L.dptr(s)))

    def run(self):
        e = self._eng
        e._check(e._lib.bq_plan_run(e._ctx, self._handle()))

    def results(self):
        P, M = self.nprob, self.M
        mean, var = np.empty((P, M)), np.empty((P, M))
        logml = np.empty(P)
        status = np.zeros(P, dtype=np.int32)
        e = self._eng
        e._check(e._lib.bq_plan_results(e._ctx, self._handle(), L.dptr(mean) if M else None,
                                        L.dptr(var) if M else None, L.dptr(logml),
                                        status.ctypes.data_as(L._i32p)))
        return mean, var, logml, status


_engines = {}


def get_engine(device=None):
    """Process-wide engine for a device (default: LOCAL_RANK or 0)."""
    import os
    if device is None:
        device = int(os.environ.get("BQ_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = C.c_int(0)
        L.load_library().bq_device_count(C.byref(n))
        if n.value > 0:
            device %= n.value
    if device not in _engines:
        _engines[device] = Engine(device)
    return _engines[device]


def set_engine(eng, device=0):
    """Install an engine object (tests install a CPU test double here)."""
    _engines[device] = eng
