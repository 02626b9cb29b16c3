// linalg.hip -- the linalg_c drop-ins (cho_factor, cho_solve, logdet: linalg_c.pyx:55-210) on
// host buffers, the Gram entry points and the device-resident Cholesky.
#include "host.h"
#include <vector>

using namespace bqh;

// ===========================================================================
// linalg_c drop-ins (host buffers)
// ===========================================================================
// upload an n x n host matrix (ld n) into a padded ntot x ntot device matrix
static int upload_padded(bq_ctx *c, const double *H, int n, DevBuf &A, int &ntot, long &lda)
{
    ntot = (int)roundup(n, 64);
    lda = pick_ld(ntot);
    HIPCHK(c, A.alloc(sizeof(double) * (size_t)lda * ntot));
    HIPCHK(c, hipMemcpy2DAsync(A.p, sizeof(double) * lda, H, sizeof(double) * n,
                               sizeof(double) * n, n, hipMemcpyHostToDevice, c->stream));
    if (ntot > n) {
        BQCHK(launch_pad_identity(c, A.d(), lda, n, ntot));
    }
    return BQ_OK;
}

extern "C" int bq_cho_factor(bq_ctx *c, const double *C, double *L, int64_t n, int64_t *info_out)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (info_out)
        *info_out = 0;
    if (!C || !L || n < 0)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (n == 0)
        return BQ_OK;
    if (n > 65536)
        return fail(c, BQ_ERR_BAD_ARG, "n too large");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->solve_kcopy && n <= 64) {
        // one 64 x 64 block: the diagonal factor's kernel straight on the mapped staging buffer
        // ([flag | the block with its identity padding]), one launch, one synchronisation
        double *hs = nullptr, *ds = nullptr;
        BQCHK(ctx_stage(c, 1 + 4096, &hs, &ds));
        if (c->panel_ws.bytes < sizeof(double) * panel_ws_doubles(64, 1))
            HIPCHK(c, c->panel_ws.alloc(sizeof(double) * panel_ws_doubles(64, 1)));
        hs[0] = 0.0; // (the flag is its first four bytes)
        double *blk = hs + 1;
        std::memset(blk, 0, sizeof(double) * 4096);
        for (int64_t j = 0; j < 64; ++j) {
            if (j < n)
                std::memcpy(blk + 64 * j, C + j * n, sizeof(double) * (size_t)n);
            else
                blk[65 * j] = 1.0;
        }
        BQCHK(launch_potf2(c, ds + 1, 64, 0, 0, c->panel_ws.d(), 0, reinterpret_cast<int *>(ds), 1));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        int hinfo = 0;
        std::memcpy(&hinfo, hs, sizeof hinfo);
        if (hinfo != 0) {
            if (info_out)
                *info_out = hinfo;
            return fail(c, BQ_ERR_NOT_PD, "matrix is not positive definite");
        }
        for (int64_t j = 0; j < n; ++j) {
            if (C != L && j > 0)
                std::memcpy(L + j * n, C + j * n, sizeof(double) * (size_t)j);
            std::memcpy(L + j + j * n, blk + j + 64 * j, sizeof(double) * (size_t)(n - j));
        }
        return BQ_OK;
    }
    DevBuf A, ws;
    int ntot;
    long lda;
    // a small matrix (the reference's own sizes): in and out through the mapped staging buffer,
    // one kernel each way and ONE synchronisation
    const bool small = c->solve_kcopy && (size_t)n * n + 1 <= (32u << 10);
    double *hs = nullptr, *ds = nullptr;
    HIPCHK(c, ws.alloc(BQ_DINV_STRIDE * sizeof(double) + 64));
    double *dinv = ws.d();
    int *info = reinterpret_cast<int *>(ws.d() + BQ_DINV_STRIDE);
    if (small) {
        BQCHK(ctx_stage(c, (size_t)n * n + 1, &hs, &ds));
        ntot = (int)roundup(n, 64);
        lda = pick_ld(ntot);
        HIPCHK(c, A.alloc(sizeof(double) * (size_t)lda * ntot));
        std::memcpy(hs + 1, C, sizeof(double) * (size_t)n * n);
        BQCHK(launch_mat_in(c, ds + 1, (int)n, A.d(), lda, ntot, info));
    } else {
        BQCHK(upload_padded(c, C, (int)n, A, ntot, lda));
        HIPCHK(c, hipMemsetAsync(info, 0, sizeof(int), c->stream));
    }
    if (c->panel_ws.bytes < sizeof(double) * panel_ws_doubles(ntot, 1))
        HIPCHK(c, c->panel_ws.alloc(sizeof(double) * panel_ws_doubles(ntot, 1)));
    BQCHK(enqueue_potrf_partial(c, A.d(), lda, 0, 1, ntot, ntot, dinv, info, c->panel_ws.d(),
                                c->panel_ws.bytes / sizeof(double)));
    int hinfo = 0;
    if (small) {
        BQCHK(launch_mat_out(c, ds, A.d(), lda, (int)n, info));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hinfo = (int)hs[0];
    } else {
        HIPCHK(c, hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (hinfo != 0) {
        if (info_out)
            *info_out = hinfo;
        return fail(c, BQ_ERR_NOT_PD, "matrix is not positive definite");
    }
    if (small) {
        // the lower triangle out of the staging buffer; the strict upper part of L keeps what the
        // caller had there (C's values after the reference's C -> L copy)
        const double *src = hs + 1;
        for (int64_t j = 0; j < n; ++j) {
            if (C != L && j > 0)
                std::memcpy(L + j * n, C + j * n, sizeof(double) * (size_t)j);
            std::memcpy(L + j + j * n, src + j + j * n, sizeof(double) * (size_t)(n - j));
        }
        return BQ_OK;
    }
    // copy back only the lower triangle; the strict upper part of L keeps what
    // the caller had there (C's values after the reference's C -> L copy)
    if (C != L) {
        // straight into L, then the caller's strict upper triangle over what the device left there
        HIPCHK(c, hipMemcpy2D(L, sizeof(double) * n, A.p, sizeof(double) * lda, sizeof(double) * n,
                              n, hipMemcpyDeviceToHost));
        for (int64_t j = 1; j < n; ++j)
            std::memcpy(L + j * n, C + j * n, sizeof(double) * (size_t)j);
        return BQ_OK;
    }
    std::vector<double> tmp((size_t)n * n);
    HIPCHK(c, hipMemcpy2D(tmp.data(), sizeof(double) * n, A.p, sizeof(double) * lda,
                          sizeof(double) * n, n, hipMemcpyDeviceToHost));
    for (int64_t j = 0; j < n; ++j)
        std::memcpy(L + j + j * n, tmp.data() + j + j * n, sizeof(double) * (size_t)(n - j));
    return BQ_OK;
}

extern "C" int bq_cho_solve(bq_ctx *c, const double *L, const double *B, double *X, int64_t n,
                            int64_t nrhs)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!L || !B || !X || n < 0 || nrhs < 0)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (n == 0 || nrhs == 0)
        return BQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->solve_kcopy && n <= 64 && nrhs <= 64) {
        // the reference's own sizes: one launch on the mapped staging buffer [X | L | B]
        double *hs = nullptr, *ds = nullptr;
        const size_t nx = (size_t)n * nrhs;
        BQCHK(ctx_stage(c, 2 * nx + (size_t)n * n, &hs, &ds));
        std::memcpy(hs + nx, L, sizeof(double) * (size_t)n * n);
        std::memcpy(hs + nx + (size_t)n * n, B, sizeof(double) * nx);
        BQCHK(launch_small_potrs(c, ds, (int)n, (int)nrhs));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::memcpy(X, hs, sizeof(double) * nx);
        return BQ_OK;
    }
    DevBuf A, ws, Xd;
    int npad;
    long ldl;
    // one vector against a small factor (the reference's own sizes): factor and vector in, solution
    // out through the mapped staging buffer -- [x n | L n^2 | b n] -- by kernels
    const bool small = c->solve_kcopy && nrhs == 1 && (size_t)n * n + 2 * (size_t)n <= (32u << 10);
    double *hs = nullptr, *ds = nullptr;
    if (small) {
        BQCHK(ctx_stage(c, (size_t)n * n + 2 * (size_t)n, &hs, &ds));
        npad = (int)roundup(n, 64);
        ldl = pick_ld(npad);
        HIPCHK(c, A.alloc(sizeof(double) * (size_t)ldl * npad));
        std::memcpy(hs + n, L, sizeof(double) * (size_t)n * n);
        std::memcpy(hs + n + (size_t)n * n, B, sizeof(double) * (size_t)n);
        BQCHK(launch_mat_in(c, ds + n, (int)n, A.d(), ldl, npad, nullptr));
    } else {
        // the strict upper triangle of L is never read by the sweeps
        BQCHK(upload_padded(c, L, (int)n, A, npad, ldl));
    }
    // the block inverses of the factor's diagonal: 16 x 16 (panel solve), then B wide
    DevBuf wide, X2;
    HIPCHK(c, ws.alloc(sizeof(double) * BQ_DINV_HALF * (size_t)(npad / 64)));
    HIPCHK(c, wide.alloc(sizeof(double) * wide_alloc_doubles(npad)));
    BQCHK(launch_diag_winv(c, A.d(), ldl, npad, ws.d()));
    BQCHK(compute_wide_inverses(c, A.d(), ldl, npad, ws.d(), wide.d()));
    const WideInv w = wide_views(wide.d(), npad);
    if (nrhs == 1) {
        // one right-hand side: the GEMV sweeps (trsv.h)
        HIPCHK(c, Xd.alloc(sizeof(double) * (2 * (size_t)npad + trsv_flow_ws_doubles(npad, w.B))));
        double *x = Xd.d(), *y = Xd.d() + npad, *fw = Xd.d() + 2 * (size_t)npad;
        // (X may alias B, linalg_c.pyx:128: the right-hand side is staged before any attempt
        // writes X; a timed-out hand-off re-issues the solve on the per-block kernels)
        std::vector<double> b0;
        if (!small && X == B)
            b0.assign(B, B + n);
        const double *bsrc = b0.empty() ? B : b0.data();
        BQCHK(with_flow_fallback(c, [&]() -> int {
            HIPCHK(c, hipMemsetAsync(Xd.p, 0, sizeof(double) * 2 * (size_t)npad, c->stream));
            if (small)
                BQCHK(launch_flow_in(c, ds + n + (size_t)n * n, (int)n, x, npad, nullptr, 0));
            else
                HIPCHK(c, hipMemcpyAsync(Xd.p, bsrc, sizeof(double) * n, hipMemcpyHostToDevice,
                                         c->stream));
            BQCHK(enqueue_forward_vec(c, x, y, A.d(), ldl, npad, w, fw));
            BQCHK(enqueue_backward_vec(c, y, x, A.d(), ldl, npad, w, fw));
            if (small)
                BQCHK(launch_flow_out(c, x, (int)n, ds));
            else
                HIPCHK(c, hipMemcpyAsync(X, x, sizeof(double) * n, hipMemcpyDeviceToHost,
                                         c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            return BQ_OK;
        }));
        if (small)
            std::memcpy(X, hs, sizeof(double) * (size_t)n);
        return BQ_OK;
    }
    return solve_rows_host(c, A.d(), ldl, (int)n, npad, w, B, nrhs, X);
}

extern "C" int bq_logdet(bq_ctx *c, const double *L, int64_t n, double *out)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!L || !out || n < 0)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    HIPCHK(c, hipSetDevice(c->device));
    if (n == 0) {
        *out = 0.0;
        return BQ_OK;
    }
    // only the diagonal travels
    std::vector<double> diag((size_t)n);
    for (int64_t i = 0; i < n; ++i)
        diag[(size_t)i] = L[i + i * n];
    if (c->solve_kcopy && n + 1 <= (32 << 10)) {
        // (the diagonal and the result through the mapped staging buffer: one launch, no
        // allocation, no copy operation)
        double *hs = nullptr, *ds = nullptr;
        BQCHK(ctx_stage(c, (size_t)n + 1, &hs, &ds));
        std::memcpy(hs + 1, diag.data(), sizeof(double) * (size_t)n);
        BQCHK(launch_logdet(c, ds + 1, 0L, (int)n, ds));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        *out = hs[0];
        return BQ_OK;
    }
    DevBuf dv;
    HIPCHK(c, dv.alloc(sizeof(double) * (n + 1)));
    HIPCHK(c, hipMemcpyAsync(dv.p, diag.data(), sizeof(double) * n, hipMemcpyHostToDevice,
                             c->stream));
    BQCHK(launch_logdet(c, dv.d(), 0L, (int)n, dv.d() + n));
    HIPCHK(c, hipMemcpyAsync(out, dv.d() + n, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

// ===========================================================================
// Gram
// ===========================================================================
extern "C" int bq_gram_gauss_dev(bq_ctx *c, const double *x_dev, int64_t d, int64_t n, double h,
                                 const double *w, double s, double *K_dev, int64_t ldk)
{
    BQCHK(check_dims(c, d, n));
    BQCHK(check_w(c, d, h, w, s));
    if (!x_dev || !K_dev || ldk < n)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    // parameters ride in a tiny device buffer so the same kernel serves the
    // batched callers; uploaded only when they change
    GaussParams g = make_params((int)d, h, w, s);
    if (!c->gbuf.p) {
        HIPCHK(c, c->gbuf.alloc(sizeof(GaussParams)));
        c->gbuf_valid = false;
    }
    if (!c->gbuf_valid || std::memcmp(&c->gbuf_host, &g, sizeof g) != 0) {
        // synchronous on purpose: the staging copy of `g` must not outlive this frame
        HIPCHK(c, hipMemcpyAsync(c->gbuf.p, &g, sizeof g, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->gbuf_host = g;
        c->gbuf_valid = true;
    }
    return launch_gram_sym(c, (int)d, x_dev, 0, static_cast<GaussParams *>(c->gbuf.p), 0, K_dev,
                           ldk, 0, (int)n, 1);
}

extern "C" int bq_gram_gauss(bq_ctx *c, const double *x, int64_t d, int64_t n, double h,
                             const double *w, double s, double *K_out)
{
    BQCHK(check_dims(c, d, n));
    BQCHK(check_w(c, d, h, w, s));
    if (!x || !K_out)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf xd, Kd, gd;
    HIPCHK(c, xd.alloc(sizeof(double) * d * n));
    HIPCHK(c, Kd.alloc(sizeof(double) * (size_t)n * n));
    HIPCHK(c, gd.alloc(sizeof(GaussParams)));
    GaussParams g = make_params((int)d, h, w, s);
    HIPCHK(c, hipMemcpyAsync(xd.p, x, sizeof(double) * d * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(gd.p, &g, sizeof g, hipMemcpyHostToDevice, c->stream));
    BQCHK(launch_gram_sym(c, (int)d, xd.d(), 0, static_cast<GaussParams *>(gd.p), 0, Kd.d(), n, 0,
                          (int)n, 1));
    HIPCHK(c, hipMemcpyAsync(K_out, Kd.p, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

extern "C" int bq_gram_gauss_cross(bq_ctx *c, const double *x1, int64_t n1, const double *x2,
                                   int64_t n2, int64_t d, double h, const double *w, double *K_out)
{
    BQCHK(check_dims(c, d, n1 > 0 ? n1 : 1));
    BQCHK(check_w(c, d, h, w, 0.0));
    if (n1 < 0 || n2 < 0 || (n1 && !x1) || (n2 && !x2) || (!K_out && n1 * n2))
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (n1 == 0 || n2 == 0)
        return BQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf a, b, Kd;
    HIPCHK(c, a.alloc(sizeof(double) * d * n1));
    HIPCHK(c, b.alloc(sizeof(double) * d * n2));
    HIPCHK(c, Kd.alloc(sizeof(double) * (size_t)n1 * n2));
    HIPCHK(c, hipMemcpyAsync(a.p, x1, sizeof(double) * d * n1, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b.p, x2, sizeof(double) * d * n2, hipMemcpyHostToDevice, c->stream));
    GaussParams g = make_params((int)d, h, w, 0.0);
    BQCHK(launch_gram_cross(c, (int)d, a.d(), (int)n1, b.d(), (int)n2, g, Kd.d(), n1));
    HIPCHK(c, hipMemcpyAsync(K_out, Kd.p, sizeof(double) * (size_t)n1 * n2, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

// ===========================================================================
// device-resident Cholesky
// ===========================================================================
extern "C" int bq_potrf_dev(bq_ctx *c, double *A_dev, int64_t n, int64_t lda, int32_t *info_dev)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!A_dev || !info_dev || n <= 0 || (n & 63) || lda < n || (lda & 1))
        return fail(c, BQ_ERR_BAD_ARG, "potrf_dev: n must be a positive multiple of 64, lda even");
    if (!c->dinv64.p)
        HIPCHK(c, c->dinv64.alloc(BQ_DINV_STRIDE * sizeof(double)));
    HIPCHK(c, hipMemsetAsync(info_dev, 0, sizeof(int32_t), c->stream));
    if (c->panel_ws.bytes < sizeof(double) * panel_ws_doubles((int)n, 1))
        HIPCHK(c, c->panel_ws.alloc(sizeof(double) * panel_ws_doubles((int)n, 1)));
    return enqueue_potrf_partial(c, A_dev, lda, 0, 1, (int)n, (int)n, c->dinv64.d(), info_dev,
                                 c->panel_ws.d(), c->panel_ws.bytes / sizeof(double));
}
