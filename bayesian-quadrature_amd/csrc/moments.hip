// moments.hip -- host side of the closed-form Gaussian-kernel integrals, of the BQ moments and
// of the acquisition entry points, with their kernels (moments.h).
//
// Reference: gauss_c.pyx:95-164,235-339,416-531,617-713 and bq_c.pyx:157-213,264-355.
// The d x d (or 2d x 2d) covariance algebra is done here on the host in plain
// loops; the per-point / per-pair work runs in the kernels of moments.h.
#include "host.h"
#include "moments.h"

using namespace bqh;

namespace {

constexpr int MAXF = 2 * BQ_MAXD;
constexpr double LOG_2PI = 1.8378770664093453;

// row-major small dense helpers, n <= MAXF
struct Small {
    int n = 0;
    double a[MAXF * MAXF] = {0};
    double &operator()(int r, int c) { return a[r * n + c]; }
    double operator()(int r, int c) const { return a[r * n + c]; }
};

Small small_from_colmajor(const double *m, int n)
{
    Small s;
    s.n = n;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c)
            s(r, c) = m[r + c * n];
    return s;
}

Small small_mul(const Small &x, const Small &y)
{
    Small z;
    z.n = x.n;
    for (int r = 0; r < x.n; ++r)
        for (int c = 0; c < x.n; ++c) {
            double t = 0.0;
            for (int k = 0; k < x.n; ++k)
                t += x(r, k) * y(k, c);
            z(r, c) = t;
        }
    return z;
}

// lower Cholesky factor in place; false if not positive definite
bool small_chol(Small &m)
{
    const int n = m.n;
    for (int j = 0; j < n; ++j) {
        double d = m(j, j);
        for (int k = 0; k < j; ++k)
            d -= m(j, k) * m(j, k);
        if (!(d > 0.0))
            return false;
        d = std::sqrt(d);
        m(j, j) = d;
        for (int i = j + 1; i < n; ++i) {
            double t = m(i, j);
            for (int k = 0; k < j; ++k)
                t -= m(i, k) * m(j, k);
            m(i, j) = t / d;
        }
        for (int c = j + 1; c < n; ++c)
            m(j, c) = 0.0;
    }
    return true;
}

Small small_inv_lower(const Small &L)
{
    const int n = L.n;
    Small X;
    X.n = n;
    for (int c = 0; c < n; ++c) {
        X(c, c) = 1.0 / L(c, c);
        for (int r = c + 1; r < n; ++r) {
            double t = 0.0;
            for (int k = c; k < r; ++k)
                t += L(r, k) * X(k, c);
            X(r, c) = -t / L(r, r);
        }
    }
    return X;
}

// symmetric positive definite inverse via the Cholesky factor
bool small_spd_inv(const Small &C, Small &out)
{
    Small L = C;
    if (!small_chol(L))
        return false;
    const Small Li = small_inv_lower(L);
    out.n = C.n;
    for (int r = 0; r < C.n; ++r)
        for (int c = 0; c < C.n; ++c) {
            double t = 0.0;
            for (int k = 0; k < C.n; ++k)
                t += Li(k, r) * Li(k, c);
            out(r, c) = t;
        }
    return true;
}

template <int D>
bool fill_form(GaussForm<D> &f, const double *mu, const Small &C)
{
    Small L = C;
    if (!small_chol(L))
        return false;
    const Small Li = small_inv_lower(L);
    double logdet = 0.0;
    for (int r = 0; r < D; ++r) {
        f.mu[r] = mu ? mu[r] : 0.0;
        logdet += 2.0 * std::log(L(r, r));
        for (int c = 0; c < D; ++c)
            f.linv[r * D + c] = Li(r, c);
    }
    f.logc = -0.5 * (D * LOG_2PI + logdet);
    return true;
}

Small cov_plus_w2(const Small &cov, const double *w, double cscale)
{
    Small m;
    m.n = cov.n;
    for (int r = 0; r < cov.n; ++r)
        for (int c = 0; c < cov.n; ++c)
            m(r, c) = cscale * cov(r, c) + (r == c ? w[r] * w[r] : 0.0);
    return m;
}

int check_int_args(bq_ctx *c, int64_t d, const double *w, const double *mu, const double *cov)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (d < 1 || d > BQ_MAXD)
        return fail(c, BQ_ERR_BAD_ARG, "d must be in 1..%d", BQ_MAXD);
    if (!w || !mu || !cov)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    return BQ_OK;
}

#define BQ_DISPATCH_D(d_, ...)                                                                     \
    switch (d_) {                                                                                  \
    case 1: { constexpr int D = 1; __VA_ARGS__; } break;                                           \
    case 2: { constexpr int D = 2; __VA_ARGS__; } break;                                           \
    case 3: { constexpr int D = 3; __VA_ARGS__; } break;                                           \
    case 4: { constexpr int D = 4; __VA_ARGS__; } break;                                           \
    case 5: { constexpr int D = 5; __VA_ARGS__; } break;                                           \
    case 6: { constexpr int D = 6; __VA_ARGS__; } break;                                           \
    case 7: { constexpr int D = 7; __VA_ARGS__; } break;                                           \
    default: { constexpr int D = 8; __VA_ARGS__; } break;                                          \
    }

// ---- device-side pieces shared by the host-buffer entry points and the moments ----

// out_dev[i] = scale exp(add + log N(x_i | mu, C)); with alpha_dev: *sum_dev = sum out alpha
int dev_int_K(bq_ctx *c, int d, const double *x_dev, int n, const double *mu, const Small &C,
              double scale, double add, double *out_dev, const double *alpha_dev, double *part_dev,
              double *sum_dev)
{
    const int nb = (n + 255) / 256;
    bool ok = true;
    BQ_DISPATCH_D(d, {
        GaussForm<D> f;
        ok = fill_form<D>(f, mu, C);
        if (ok)
            hipLaunchKernelGGL(int_K_kernel<D>, dim3(nb), dim3(256), 0, c->stream, x_dev, n, f,
                               scale, add, out_dev, alpha_dev, alpha_dev ? part_dev : nullptr);
    });
    if (!ok)
        return fail(c, BQ_ERR_NOT_PD, "matrix is not positive definite");
    HIPCHK(c, hipGetLastError());
    if (alpha_dev) {
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, c->stream, part_dev,
                           (const double *)nullptr, nb, sum_dev);
        HIPCHK(c, hipGetLastError());
    }
    return BQ_OK;
}

// C2 = [[cov + W1, cov], [cov, cov + W2]] with mean [mu, mu]
int dev_int_K1_K2(bq_ctx *c, int d, const double *x1_dev, int n1, const double *x2_dev, int n2,
                  const double *w1, const double *w2, const double *mu, const Small &cov,
                  double scale, double *out_dev, const double *alpha_dev, double *beta_dev)
{
    Small C2;
    C2.n = 2 * d;
    double mu2[MAXF];
    for (int r = 0; r < d; ++r) {
        mu2[r] = mu2[r + d] = mu[r];
        for (int q = 0; q < d; ++q) {
            C2(r, q) = cov(r, q) + (r == q ? w1[r] * w1[r] : 0.0);
            C2(r + d, q + d) = cov(r, q) + (r == q ? w2[r] * w2[r] : 0.0);
            C2(r, q + d) = cov(r, q);
            C2(r + d, q) = cov(r, q);
        }
    }
    bool ok = true;
    BQ_DISPATCH_D(d, {
        GaussForm<2 * D> f;
        ok = fill_form<2 * D>(f, mu2, C2);
        if (ok)
            hipLaunchKernelGGL(int_K1_K2_kernel<D>, dim3((n1 + 63) / 64), dim3(256), 0, c->stream,
                               x1_dev, n1, x2_dev, n2, f, scale, out_dev, alpha_dev, beta_dev);
    });
    if (!ok)
        return fail(c, BQ_ERR_NOT_PD, "matrix is not positive definite");
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// int int K1 K2 K1: prepares b = G x, n1 = log N(x | mu, W1 + S) in work (d*n + n doubles)
// and either stores the n x n matrix or reduces alpha' I alpha into *sum_dev.
int dev_iikk(bq_ctx *c, int d, const double *x_dev, int n, double h1, const double *w1, double h2,
             const double *w2, const double *mu, const Small &cov, double *work_dev,
             double *out_dev, const double *alpha_dev, double *part_dev, double *sum_dev)
{
    const Small W1S = cov_plus_w2(cov, w1, 1.0);
    Small W1Sinv;
    if (!small_spd_inv(W1S, W1Sinv))
        return fail(c, BQ_ERR_NOT_PD, "matrix is not positive definite");
    const Small G = small_mul(cov, W1Sinv); // S (W1 + S)^-1
    const Small A = small_mul(G, cov);
    Small Cm;
    Cm.n = d;
    for (int r = 0; r < d; ++r)
        for (int q = 0; q < d; ++q)
            Cm(r, q) = (r == q ? w2[r] * w2[r] : 0.0) + 2.0 * cov(r, q) - 2.0 * A(r, q);
    double *b_dev = work_dev, *n1_dev = work_dev + (size_t)d * n;
    const double scale = (h1 * h1) * (h1 * h1) * (h2 * h2);
    const int gx = (n + 63) / 64, gy = (n + 255) / 256;
    bool ok = true;
    BQ_DISPATCH_D(d, {
        GaussForm<D> f1, g, fc;
        ok = fill_form<D>(f1, mu, W1S) && fill_form<D>(fc, nullptr, Cm);
        if (ok) {
            for (int r = 0; r < D; ++r)
                for (int q = 0; q < D; ++q)
                    g.linv[r * D + q] = G(r, q);
            hipLaunchKernelGGL(iikk_prepare_kernel<D>, dim3((n + 255) / 256), dim3(256), 0,
                               c->stream, x_dev, n, f1, g, b_dev, n1_dev);
            hipLaunchKernelGGL(int_int_K1_K2_K1_kernel<D>, dim3(gx, gy), dim3(256), 0, c->stream,
                               b_dev, n1_dev, n, fc, scale, out_dev, alpha_dev,
                               alpha_dev ? part_dev : nullptr);
        }
    });
    if (!ok)
        return fail(c, BQ_ERR_NOT_PD, "matrix is not positive definite");
    HIPCHK(c, hipGetLastError());
    if (alpha_dev) {
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, c->stream, part_dev,
                           (const double *)nullptr, gx * gy, sum_dev);
        HIPCHK(c, hipGetLastError());
    }
    return BQ_OK;
}

} // namespace

// ===========================================================================
// gauss_c drop-ins: host buffers in and out
// ===========================================================================
extern "C" int bq_int_K(bq_ctx *c, const double *x, int64_t d, int64_t n, double h, const double *w,
                        const double *mu, const double *cov, double *out)
{
    BQCHK(check_int_args(c, d, w, mu, cov));
    if (n < 0 || (n && (!x || !out)))
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (n == 0)
        return BQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf xd, od;
    HIPCHK(c, xd.alloc(sizeof(double) * d * n));
    HIPCHK(c, od.alloc(sizeof(double) * n));
    HIPCHK(c, hipMemcpyAsync(xd.p, x, sizeof(double) * d * n, hipMemcpyHostToDevice, c->stream));
    const Small S = small_from_colmajor(cov, (int)d);
    BQCHK(dev_int_K(c, (int)d, xd.d(), (int)n, mu, cov_plus_w2(S, w, 1.0), h * h, 0.0, od.d(),
                    nullptr, nullptr, nullptr));
    HIPCHK(c, hipMemcpyAsync(out, od.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

extern "C" int bq_int_K1_K2(bq_ctx *c, const double *x1, int64_t n1, const double *x2, int64_t n2,
                            int64_t d, double h1, const double *w1, double h2, const double *w2,
                            const double *mu, const double *cov, double *out)
{
    BQCHK(check_int_args(c, d, w1, mu, cov));
    if (!w2 || n1 < 0 || n2 < 0 || (n1 && !x1) || (n2 && !x2) || (n1 * n2 && !out))
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (n1 == 0 || n2 == 0)
        return BQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf a, b, od;
    HIPCHK(c, a.alloc(sizeof(double) * d * n1));
    HIPCHK(c, b.alloc(sizeof(double) * d * n2));
    HIPCHK(c, od.alloc(sizeof(double) * (size_t)n1 * n2));
    HIPCHK(c, hipMemcpyAsync(a.p, x1, sizeof(double) * d * n1, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b.p, x2, sizeof(double) * d * n2, hipMemcpyHostToDevice, c->stream));
    const Small S = small_from_colmajor(cov, (int)d);
    BQCHK(dev_int_K1_K2(c, (int)d, a.d(), (int)n1, b.d(), (int)n2, w1, w2, mu, S,
                        (h1 * h1) * (h2 * h2), od.d(), nullptr, nullptr));
    HIPCHK(c, hipMemcpyAsync(out, od.p, sizeof(double) * (size_t)n1 * n2, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

extern "C" int bq_int_int_K1_K2_K1(bq_ctx *c, const double *x, int64_t d, int64_t n, double h1,
                                   const double *w1, double h2, const double *w2, const double *mu,
                                   const double *cov, double *out)
{
    BQCHK(check_int_args(c, d, w1, mu, cov));
    if (!w2 || n < 0 || (n && (!x || !out)))
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (n == 0)
        return BQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf xd, wk, od;
    HIPCHK(c, xd.alloc(sizeof(double) * d * n));
    HIPCHK(c, wk.alloc(sizeof(double) * (d + 1) * n));
    HIPCHK(c, od.alloc(sizeof(double) * (size_t)n * n));
    HIPCHK(c, hipMemcpyAsync(xd.p, x, sizeof(double) * d * n, hipMemcpyHostToDevice, c->stream));
    const Small S = small_from_colmajor(cov, (int)d);
    BQCHK(dev_iikk(c, (int)d, xd.d(), (int)n, h1, w1, h2, w2, mu, S, wk.d(), od.d(), nullptr,
                   nullptr, nullptr));
    HIPCHK(c, hipMemcpyAsync(out, od.p, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

extern "C" int bq_int_int_K1_K2(bq_ctx *c, const double *x, int64_t d, int64_t n, double h1,
                                const double *w1, double h2, const double *w2, const double *mu,
                                const double *cov, double *out)
{
    BQCHK(check_int_args(c, d, w1, mu, cov));
    if (!w2 || n < 0 || (n && (!x || !out)))
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (n == 0)
        return BQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const Small S = small_from_colmajor(cov, (int)d);
    const Small W = cov_plus_w2(S, w1, 2.0);
    Small Wl = W, Winv;
    if (!small_chol(Wl) || !small_spd_inv(W, Winv))
        return fail(c, BQ_ERR_NOT_PD, "matrix is not positive definite");
    double logdet = 0.0;
    for (int r = 0; r < d; ++r)
        logdet += 2.0 * std::log(Wl(r, r));
    const double N0 = -0.5 * (d * LOG_2PI + logdet); // log N(0 | 0, W1 + 2S)
    const Small SWS = small_mul(small_mul(S, Winv), S);
    Small Cm;
    Cm.n = (int)d;
    for (int r = 0; r < d; ++r)
        for (int q = 0; q < d; ++q)
            Cm(r, q) = (r == q ? w2[r] * w2[r] : 0.0) + S(r, q) - SWS(r, q);
    DevBuf xd, od;
    HIPCHK(c, xd.alloc(sizeof(double) * d * n));
    HIPCHK(c, od.alloc(sizeof(double) * n));
    HIPCHK(c, hipMemcpyAsync(xd.p, x, sizeof(double) * d * n, hipMemcpyHostToDevice, c->stream));
    BQCHK(dev_int_K(c, (int)d, xd.d(), (int)n, mu, Cm, (h1 * h1) * (h2 * h2), N0, od.d(), nullptr,
                    nullptr, nullptr));
    HIPCHK(c, hipMemcpyAsync(out, od.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

// ===========================================================================
// resident-factor solve and the BQ moments on device-resident fits
// ===========================================================================
extern "C" int bq_gp_solve(bq_ctx *c, bq_fit *f, const double *B, int64_t nrhs, double *X)
{
    BQCHK(check_fit(c, f));
    if (nrhs < 0 || (nrhs && (!B || !X)))
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (nrhs == 0)
        return BQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const int n = f->n, npad = f->npad;
    if (nrhs == 1) {
        // one right-hand side: the GEMV sweeps (trsv.h)
        WideInv wv;
        BQCHK(fit_wide(c, f, wv));
        BQCHK(fit_vec(c, f));
        double *x = f->vec.d(), *y = f->vec.d() + npad;
        // (a timed-out hand-off of the one-launch sweeps re-issues the solve on the per-block
        // kernels: with_flow_fallback)
        BQCHK(with_flow_fallback(c, [&]() -> int {
            // through the fit's pinned staging vector (zero padded): truly asynchronous copies
            std::memcpy(f->hvec, B, sizeof(double) * n);
            std::memset(f->hvec + n, 0, sizeof(double) * (npad - n));
            if (c->solve_kcopy && trsv_flow_ok(c, npad, wv.B, true)) {
                // [x | y | ws_f | x_out | ws_b]: the vector comes in and goes out through kernels
                // on the mapped pinned vector, and one of them sets both sweeps' hand-off slots
                double *hdev = nullptr;
                HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&hdev), f->hvec, 0));
                const size_t wsn = trsv_flow_ws_doubles(npad, wv.B);
                double *wsf = y + npad, *xo = wsf + wsn, *wsb = xo + npad;
                BQCHK(launch_flow_in(c, hdev, n, x, npad, y, 2 * (size_t)npad + 2 * wsn));
                BQCHK(launch_trsv_flow(c, true, f->A.d(), f->ldl, npad, wv.B, wv.nr, wv.tt, x, y,
                                       wsf, true));
                BQCHK(launch_trsv_flow(c, false, f->A.d(), f->ldl, npad, wv.B, wv.nt, wv.uu, y, xo,
                                       wsb, true));
                BQCHK(launch_flow_out(c, xo, n, hdev));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                return BQ_OK;
            }
            double *hmap = nullptr;
            if (c->solve_kcopy)
                HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&hmap), f->hvec, 0));
            if (hmap)
                BQCHK(launch_flow_in(c, hmap, n, x, npad, nullptr, 0));
            else
                HIPCHK(c, hipMemcpyAsync(x, f->hvec, sizeof(double) * npad, hipMemcpyHostToDevice,
                                         c->stream));
            BQCHK(fit_replay(c, f, 0, [&]() -> int {
                double *ws = f->vec.d() + 2 * (size_t)npad;
                BQCHK(enqueue_forward_vec(c, x, y, f->A.d(), f->ldl, npad, wv, ws));
                return enqueue_backward_vec(c, y, x, f->A.d(), f->ldl, npad, wv, ws);
            }));
            if (hmap)
                BQCHK(launch_flow_out(c, x, n, hmap));
            else
                HIPCHK(c, hipMemcpyAsync(f->hvec, x, sizeof(double) * n, hipMemcpyDeviceToHost,
                                         c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            return BQ_OK;
        }));
        std::memcpy(X, f->hvec, sizeof(double) * n);
        return BQ_OK;
    }
    WideInv wi;
    BQCHK(fit_wide(c, f, wi));
    return solve_rows_host(c, f->A.d(), f->ldl, n, npad, wi, B, nrhs, X);
}

extern "C" int bq_bq_Z_mean(bq_ctx *c, bq_fit *gp_l, const double *mu, const double *cov,
                            double *out)
{
    BQCHK(check_fit(c, gp_l));
    if (!out)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    BQCHK(check_int_args(c, gp_l->d, gp_l->w, mu, cov));
    HIPCHK(c, hipSetDevice(c->device));
    BQCHK(fit_alpha(c, gp_l));
    const int d = gp_l->d, n = gp_l->n;
    Scratch sc(c);
    const size_t opart = sc.take((size_t)(n + 255) / 256 + 1);
    BQCHK(sc.commit());
    double *part = sc.at(opart);
    const Small S = small_from_colmajor(cov, d);
    double *sum_dev = part + (n + 255) / 256;
    BQCHK(dev_int_K(c, d, gp_l->pts.d(), n, mu, cov_plus_w2(S, gp_l->w, 1.0), gp_l->h * gp_l->h,
                    0.0, nullptr, gp_l->alpha.d(), part, sum_dev));
    HIPCHK(c, hipMemcpyAsync(out, sum_dev, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

extern "C" int bq_bq_Z_var(bq_ctx *c, bq_fit *gp_tl, bq_fit *gp_l, const double *mu,
                           const double *cov, double *out)
{
    BQCHK(check_fit(c, gp_tl));
    BQCHK(check_fit(c, gp_l));
    if (!out)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (gp_tl->d != gp_l->d)
        return fail(c, BQ_ERR_BAD_ARG, "dimension mismatch");
    BQCHK(check_int_args(c, gp_l->d, gp_l->w, mu, cov));
    HIPCHK(c, hipSetDevice(c->device));
    BQCHK(fit_alpha(c, gp_l));
    const int d = gp_l->d, ns = gp_tl->n, nsc = gp_l->n, npad = gp_tl->npad;
    const Small S = small_from_colmajor(cov, d);
    const int gx = (nsc + 63) / 64, gy = (nsc + 255) / 256;
    WideInv wi;
    BQCHK(fit_wide(c, gp_tl, wi));
    Scratch sc(c);
    const size_t owork = sc.take((size_t)(d + 1) * nsc), opart = sc.take((size_t)gx * gy + 4);
    BQCHK(sc.commit());
    double *work = sc.at(owork), *part = sc.at(opart);
    BQCHK(fit_vec(c, gp_tl));
    double *sol = gp_tl->vec.d(), *X = gp_tl->vec.d() + npad;
    double *scal = part + (size_t)gx * gy; // [0] = alpha' I alpha, [1] = -beta' K^-1 beta
    // alpha' (int int K_l K_tl K_l) alpha
    BQCHK(dev_iikk(c, d, gp_l->pts.d(), nsc, gp_l->h, gp_l->w, gp_tl->h, gp_tl->w, mu, S, work,
                   nullptr, gp_l->alpha.d(), part, scal));
    double hs[2];
    BQCHK(with_flow_fallback(c, [&]() -> int {
        // beta = (int K_tl K_l) alpha
        HIPCHK(c, hipMemsetAsync(sol, 0, sizeof(double) * (size_t)npad, c->stream));
        BQCHK(dev_int_K1_K2(c, d, gp_tl->pts.d(), ns, gp_l->pts.d(), nsc, gp_tl->w, gp_l->w, mu, S,
                            (gp_tl->h * gp_tl->h) * (gp_l->h * gp_l->h), nullptr, gp_l->alpha.d(),
                            sol));
        // beta' K_tl^-1 beta = |L_tl^-1 beta|^2: the forward sweep alone and a sum of squares
        // (the reference solves with both sweeps and takes the dot product, bq_c.pyx:348-351; the
        // symmetric form errs with cond(L) instead of cond(K))
        BQCHK(fit_replay(c, gp_tl, 2, [&]() -> int {
            return enqueue_forward_vec(c, sol, X, gp_tl->A.d(), gp_tl->ldl, npad, wi,
                                       gp_tl->vec.d() + 2 * (size_t)npad);
        }));
        BQCHK(launch_neg_sumsq(c, X, npad, scal + 1));
        HIPCHK(c, hipMemcpyAsync(hs, scal, sizeof hs, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return BQ_OK;
    }));
    *out = hs[0] + hs[1]; // hs[1] holds -|L^-1 beta|^2
    return BQ_OK;
}

// ===========================================================================
// batched expected-squared-mean systems
// ===========================================================================
extern "C" int bq_esm_batch(bq_ctx *c, const double *x_sc, const double *l_sc, int64_t ns,
                            int64_t nsc, const double *x_a, int64_t M, double h, double w,
                            double thresh, const double *mu, const double *cov, double *A_a,
                            double *A_sc_l, int32_t *status)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!x_sc || !l_sc || !mu || !cov || ns < 0 || nsc < ns || nsc < 1 || M < 0 ||
        (M && (!x_a || !A_a || !A_sc_l || !status)))
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    const double wv[1] = {w};
    BQCHK(check_w(c, 1, h, wv, 0.0));
    if (M == 0)
        return BQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    EsmLayout L;
    L.ns = (int)ns;
    L.nsc = (int)nsc;
    L.npad = (int)roundup(nsc + 1, 64);
    L.ntot = L.npad + 64;
    const long lda = pick_ld(L.ntot);
    // per candidate: the bordered matrix, the panel scratch of the one-launch sweep, the
    // diagonal factor's record, failure flag and outputs
    const size_t per = sizeof(double) * ((size_t)lda * L.ntot + panel_ws_doubles(L.ntot, 1) +
                                         BQ_DINV_STRIDE + 2) + sizeof(int);
    size_t freeb = 0, totalb = 0;
    HIPCHK(c, hipMemGetInfo(&freeb, &totalb));
    int64_t chunk = std::max<int64_t>(1, (int64_t)((freeb / 2) / per));
    chunk = std::min<int64_t>(std::min<int64_t>(chunk, M), 32768);

    const GaussParams g = make_params(1, h, wv, 0.0);
    // jitter amounts exactly as two successive improve_covariance_conditioning calls
    // produce them (bq.py:470-476): max(K_l) is the kernel scale on the diagonal
    const double eps = std::numeric_limits<double>::epsilon();
    std::vector<double> j1((size_t)M), j2((size_t)M);
    for (int64_t a = 0; a < M; ++a) {
        bool any = false;
        for (int64_t j = ns; j < nsc && !any; ++j)
            any = std::fabs(x_sc[j] - x_a[a]) < thresh;
        const double first = any ? std::max(eps, g.c) * 1e-4 : 0.0;
        j1[(size_t)a] = first;
        j2[(size_t)a] = std::max(eps, g.c + first) * 1e-4;
    }

    DevBuf xs, ls, xa, ik, ika, dj1, dj2, Ad, dinv, info, outd, panel;
    HIPCHK(c, xs.alloc(sizeof(double) * nsc));
    HIPCHK(c, ls.alloc(sizeof(double) * nsc));
    HIPCHK(c, xa.alloc(sizeof(double) * M));
    HIPCHK(c, ik.alloc(sizeof(double) * nsc));
    HIPCHK(c, ika.alloc(sizeof(double) * M));
    HIPCHK(c, dj1.alloc(sizeof(double) * M));
    HIPCHK(c, dj2.alloc(sizeof(double) * M));
    HIPCHK(c, Ad.alloc(sizeof(double) * (size_t)lda * L.ntot * (size_t)chunk));
    HIPCHK(c, dinv.alloc(sizeof(double) * BQ_DINV_STRIDE * (size_t)chunk));
    HIPCHK(c, panel.alloc(sizeof(double) * sweep_ws_doubles(c, L.ntot, (int)chunk)));
    HIPCHK(c, info.alloc(sizeof(int) * (size_t)chunk));
    HIPCHK(c, outd.alloc(sizeof(double) * 2 * (size_t)chunk));
    HIPCHK(c, hipMemcpyAsync(xs.p, x_sc, sizeof(double) * nsc, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(ls.p, l_sc, sizeof(double) * nsc, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(xa.p, x_a, sizeof(double) * M, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dj1.p, j1.data(), sizeof(double) * M, hipMemcpyHostToDevice,
                             c->stream));
    HIPCHK(c, hipMemcpyAsync(dj2.p, j2.data(), sizeof(double) * M, hipMemcpyHostToDevice,
                             c->stream));
    // int K(x, .) p(x) dx at the nsc shared points and at every candidate
    const Small S = small_from_colmajor(cov, 1);
    const Small WS = cov_plus_w2(S, wv, 1.0);
    BQCHK(dev_int_K(c, 1, xs.d(), (int)nsc, mu, WS, h * h, 0.0, ik.d(), nullptr, nullptr, nullptr));
    BQCHK(dev_int_K(c, 1, xa.d(), (int)M, mu, WS, h * h, 0.0, ika.d(), nullptr, nullptr, nullptr));

    std::vector<double> hout((size_t)chunk * 2);
    std::vector<int> hinfo((size_t)chunk);
    for (int64_t a0 = 0; a0 < M; a0 += chunk) {
        const int nb = (int)std::min(chunk, M - a0);
        HIPCHK(c, hipMemsetAsync(info.p, 0, sizeof(int) * nb, c->stream));
        {
            Bracket br(c, BQ_K_GRAM, 8.0 * L.ntot * (L.ntot + 1.0) / 2.0 * nb);
            dim3 grid((L.ntot + 127) / 128, (L.ntot + 63) / 64, nb);
            hipLaunchKernelGGL(assemble_esm_kernel, grid, dim3(256), 0, c->stream, xs.d(),
                               xa.d() + a0, ik.d(), ika.d() + a0, ls.d(), dj1.d() + a0,
                               dj2.d() + a0, thresh, g, Ad.d(), lda, lda * (long)L.ntot, L);
            HIPCHK(c, hipGetLastError());
        }
        BQCHK(enqueue_potrf_partial(c, Ad.d(), lda, lda * (long)L.ntot, nb, L.ntot, L.npad,
                                    dinv.d(), info.i(), panel.d(),
                                    panel.bytes / sizeof(double)));
        hipLaunchKernelGGL(esm_finalize_kernel, dim3((nb + 255) / 256), dim3(256), 0, c->stream,
                           Ad.d(), lda, lda * (long)L.ntot, L, nb, outd.d());
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(hout.data(), outd.p, sizeof(double) * 2 * nb,
                                 hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hinfo.data(), info.p, sizeof(int) * nb, hipMemcpyDeviceToHost,
                                 c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int k = 0; k < nb; ++k) {
            A_a[a0 + k] = hout[(size_t)2 * k];
            A_sc_l[a0 + k] = hout[(size_t)2 * k + 1];
            status[a0 + k] = hinfo[(size_t)k];
        }
    }
    return BQ_OK;
}

// ===========================================================================
// The acquisition loop as a bordered UPDATE of the resident factor.
//
// For a candidate x_a the reference factors the (nsc + 1)^2 matrix
//     [ K_sc + J_c   k_a       ]     K_sc = K_l(x_sc, x_sc), k_a = K_l(x_sc, x_a),
//     [ k_a^T        k0 + j_a  ]     J_c: jitter on the candidates closer to x_a than thresh
// from scratch and solves it against b = int K_l(x_sca, x) p(x) dx (bq.py:463-480,
// bq_c.pyx:455-470).  K_sc is gp_l's own Gram matrix, whose factor L is resident: with
// M = K_sc + J_c,
//     A_a    = (b_a - k_a^T M^-1 b_sc) / (k0 + j_a - k_a^T M^-1 k_a)
//     A_sc.l = l^T M^-1 b_sc - A_a l^T M^-1 k_a.
// Every term is a bilinear form u^T M^-1 v = (L_M^-1 u) . (L_M^-1 v) of k_a (M of them), b_sc
// and l -- sums of products, so the Schur complement in the denominator is the difference of
// sums of squares that the reference's last pivot is.  The jitter sits on candidate points,
// the LAST rows of x_sc: with the factor cut at a column p <= ns,
//     L = [ L11 0 ; L21 L22 ],   L_M = [ L11 0 ; L21 L22' ],   L22' L22'^T = L22 L22^T + J_c,
// so one forward sweep F = R L^-T over the M + 2 rows gives the leading p entries of every
// L_M^-1 u as they stand, and the trailing nt = nsc - p (< nc + 64) entries are
// L22'^-1 (L22 F[p:]) -- an nt x nt Cholesky per DISTINCT set of jittered candidates and two
// nt x nt triangular products per candidate, on the host.  The device does the O(n^2 M) part:
// the sweep, one GEMM of the leading parts against b and l, the squared norms.  No inverse
// of K_sc is formed and nothing is subtracted but the Schur complement itself.  (A first
// version corrected K_sc^-1 by Woodbury: with the jitter larger than K_sc's small
// eigenvalues it lost cond(K) -- 3e-10 against the CPU restatement's 7e-13 on the ill-conditioned case
// of test_acquisition_and_posterior_vs_extended_precision.)  status[a] = 1 where the Schur
// complement is not positive (bq.py:481-490's fallback).  Requires a noise-free gp_l (Kxoxo
// carries no s^2 term, bq.py:465): BQ_ERR_BAD_ARG else, and the caller uses bq_esm_batch.
// ===========================================================================
// largest host tail / number of distinct jittered sets bq_esm_border takes on itself
#define BQ_ESM_MAX_TAIL 320
#define BQ_ESM_MAX_GROUPS 64

extern "C" int bq_esm_border(bq_ctx *c, bq_fit *gp_l, int64_t ns, const double *x_a, int64_t M,
                             double thresh, const double *mu, const double *cov, double *A_a,
                             double *A_sc_l, int32_t *status)
{
    BQCHK(check_fit(c, gp_l));
    if (gp_l->d != 1)
        return fail(c, BQ_ERR_BAD_ARG, "esm_border: one-dimensional fits only");
    if (gp_l->s != 0.0)
        return fail(c, BQ_ERR_BAD_ARG, "esm_border: gp_l carries a noise term");
    const int64_t nsc = gp_l->n;
    if (!mu || !cov || ns < 0 || ns > nsc || M < 0 || (M && (!x_a || !A_a || !A_sc_l || !status)))
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (M == 0)
        return BQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const int nc = (int)(nsc - ns), npad = gp_l->npad;
    const int p = (int)(ns / 64 * 64);   // the cut: a multiple of 64 at or below ns
    const int nt = (int)nsc - p;         // trailing block handled on the host
    const double h = gp_l->h, wv[1] = {gp_l->w[0]};
    GaussParams g = make_params(1, h, wv, 0.0);
    // The trailing block lives on the host: an nt^3 / 3 Cholesky per DISTINCT set of jittered
    // candidates and O(nt^2) per candidate, single-threaded.  With many candidates (nt grows
    // with n_candidate) or many distinct sets that tail would dominate: then all candidates go
    // through the batched refactorisation on the device instead (bq_esm_batch).
    std::vector<double> xsc((size_t)nsc);
    HIPCHK(c, hipMemcpyAsync(xsc.data(), gp_l->pts.p, sizeof(double) * nsc, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {
        std::vector<std::vector<int>> sets;
        std::vector<int> cl;
        for (int64_t a = 0; a < M && sets.size() <= BQ_ESM_MAX_GROUPS; ++a) {
            cl.clear();
            for (int i = 0; i < nc; ++i)
                if (std::fabs(xsc[(size_t)(ns + i)] - x_a[a]) < thresh)
                    cl.push_back(i);
            if (std::find(sets.begin(), sets.end(), cl) == sets.end())
                sets.push_back(cl);
        }
        if (nt > BQ_ESM_MAX_TAIL || sets.size() > BQ_ESM_MAX_GROUPS) {
            std::vector<double> lsc((size_t)nsc);
            HIPCHK(c, hipMemcpyAsync(lsc.data(), gp_l->y.p, sizeof(double) * nsc,
                                     hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            return bq_esm_batch(c, xsc.data(), lsc.data(), ns, nsc, x_a, M, h, wv[0], thresh, mu,
                                cov, A_a, A_sc_l, status);
        }
    }
    // rows: [b_sc; l_sc; padding to 64] then the M borders
    const int T = 64, rb = 0, rl = 1;
    const int mrows = (int)roundup(T + M, 64);
    // temporaries from the context's scratch buffer (kept between calls: marginalize /
    // choose_next call this once per hyper-parameter sample)
    Scratch sc(c);
    const size_t oF = sc.take((size_t)mrows * npad), oG = sc.take((size_t)mrows * T),
                 oxa = sc.take((size_t)M), oik = sc.take((size_t)nsc), oika = sc.take((size_t)M),
                 osq = sc.take((size_t)mrows);
    BQCHK(sc.commit());
    struct View {
        double *q;
        void *p;
        double *d() const { return q; }
    };
    auto view = [&](size_t off) { return View{sc.at(off), sc.at(off)}; };
    const View F = view(oF), G = view(oG), xa = view(oxa), ik = view(oik), ika = view(oika),
               sq = view(osq);
    const View &R = F; // the sweep runs in place
    HIPCHK(c, hipMemcpyAsync(xa.p, x_a, sizeof(double) * M, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(R.p, 0, sizeof(double) * (size_t)mrows * npad, c->stream));
    HIPCHK(c, hipMemsetAsync(G.p, 0, sizeof(double) * (size_t)mrows * T, c->stream));
    HIPCHK(c, hipMemsetAsync(sq.p, 0, sizeof(double) * (size_t)mrows, c->stream));
    const Small S = small_from_colmajor(cov, 1);
    const Small WS = cov_plus_w2(S, wv, 1.0);
    BQCHK(dev_int_K(c, 1, gp_l->pts.d(), (int)nsc, mu, WS, h * h, 0.0, ik.d(), nullptr, nullptr,
                    nullptr));
    BQCHK(dev_int_K(c, 1, xa.d(), (int)M, mu, WS, h * h, 0.0, ika.d(), nullptr, nullptr, nullptr));
    HIPCHK(c, hipMemcpy2DAsync(R.d() + rb, sizeof(double) * mrows, ik.p, sizeof(double),
                               sizeof(double), nsc, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(R.d() + rl, sizeof(double) * mrows, gp_l->y.p, sizeof(double),
                               sizeof(double), nsc, hipMemcpyDeviceToDevice, c->stream));
    // rows T .. T + M - 1: the borders k_a^T
    BQCHK(launch_gram_cross(c, 1, xa.d(), (int)M, gp_l->pts.d(), (int)nsc, g, R.d() + T, mrows));
    // F = R L^-T (the accurate 64-column sweep, see enqueue_forward_rows_blk); over the leading
    // p columns G = -F F_tail^T (mrows x T) and the squared norms
    BQCHK(fit_dw(c, gp_l));
    BQCHK(enqueue_forward_rows_blk(c, F.d(), mrows, mrows, gp_l->A.d(), gp_l->ldl, npad,
                                   gp_l->dw.d()));
    if (p > 0) {
        BQCHK(launch_gemm(c, BQ_K_GEMM, G.d(), mrows, 0, F.d(), mrows, 0, F.d(), 1, mrows, 0, mrows,
                          T, p, 0, 1));
        BQCHK(launch_rowdot(c, F.d(), (long)mrows, mrows, mrows, p, nullptr, 0.0, nullptr, sq.d()));
    }
    // (of G only columns rb and rl -- the products with the b and l rows -- are read)
    std::vector<double> hG((size_t)mrows * 2), hsq((size_t)mrows), hb((size_t)M),
        F2((size_t)mrows * nt), L22((size_t)nt * nt);
    HIPCHK(c, hipMemcpyAsync(hG.data(), G.p, sizeof(double) * hG.size(), hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipMemcpyAsync(hsq.data(), sq.p, sizeof(double) * mrows, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipMemcpyAsync(hb.data(), ika.p, sizeof(double) * M, hipMemcpyDeviceToHost,
                             c->stream));
    // F2[r + mrows j] = F[r, p + j]; L22[i + nt j] = L[p + i, p + j]
    HIPCHK(c, hipMemcpyAsync(F2.data(), F.d() + (size_t)p * mrows, sizeof(double) * F2.size(),
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(L22.data(), sizeof(double) * nt,
                               gp_l->A.d() + p + (size_t)p * gp_l->ldl, sizeof(double) * gp_l->ldl,
                               sizeof(double) * nt, nt, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int j = 1; j < nt; ++j) // strict upper triangle of the factor block is not the factor's
        for (int i = 0; i < j; ++i)
            L22[(size_t)i + (size_t)nt * j] = 0.0;
    // leading-part dot of any row r with tail row t (the GEMM stored its negative)
    auto dot1 = [&](int r, int t) { return p > 0 ? -hG[(size_t)r + (size_t)mrows * t] : 0.0; };
    // the trailing part of L_M^-1 u for row r: L22'^-1 (L22 F2[r, :])
    auto trail = [&](const std::vector<double> &Lm, bool same, int r, std::vector<double> &out) {
        out.assign((size_t)nt, 0.0);
        if (same) { // L22' = L22: the sweep's own entries
            for (int i = 0; i < nt; ++i)
                out[(size_t)i] = F2[(size_t)r + (size_t)mrows * i];
            return;
        }
        for (int i = 0; i < nt; ++i) { // w = L22 F2_r
            double v = 0.0;
            for (int q = 0; q <= i; ++q)
                v += L22[(size_t)i + (size_t)nt * q] * F2[(size_t)r + (size_t)mrows * q];
            out[(size_t)i] = v;
        }
        for (int i = 0; i < nt; ++i) { // forward substitution with L22'
            double v = out[(size_t)i];
            for (int q = 0; q < i; ++q)
                v -= Lm[(size_t)i + (size_t)nt * q] * out[(size_t)q];
            out[(size_t)i] = v / Lm[(size_t)i + (size_t)nt * i];
        }
    };
    auto vdot = [&](const std::vector<double> &u, const std::vector<double> &v) {
        double sacc = 0.0;
        for (int i = 0; i < nt; ++i)
            sacc += u[(size_t)i] * v[(size_t)i];
        return sacc;
    };
    // trailing Schur complement S22 = L22 L22^T
    std::vector<double> S22((size_t)nt * nt, 0.0);
    for (int j = 0; j < nt; ++j)
        for (int i = j; i < nt; ++i) {
            double v = 0.0;
            for (int q = 0; q <= j; ++q)
                v += L22[(size_t)i + (size_t)nt * q] * L22[(size_t)j + (size_t)nt * q];
            S22[(size_t)i + (size_t)nt * j] = v;
        }
    struct Group {
        std::vector<int> close;
        std::vector<double> Lm, fb, fl;
        double lMb = 0.0;
        bool ok = true;
    };
    std::vector<Group> groups;
    const double eps = std::numeric_limits<double>::epsilon();
    const double j1v = std::max(eps, g.c) * 1e-4;
    std::vector<int> close;
    std::vector<double> fa;
    for (int64_t a = 0; a < M; ++a) {
        const int ra = T + (int)a;
        close.clear();
        for (int i = 0; i < nc; ++i)
            if (std::fabs(xsc[(size_t)(ns + i)] - x_a[a]) < thresh)
                close.push_back(i);
        // jitter exactly as two successive improve_covariance_conditioning calls produce
        // it (bq.py:470-476): max(K_l) is the kernel scale on the diagonal
        const double j1 = close.empty() ? 0.0 : j1v;
        const double j2 = std::max(eps, g.c + j1) * 1e-4;
        Group *gr = nullptr;
        for (auto &q : groups)
            if (q.close == close)
                gr = &q;
        if (!gr) {
            groups.emplace_back();
            gr = &groups.back();
            gr->close = close;
            // L22' = chol(S22 + J_c) (the factor block itself when nothing is jittered)
            gr->Lm = close.empty() ? L22 : S22;
            if (!close.empty()) {
                for (int i : close)
                    gr->Lm[(size_t)(ns - p + i) * (nt + 1)] += j1;
                for (int k = 0; k < nt && gr->ok; ++k) {
                    double d = gr->Lm[(size_t)k + (size_t)nt * k];
                    for (int q = 0; q < k; ++q)
                        d -= gr->Lm[(size_t)k + (size_t)nt * q] * gr->Lm[(size_t)k + (size_t)nt * q];
                    if (!(d > 0.0)) {
                        gr->ok = false;
                        break;
                    }
                    d = std::sqrt(d);
                    gr->Lm[(size_t)k + (size_t)nt * k] = d;
                    for (int i = k + 1; i < nt; ++i) {
                        double v = gr->Lm[(size_t)i + (size_t)nt * k];
                        for (int q = 0; q < k; ++q)
                            v -= gr->Lm[(size_t)i + (size_t)nt * q] * gr->Lm[(size_t)k + (size_t)nt * q];
                        gr->Lm[(size_t)i + (size_t)nt * k] = v / d;
                    }
                }
            }
            if (gr->ok) {
                trail(gr->Lm, gr->close.empty(), rb, gr->fb);
                trail(gr->Lm, gr->close.empty(), rl, gr->fl);
                gr->lMb = dot1(rl, rb) + vdot(gr->fl, gr->fb);
            }
        }
        bool ok = gr->ok;
        double schur = 0.0, kMb = 0.0, lMk = 0.0;
        if (ok) {
            trail(gr->Lm, gr->close.empty(), ra, fa);
            const double kMk = -hsq[(size_t)ra] + vdot(fa, fa); // rowdot: 0 - |leading part|^2
            kMb = dot1(ra, rb) + vdot(fa, gr->fb);
            lMk = dot1(ra, rl) + vdot(fa, gr->fl);
            schur = g.c + j2 - kMk;
            ok = schur > 0.0 && std::isfinite(schur);
        }
        if (!ok) {
            status[a] = 1;
            A_a[a] = 0.0;
            A_sc_l[a] = 0.0;
            continue;
        }
        status[a] = 0;
        A_a[a] = (hb[(size_t)a] - kMb) / schur;
        A_sc_l[a] = gr->lMb - A_a[a] * lMk;
    }
    return BQ_OK;
}
