// k_gram.hip -- the Gram kernels (gram.h) and their launchers.
#include "host.h"
#include "gram.h"

namespace bqh {

template <int D>
void launch_gram_sym_d(bq_ctx *c, const double *x, long xstride, const GaussParams *gp,
                       int gpstride, double *K, long ldk, long kstride, int n, int batch)
{
    // whole 64 x 64 blocks: every exp once, block and transpose stored (N = 4096: 20 us against
    // 27 us with the full sweep below, which stays for ragged sizes)
    if ((n % 64) == 0 && (ldk % 2) == 0) {
        const int T = n / 64;
        hipLaunchKernelGGL(gram_tri_kernel<D>, dim3(T * (T + 1) / 2, 1, batch), dim3(256), 0, c->cur,
                           x, xstride, gp, gpstride, K, ldk, kstride, n);
        return;
    }
    dim3 grid((n + 127) / 128, (n + 63) / 64, batch);
    hipLaunchKernelGGL(gram_sym_kernel<D>, grid, dim3(256), 0, c->cur, x, xstride, gp, gpstride,
                       K, ldk, kstride, n);
}

int launch_gram_sym(bq_ctx *c, int d, const double *x, long xstride, const GaussParams *gp,
                    int gpstride, double *K, long ldk, long kstride, int n, int batch)
{
    Bracket br(c, BQ_K_GRAM, (8.0 * n * n + 8.0 * d * n) * batch);
    switch (d) {
    case 1: launch_gram_sym_d<1>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 2: launch_gram_sym_d<2>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 3: launch_gram_sym_d<3>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 4: launch_gram_sym_d<4>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 5: launch_gram_sym_d<5>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 6: launch_gram_sym_d<6>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 7: launch_gram_sym_d<7>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 8: launch_gram_sym_d<8>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    default: return fail(c, BQ_ERR_BAD_ARG, "d must be in 1..%d", BQ_MAXD);
    }
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

template <int D>
void launch_gram_cross_d(bq_ctx *c, const double *x1, int n1, const double *x2, int n2,
                         const GaussParams &g, double *K, long ldk)
{
    dim3 grid((n1 + 63) / 64, (n2 + 63) / 64, 1);
    hipLaunchKernelGGL(gram_cross_kernel<D>, grid, dim3(256), 0, c->cur, x1, n1, x2, n2, g, K,
                       ldk);
}

// K (n1p x n2p, zero padded beyond n1 x n2) in one launch: gram_cross_pad_kernel
int launch_gram_cross_pad(bq_ctx *c, int d, const double *x1, int n1, int n1p, const double *x2,
                          int n2, int n2p, const GaussParams &g, double *K, long ldk)
{
    if (n1p <= 0 || n2p <= 0)
        return BQ_OK;
    if ((n1p & 63) || (n2p & 63) || n1 > n1p || n2 > n2p)
        return fail(c, BQ_ERR_BAD_ARG, "gram_cross_pad: padded sizes must be multiples of 64");
    Bracket br(c, BQ_K_GRAM, 8.0 * n1p * n2p);
    dim3 grid(n1p / 64, n2p / 64, 1);
#define GCP(D_)                                                                                    \
    hipLaunchKernelGGL(gram_cross_pad_kernel<D_>, grid, dim3(256), 0, c->cur, x1, n1, x2, n2, g, K, \
                       ldk)
    switch (d) {
    case 1: GCP(1); break;
    case 2: GCP(2); break;
    case 3: GCP(3); break;
    case 4: GCP(4); break;
    case 5: GCP(5); break;
    case 6: GCP(6); break;
    case 7: GCP(7); break;
    case 8: GCP(8); break;
    default: return fail(c, BQ_ERR_BAD_ARG, "d must be in 1..%d", BQ_MAXD);
    }
#undef GCP
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_gram_cross(bq_ctx *c, int d, const double *x1, int n1, const double *x2, int n2,
                      const GaussParams &g, double *K, long ldk)
{
    if (n1 <= 0 || n2 <= 0)
        return BQ_OK;
    Bracket br(c, BQ_K_GRAM, 8.0 * n1 * n2);
    switch (d) {
    case 1: launch_gram_cross_d<1>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 2: launch_gram_cross_d<2>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 3: launch_gram_cross_d<3>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 4: launch_gram_cross_d<4>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 5: launch_gram_cross_d<5>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 6: launch_gram_cross_d<6>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 7: launch_gram_cross_d<7>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 8: launch_gram_cross_d<8>(c, x1, n1, x2, n2, g, K, ldk); break;
    default: return fail(c, BQ_ERR_BAD_ARG, "d must be in 1..%d", BQ_MAXD);
    }
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

} // namespace bqh
