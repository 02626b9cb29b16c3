// k_reduce.hip -- read-outs, single-vector sweeps and small utilities (reduce.h, trsv.h) and
// their launchers.
#include "host.h"
#include "reduce.h"
#include "trsv.h"

namespace bqh {

int launch_finalize(bq_ctx *c, const double *A, long lda, long astride, Layout L, double *scal,
                    double *mean, double *var, long mstride, int batch, double work)
{
    Bracket br(c, BQ_K_REDUCE, work);
    hipLaunchKernelGGL(finalize_kernel, dim3(1, 1, batch), dim3(256), 0, c->stream, A, lda, astride,
                       L, scal, mean, var, mstride);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_plan_readout(bq_ctx *c, const double *A, long lda, long astride, Layout L,
                        const GaussParams *gp, double *scal, double *mean, double *var,
                        long mstride, int batch)
{
    Bracket br(c, BQ_K_REDUCE, 8.0 * (L.M + 1.0) * L.npad * batch);
    hipLaunchKernelGGL(plan_readout_kernel, dim3((L.M + 16) / 16, 1, batch), dim3(1024), 0,
                       c->stream, A, lda, astride, L, gp, scal, mean, var, mstride);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// mean_i = v_i . z, var_i = k0 - |v_i|^2 over the Mp (multiple of 16) rows of V
int launch_rowdot(bq_ctx *c, const double *V, long ldv, int M, int Mp, int npad, const double *z,
                  double k0, double *mean, double *var)
{
    Bracket br(c, BQ_K_REDUCE);
    hipLaunchKernelGGL(rowdot_kernel, dim3(Mp / 16), dim3(1024), 0, c->stream, V, ldv, M, npad, z,
                       k0, mean, var);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_predict_mean(bq_ctx *c, int d, const double *xo, int M, const double *pts, int n,
                        const double *alpha, const GaussParams &g, double *mean)
{
    Bracket br(c, BQ_K_REDUCE);
    dim3 grid((unsigned)((M + 3) / 4));
#define PM(D_)                                                                                     \
    hipLaunchKernelGGL(predict_mean_kernel<D_>, grid, dim3(256), 0, c->stream, xo, M, pts, n,      \
                       alpha, g, mean)
    switch (d) {
    case 1: PM(1); break;
    case 2: PM(2); break;
    case 3: PM(3); break;
    case 4: PM(4); break;
    case 5: PM(5); break;
    case 6: PM(6); break;
    case 7: PM(7); break;
    default: PM(8); break;
    }
#undef PM
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_neg_identity(bq_ctx *c, double *nr, int B, int npad)
{
    hipLaunchKernelGGL(neg_identity_kernel, dim3((npad + 255) / 256), dim3(256), 0, c->stream, nr,
                       B, npad);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_pad_identity(bq_ctx *c, double *A, long lda, int n, int ntot)
{
    hipLaunchKernelGGL(pad_identity_kernel, dim3((ntot + 255) / 256, ntot), dim3(256), 0, c->stream,
                       A, lda, n, ntot);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// out = 2 sum log diag[i * (stride + 1)]
int launch_logdet(bq_ctx *c, const double *diag, long stride, int n, double *out)
{
    hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(256), 0, c->stream, diag, stride, n, out);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// dst (cols x rows, ld ldd) = src (rows x cols, ld lds)^T, 64 x 64 tiles
int launch_transpose_pad(bq_ctx *c, const double *src, long lds, int rows, int cols, double *dst,
                         long ldd)
{
    const dim3 g((rows + 63) / 64, (unsigned)((cols + 63) / 64));
    hipLaunchKernelGGL(transpose_pad_kernel, g, dim3(256), 0, c->stream, src, lds, rows, cols, dst,
                       ldd);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// dst block = src block^T for `batch` blocks of ld B, bx x by tiles of 64 each
int launch_transpose_blocks(bq_ctx *c, const double *src, double *dst, int B, long bstride,
                            int bx, int by, int batch)
{
    hipLaunchKernelGGL(transpose_blocks_kernel, dim3(bx, by, batch), dim3(256), 0, c->stream, src,
                       dst, B, bstride);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_neg_sumsq(bq_ctx *c, const double *v, int n, double *out)
{
    hipLaunchKernelGGL(neg_sumsq_kernel, dim3(1), dim3(256), 0, c->stream, v, n, out);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// one B-column step of the forward single-vector sweep (trsv.h)
int launch_trsv_fwd(bq_ctx *c, const double *L, long ldl, int J, int bJ, int B, int nupd,
                    const double *nr, const double *tt, double *x, double *y, double work)
{
    Bracket br(c, BQ_K_GEMM, work);
#define BQ_TRSV_FWD(NB_)                                                                           \
    hipLaunchKernelGGL((trsv_fwd_step_kernel<NB_>), dim3(bJ / 16 + nupd), dim3(1024), 0, c->cur,  \
                       L, ldl, J, bJ, B, nr, tt, x, y)
    if (B == 512)
        BQ_TRSV_FWD(8);
    else if (B == 256)
        BQ_TRSV_FWD(4);
    else
        BQ_TRSV_FWD(0);
#undef BQ_TRSV_FWD
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_trsv_bwd(bq_ctx *c, const double *L, long ldl, int J, int bJ, int B, int bn, int nupd,
                    const double *nt, const double *uu, double *x, double *y, double work)
{
    Bracket br(c, BQ_K_GEMM, work);
    hipLaunchKernelGGL(trsv_bwd_step_kernel, dim3(bJ / 16 + nupd), dim3(1024), 0, c->cur, L, ldl, J,
                       bJ, B, bn, nt, uu, x, y);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

} // namespace bqh
