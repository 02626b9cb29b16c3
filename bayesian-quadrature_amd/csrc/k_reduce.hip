// k_reduce.hip -- read-outs, single-vector sweeps and small utilities (reduce.h, trsv.h) and
// their launchers.
#include "host.h"
#include "reduce.h"
#include "trsv.h"
#include "trsvflow.h"

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
                  double k0, double *mean, double *var, long zstride)
{
    Bracket br(c, BQ_K_REDUCE);
    hipLaunchKernelGGL(rowdot_kernel, dim3(Mp / 16), dim3(1024), 0, c->stream, V, ldv, M, npad, z,
                       zstride, k0, mean, var);
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

// doubles of workspace a one-launch sweep needs behind its vectors: the ticket (one 64-byte
// line) and the x versions 1 .. ns - 1
size_t trsv_flow_ws_doubles(int npad, int B)
{
    const int ns = (npad + B - 1) / B;
    return 8 + (size_t)(ns > 1 ? ns - 1 : 0) * npad;
}

bool trsv_flow_ok(const bq_ctx *c, int npad, int B, bool prefilled)
{
    // (below 2048 rows the sweep is two to four steps: the memsets cost what the launches did --
    // unless the caller fills the slots on its way in, bq_gp_solve: then from 1024 rows, half the
    // threshold: n = 1000 0.061 -> 0.053 ms, 1536 0.077 -> 0.074)
    return c->trsv_flow && c->flow_abort && B <= 512 && (B & 63) == 0 && (npad & 63) == 0 &&
           npad >= (prefilled ? c->trsv_flow_min / 2 : c->trsv_flow_min) && (npad + B - 1) / B >= 2;
}

// ws: trsv_flow_ws_doubles(npad, B) doubles; x0 is read, y written (npad each)
int launch_flow_in(bq_ctx *c, const double *hsrc, int n, double *x, int npad, double *fill,
                   size_t nfill)
{
    const long blocks = std::min<long>(1024, (std::max<long>(npad, (long)nfill) + 255) / 256);
    hipLaunchKernelGGL(flow_in_kernel, dim3((unsigned)blocks), dim3(256), 0, c->cur, hsrc, n, x, npad,
                       reinterpret_cast<unsigned long long *>(fill), (long)nfill);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_copy_words2(bq_ctx *c, void *d1, const void *s1, size_t n1, void *d2, const void *s2,
                       size_t n2)
{
    hipLaunchKernelGGL(copy_words2_kernel, dim3(1), dim3(256), 0, c->cur,
                       static_cast<unsigned long long *>(d1),
                       static_cast<const unsigned long long *>(s1), (int)n1,
                       static_cast<unsigned long long *>(d2),
                       static_cast<const unsigned long long *>(s2), (int)n2);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_plan_scatter(bq_ctx *c, const double *stage, int nprob, int d, int n, int M, int ntot,
                        int npad, int gw, double *gp, double *pts, double *yd)
{
    const long total = (long)nprob * (gw + (long)d * n + (long)d * M + n);
    hipLaunchKernelGGL(plan_scatter_kernel, dim3((unsigned)std::min<long>(256, (total + 255) / 256)),
                       dim3(256), 0, c->cur, stage, nprob, d, n, M, ntot, npad, gw, gp, pts, yd);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_plan_gather(bq_ctx *c, double *out, const double *scal, const int *info,
                       const double *mean, const double *var, int nb, int M)
{
    const long total = std::max<long>(4L * nb, (long)M * nb);
    hipLaunchKernelGGL(plan_gather_kernel, dim3((unsigned)std::min<long>(64, (total + 255) / 256)),
                       dim3(256), 0, c->cur, out, scal, info, mean, var, nb, M);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_small_potrs(bq_ctx *c, double *stage, int n, int nrhs)
{
    hipLaunchKernelGGL(small_potrs_kernel, dim3(1), dim3(1024), 0, c->cur, stage, n, nrhs);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_mat_in(bq_ctx *c, const double *stage, int n, double *A, long lda, int ntot, int *info)
{
    const long total = (long)ntot * ntot;
    hipLaunchKernelGGL(mat_in_kernel, dim3((unsigned)std::min<long>(256, (total + 255) / 256)),
                       dim3(256), 0, c->cur, stage, n, A, lda, ntot, info);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_mat_out(bq_ctx *c, double *out, const double *A, long lda, int n, const int *info)
{
    const long total = (long)n * n;
    hipLaunchKernelGGL(mat_out_kernel, dim3((unsigned)std::min<long>(256, (total + 255) / 256)),
                       dim3(256), 0, c->cur, out, A, lda, n, info);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// the context's mapped pinned staging buffer, at least `words` doubles: host and device views
int ctx_stage(bq_ctx *c, size_t words, double **host, double **dev)
{
    if (c->hstage_len < words) {
        if (c->hstage)
            (void)hipHostFree(c->hstage);
        c->hstage = nullptr;
        c->hstage_len = 0;
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->hstage), sizeof(double) * words));
        c->hstage_len = words;
    }
    *host = c->hstage;
    HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(dev), c->hstage, 0));
    return BQ_OK;
}

int launch_gather_row(bq_ctx *c, double *dst, const double *src, long stride, int n)
{
    hipLaunchKernelGGL(gather_row_kernel, dim3((n + 255) / 256), dim3(256), 0, c->cur, dst, src,
                       stride, n);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_flow_out(bq_ctx *c, const double *x, int n, double *hdst)
{
    hipLaunchKernelGGL(flow_out_kernel, dim3((n + 255) / 256), dim3(256), 0, c->cur, x, n, hdst);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_trsv_flow(bq_ctx *c, bool forward, const double *L, long ldl, int npad, int B,
                     const double *m1, const double *m2, const double *x0, double *y, double *ws,
                     bool preset)
{
    const int ns = (npad + B - 1) / B, last = (npad - 1) / B * B;
    long blocks = 0;
    double work = 0.0;
    for (int s = 0; s < ns; ++s) {
        const int J = forward ? s * B : last - s * B;
        const int bJ = std::min(B, npad - J);
        const int nupd = forward ? (s > 0 ? (npad - J - bJ) / 64 : 0) : (s > 0 ? J / 64 : 0);
        blocks += bJ / 16 + nupd;
        work += (double)bJ * bJ + 2.0 * B * (bJ + 64.0 * nupd);
    }
    // every slot starts as the sentinel (all bits set); the ticket counter starts at -1 with it
    // (preset: the caller has filled ws and y already -- flow_in_kernel, both sweeps of a solve)
    if (!preset) {
        HIPCHK(c, hipMemsetAsync(ws, 0xFF, sizeof(double) * trsv_flow_ws_doubles(npad, B), c->cur));
        HIPCHK(c, hipMemsetAsync(y, 0xFF, sizeof(double) * (size_t)npad, c->cur));
    }
    // (BQ_FLOW_FAULT=1, read per launch: the forward sweep loses a hand-off -- tests of the time-out)
    const int fault = std::getenv("BQ_FLOW_FAULT") ? std::atoi(std::getenv("BQ_FLOW_FAULT")) : 0;
    Bracket br(c, BQ_K_GEMM, work);
    int *ticket = reinterpret_cast<int *>(ws);
    double *xv = ws + 8;
    if (!forward)
        hipLaunchKernelGGL(trsv_bwd_flow_kernel, dim3((unsigned)blocks), dim3(1024), 0, c->cur, L, ldl,
                           npad, B, m1, m2, x0, xv, y, ticket, c->flow_abort, fault);
    else if (B == 512)
        hipLaunchKernelGGL(trsv_fwd_flow_kernel<8>, dim3((unsigned)blocks), dim3(1024), 0, c->cur, L,
                           ldl, npad, B, m1, m2, x0, xv, y, ticket, c->flow_abort, fault);
    else if (B == 256)
        hipLaunchKernelGGL(trsv_fwd_flow_kernel<4>, dim3((unsigned)blocks), dim3(1024), 0, c->cur, L,
                           ldl, npad, B, m1, m2, x0, xv, y, ticket, c->flow_abort, fault);
    else
        hipLaunchKernelGGL(trsv_fwd_flow_kernel<0>, dim3((unsigned)blocks), dim3(1024), 0, c->cur, L,
                           ldl, npad, B, m1, m2, x0, xv, y, ticket, c->flow_abort, fault);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// reads and clears the abort word (call with the stream synchronised)
bool flow_timed_out(bq_ctx *c)
{
    if (c->flow_abort && *static_cast<volatile int *>(c->flow_abort) != 0) {
        *c->flow_abort = 0;
        return true;
    }
    return false;
}

// a time-out nobody handled (every entry point that launches a one-launch sweep re-issues the
// solve itself: with_flow_fallback): bq_ctx_sync reports it
int flow_check(bq_ctx *c)
{
    if (flow_timed_out(c))
        return fail(c, BQ_ERR_HIP, "single-vector sweep: a hand-off between workgroups timed out "
                                   "(BQ_TRSV_FLOW=0 selects the one-launch-per-block sweeps)");
    return BQ_OK;
}

} // namespace bqh
