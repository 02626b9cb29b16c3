// k_panel.hip -- the panel side of the factorisation: assembly of the bordered system (with
// the sweep's first launch folded in), the 64 x 64 diagonal factor, the MFMA panel solve and
// the one-launch steps (gram.h, potf2.h, trsm.h, slab.h), and their launchers.
#include "host.h"
#include "gram.h"
#include "trsm.h"
#include "slab.h"

namespace bqh {

template <int D>
void launch_assemble_d(bq_ctx *c, const double *pts, long pstride, const double *y, long ystride,
                       const GaussParams *gp, int gpstride, double *A, long lda, long astride,
                       Layout L, int batch, const FirstStep &fs, int jcols)
{
    dim3 grid((L.ntot + 127) / 128, ((jcols > 0 ? jcols : L.ntot) + 63) / 64, batch);
    const long wgs = (long)grid.x * grid.y * grid.z;
    if (fs.S0 && c->potf2_8w && wgs <= 2L * c->cus)
        hipLaunchKernelGGL((assemble_first_kernel<D, 8>), grid, dim3(512), 0, c->cur, pts, pstride, y,
                           ystride, gp, gpstride, A, lda, astride, L, fs.S0, fs.lds, fs.sstride,
                           fs.dinv, (long)BQ_DINV_STRIDE, fs.info, fs.scal);
    else if (fs.S0)
        hipLaunchKernelGGL((assemble_first_kernel<D, 4>), grid, dim3(256), 0, c->cur, pts, pstride, y,
                           ystride, gp, gpstride, A, lda, astride, L, fs.S0, fs.lds, fs.sstride,
                           fs.dinv, (long)BQ_DINV_STRIDE, fs.info, fs.scal);
    else
        hipLaunchKernelGGL(assemble_kernel<D>, grid, dim3(256), 0, c->cur, pts, pstride, y, ystride,
                           gp, gpstride, A, lda, astride, L);
}

int launch_assemble(bq_ctx *c, int d, const double *pts, long pstride, const double *y,
                    long ystride, const GaussParams *gp, int gpstride, double *A, long lda,
                    long astride, Layout L, int batch, const FirstStep &fs, int jcols)
{
    if (jcols < 0 || (jcols & 63) || jcols > L.ntot || (jcols > 0 && fs.S0))
        return fail(c, BQ_ERR_BAD_ARG, "assemble: column limit");
    const double cols = jcols > 0 ? jcols : L.ntot;
    Bracket br(c, BQ_K_GRAM, 8.0 * cols * (L.ntot - 0.5 * cols + 0.5) * batch);
    switch (d) {
    case 1: launch_assemble_d<1>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs, jcols); break;
    case 2: launch_assemble_d<2>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs, jcols); break;
    case 3: launch_assemble_d<3>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs, jcols); break;
    case 4: launch_assemble_d<4>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs, jcols); break;
    case 5: launch_assemble_d<5>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs, jcols); break;
    case 6: launch_assemble_d<6>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs, jcols); break;
    case 7: launch_assemble_d<7>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs, jcols); break;
    case 8: launch_assemble_d<8>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs, jcols); break;
    default: return fail(c, BQ_ERR_BAD_ARG, "d must be in 1..%d", BQ_MAXD);
    }
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_assemble_region(bq_ctx *c, const GramSeed &sd, double *A, long lda, long astride, int m,
                           int n, int batch)
{
    if (m <= 0 || n <= 0)
        return BQ_OK;
    if ((m & 63) || (n & 63) || (sd.r & 63) || (sd.c & 63))
        return fail(c, BQ_ERR_BAD_ARG, "assemble_region: multiples of 64");
    Bracket br(c, BQ_K_GRAM, 8.0 * m * n * batch);
    const dim3 grid((m + 127) / 128, n / 64, batch);
#define BQ_ASM_REGION(D_)                                                                          \
    case D_:                                                                                       \
        hipLaunchKernelGGL(assemble_region_kernel<D_>, grid, dim3(256), 0, c->cur, sd.pts,         \
                           sd.pstride, sd.y, sd.ystride, sd.gp, sd.gpstride, A, lda, astride, sd.L, \
                           sd.r, sd.c, m, n);                                                      \
        break;
    switch (sd.d) {
        BQ_ASM_REGION(1)
        BQ_ASM_REGION(2)
        BQ_ASM_REGION(3)
        BQ_ASM_REGION(4)
        BQ_ASM_REGION(5)
        BQ_ASM_REGION(6)
        BQ_ASM_REGION(7)
        BQ_ASM_REGION(8)
    default:
        return fail(c, BQ_ERR_BAD_ARG, "d must be in 1..%d", BQ_MAXD);
    }
#undef BQ_ASM_REGION
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_potf2(bq_ctx *c, double *A, long lda, long astride, int j0, double *dinv, long dstride,
                 int *info, int batch)
{
    Bracket br(c, BQ_K_POTF2, 64.0 * 64 * 64 / 3.0 * batch);
    // (eight waves where every matrix of the batch has a CU: potf2f_body<8>)
    if (c->potf2_8w && batch <= c->cus)
        hipLaunchKernelGGL(potf2_kernel<8>, dim3(1, 1, batch), dim3(512), 0, c->cur, A, lda, astride,
                           j0, dinv, dstride, info);
    else
        hipLaunchKernelGGL(potf2_kernel<4>, dim3(1, 1, batch), dim3(256), 0, c->cur, A, lda, astride,
                           j0, dinv, dstride, info);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// the kb x kb diagonal block of every matrix of a batch, one workgroup per matrix
// (potrf_wg_kernel); rec: kb / 64 records per matrix, rstride doubles apart
int launch_potrf_wg(bq_ctx *c, double *A, long lda, long astride, int kb, double *rec, long rstride,
                    int *info, int col0, int batch)
{
    Bracket br(c, BQ_K_POTF2, (double)kb * kb * kb / 3.0 * batch);
    hipLaunchKernelGGL(potrf_wg_kernel, dim3(batch), dim3(512), 0, c->cur, A, lda, astride, kb, rec,
                       rstride, info, col0);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// the MFMA panel solve (trsm_blk_kernel): needs the block inverses potf2f_body leaves
// behind the 64 reciprocal pivots
int launch_trsm_blk(bq_ctx *c, double *X, long ldx, long xstride, int m, const double *L11,
                    long ldl, long lstride, const double *dinv, long dstride, int batch)
{
    if (m <= 0)
        return BQ_OK;
    Bracket br(c, BQ_K_TRSM, 64.0 * 64 * (double)m * batch);
    hipLaunchKernelGGL(trsm_blk_kernel, dim3((m + 63) / 64, 1, batch), dim3(256), 0, c->cur, X, ldx,
                       xstride, m, L11, ldl, lstride, dinv, dstride);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_diag_winv(bq_ctx *c, const double *L, long ldl, int npad, double *dw)
{
    hipLaunchKernelGGL(diag_winv_kernel, dim3(npad / 64), dim3(256), 0, c->stream, L, ldl, dw);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// one step of a wide panel in one launch (panel_step_kernel, slab.h)
int launch_panel_step(bq_ctx *c, double *A, long lda, long astride, int batch, int nrb,
                      double *Sin, double *Sout, long lds, long sstride, int K0, int j0,
                      double *dinv_in, double *dinv_out, int has_next, int first, double *SL,
                      int *info, double work)
{
    Bracket br(c, BQ_K_GEMM, work);
    hipLaunchKernelGGL(panel_step_kernel, dim3(nrb, 1, batch), dim3(256), 0, c->cur, A, lda,
                       astride, Sin, Sout, lds, sstride, K0, j0, dinv_in, dinv_out,
                       (long)BQ_DINV_STRIDE, has_next, first, SL, info);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// the first diagonal factor of a slab sweep and the staging of panel 0 in one launch
int launch_slab_first(bq_ctx *c, double *A, long lda, long astride, int batch, double *S, long lds,
                      long sstride, int ntot, double *dinv, int *info, int col0, long dstride)
{
    Bracket br(c, BQ_K_POTF2, 64.0 * 64 * 64 / 3.0 * batch);
    hipLaunchKernelGGL(slab_first_kernel, dim3(ntot / 64, 1, batch), dim3(256), 0, c->cur, A, lda,
                       astride, S, lds, sstride, ntot, dinv, dstride, info, col0);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// one 64-column step of a small system in one launch (slab_step_kernel); stamps: the
// profiling instantiation (bq_probe_c2_timeline)
int launch_slab_step(bq_ctx *c, double *A, long lda, long astride, int batch, double *Sin,
                     double *Sout, long lds, long sstride, int ntot, int j0, double *dinv_in,
                     double *dinv_out, int fnext, int last, int *info, int col0,
                     long long *stamps, double work, long dstride)
{
    const int T = (ntot - j0 - 64) / 64;
    Bracket br(c, BQ_K_SYRK_SMALL, work);
    const long wgs = (long)T * (T + 1) / 2 * batch;
    const bool w8 = c->potf2_8w && fnext && wgs <= c->cus;
    if (stamps && w8)
        hipLaunchKernelGGL((slab_step_kernel<true, 8>), dim3(T * (T + 1) / 2, 1, batch), dim3(512), 0,
                           c->cur, A, lda, astride, Sin, Sout, lds, sstride, ntot, j0, dinv_in,
                           dinv_out, dstride, fnext, last, info, col0, stamps, c->slab_out);
    else if (stamps)
        hipLaunchKernelGGL((slab_step_kernel<true, 4>), dim3(T * (T + 1) / 2, 1, batch), dim3(256), 0,
                           c->cur, A, lda, astride, Sin, Sout, lds, sstride, ntot, j0, dinv_in,
                           dinv_out, dstride, fnext, last, info, col0, stamps, c->slab_out);
    else if (w8)
        // a CU per workgroup: 512 threads, the diagonal factor on eight waves
        hipLaunchKernelGGL((slab_step_kernel<false, 8>), dim3(T * (T + 1) / 2, 1, batch), dim3(512),
                           0, c->cur, A, lda, astride, Sin, Sout, lds, sstride, ntot, j0, dinv_in,
                           dinv_out, dstride, fnext, last, info, col0,
                           (long long *)nullptr, c->slab_out);
    else
        hipLaunchKernelGGL((slab_step_kernel<false, 4>), dim3(T * (T + 1) / 2, 1, batch), dim3(256),
                           0, c->cur, A, lda, astride, Sin, Sout, lds, sstride, ntot, j0, dinv_in,
                           dinv_out, dstride, fnext, last, info, col0,
                           (long long *)nullptr, c->slab_out);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

} // namespace bqh
