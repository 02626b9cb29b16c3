// k_gemm.hip -- the MFMA products C -= P Q^T (gemm.h) and their launchers.
#include "host.h"
#include "gemm.h"

namespace bqh {

// function attributes of the LDS-staged kernel: 72 KiB of dynamic LDS per workgroup
int gemm_init(bq_ctx *c)
{
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_lds_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BQ_LDS_BYTES));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_lds64_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BQ_L64_BYTES));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_lds64_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BQ_L64_BYTES));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(rows_fused_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BQ_L64_BYTES));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(rows_fused_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BQ_L64_BYTES));
    return BQ_OK;
}

// C(m x n) -= P(m x k) Q(n x k)^T; tile shape from the amount of parallelism
// fuse_j0 >= 0: also factor the leading 64x64 block of C (global column fuse_j0) in the
// same launch (see gemm_sub_kernel); dinv / info as for launch_potf2
// whether C(m x n) -= P Q^T with unit-stride Q rows goes to the LDS-staged 128 x 128 kernel:
// whole 64 x 64 wave tiles, k in chunks of 32, and at least BQ_LDS_MIN_TILES workgroup tiles.
// (Round 1 asked for a full chip of tiles, 256.  The look-ahead's update of the next panel --
// m x 512 columns, 100-250 tiles, on the second stream BESIDE the bulk update -- then went to
// the register-streaming kernel at ~12 TFLOP/s and sat on the panel chain: with the LDS
// kernel N = 16384 takes 26.5 instead of 27.1 ms, 12288 13.17 instead of 13.35; the smaller
// sizes and the batched configs do not move.)
// Such an update never carries the fused diagonal factor (the factor would ride on the
// register-streaming kernel, which is slower by more than a potf2 launch costs).
#define BQ_LDS_MIN_TILES 96
// 0: no; 128 / 64: the workgroup tile of the LDS-staged kernel that takes the product.
// 128 x 128 tiles when there are at least BQ_LDS_MIN_TILES of them; products that cannot fill
// the chip with those -- a few hundred rows against a long operand (the row sweeps), the late
// updates of a batch -- take the 64 x 64 form (gemm_lds64_kernel) when that gives at least a
// workgroup per CU.
static int gemm_lds_tile(const bq_ctx *c, int m, int n, int k, int lower, int batch)
{
    if (!c->gemm_lds || (m % 64) || (n % 64) || (k % 32))
        return 0;
    long a = (long)((m + 127) / 128) * ((n + 127) / 128) * batch;
    long a64 = (long)(m / 64) * (n / 64) * batch;
    if (lower) {
        a = a / 2 + 1;
        a64 = a64 / 2 + 1;
    }
    if (n >= 128 && a >= 2L * c->cus)
        return 128;
    if (c->gemm_lds64 && a64 >= c->cus / 2 && k >= 64)
        return 64;
    return (n >= 128 && a >= BQ_LDS_MIN_TILES) ? 128 : 0;
}

bool gemm_uses_lds(const bq_ctx *c, int m, int n, int k, int lower, int batch)
{
    return gemm_lds_tile(c, m, n, k, lower, batch) != 0;
}

int launch_gemm(bq_ctx *c, int cls, double *C, long ldc, long cstride, const double *P, long ldp,
                long pstride, const double *Q, long qsj, long qsk, long qstride, int m, int n,
                int k, int lower, int batch, int fuse_j0, double *dinv, long dstride, int *info,
                int ccut)
{
    // ccut > 0: columns >= ccut of C need no update (honoured by the LDS-staged kernel only)
    if (m <= 0 || n <= 0 || k <= 0)
        return BQ_OK;
    if ((m & 15) || (n & 15) || (k & 7))
        return fail(c, BQ_ERR_BAD_ARG, "gemm: m,n must be multiples of 16 and k of 8");
    auto tiles = [&](int t) {
        long a = (long)((m + t - 1) / t) * ((n + t - 1) / t) * batch;
        return lower ? a / 2 + 1 : a;
    };
    // algorithmic flops: full product 2mnk; lower trapezoid of a trailing block
    // 2k(mn - n^2/2), i.e. m^2 k for the square update
    const double flops = (lower ? 2.0 * k * ((double)m * n - 0.5 * (double)n * n)
                                : 2.0 * (double)m * n * k) * batch;
    if (cls == BQ_K_SYRK && tiles(128) < c->cus)
        cls = BQ_K_SYRK_SMALL;
    Bracket br(c, cls, flops);
    const long cu = c->cus;
    // square trailing updates launch only their lower workgroup tiles (mode 2)
    const bool tri = lower && m == n;
    const int mode = tri ? 2 : lower;
    auto grid_for = [&](int t) {
        const unsigned gm = (unsigned)((m + t - 1) / t), gn = (unsigned)((n + t - 1) / t);
        return tri ? dim3(gm * (gm + 1) / 2, 1, batch) : dim3(gm, gn, batch);
    };
    // the 4x4x4 four-block MFMA sustains ~1.5x the rate of the 16x16x4 form on gfx950; it
    // needs unit-stride Q rows and whole wave tiles (every padded system here has them)
    const bool f444 = qsj == 1 && (m % 64) == 0 && (n % 64) == 0;
#define BQ_GEMM_SUB(TM_, TN_, T_)                                                                  \
    do {                                                                                           \
        if (f444)                                                                                  \
            hipLaunchKernelGGL((gemm_sub_kernel<TM_, TN_, 1>), grid_for(T_), dim3(256), 0, c->cur, \
                               C, ldc, cstride, P, ldp, pstride, Q, qsj, qsk, qstride, m, n, k,    \
                               mode, fuse_j0, dinv, dstride, info);                                \
        else                                                                                       \
            hipLaunchKernelGGL((gemm_sub_kernel<TM_, TN_, 0>), grid_for(T_), dim3(256), 0, c->cur, \
                               C, ldc, cstride, P, ldp, pstride, Q, qsj, qsk, qstride, m, n, k,    \
                               mode, fuse_j0, dinv, dstride, info);                                \
    } while (0)
    // a 64-column slab has no use for 128-column workgroup tiles (half of their waves idle)
    const int ldst = (f444 && fuse_j0 < 0) ? gemm_lds_tile(c, m, n, k, lower, batch) : 0;
    if (ldst == 128) {
        dim3 g = grid_for(128);
        hipLaunchKernelGGL(gemm_lds_kernel, g, dim3(256), BQ_LDS_BYTES, c->cur, C, ldc, cstride, P,
                           ldp, pstride, Q, qsk, qstride, m, n, k, mode,
                           ccut > 0 ? ccut : 0x7fffffff);
    } else if (ldst == 64) {
        dim3 g = grid_for(64);
        hipLaunchKernelGGL(gemm_lds64_kernel<false>, g, dim3(256), BQ_L64_BYTES, c->cur, C, ldc,
                           cstride, P, ldp, pstride, Q, qsk, qstride, m, n, k, mode,
                           ccut > 0 ? ccut : 0x7fffffff);
    } else if (qsk == 1 && (qsj & 1) == 0 && fuse_j0 < 0 && (m % 64) == 0 &&
               gemm_lds_tile(c, m, n, k, lower, batch) != 0 && c->gemm_lds64 &&
               tiles(64) >= c->cus / 2) {
        // Q given k-contiguous (the backward row sweep): the 64-tile kernel's transposed staging
        dim3 g = grid_for(64);
        hipLaunchKernelGGL(gemm_lds64_kernel<true>, g, dim3(256), BQ_L64_BYTES, c->cur, C, ldc,
                           cstride, P, ldp, pstride, Q, qsj, qstride, m, n, k, mode,
                           ccut > 0 ? ccut : 0x7fffffff);
    } else if (tiles(128) >= cu && n >= 128) {
        BQ_GEMM_SUB(4, 4, 128);
    } else if (tiles(64) >= cu / 2) {
        if (k == 64)
            hipLaunchKernelGGL((gemm_k64_kernel<2, 2>), grid_for(64), dim3(256), 0, c->cur, C, ldc,
                               cstride, P, ldp, pstride, Q, qsj, qsk, qstride, m, n, mode, fuse_j0,
                               dinv, dstride, info);
        else
            BQ_GEMM_SUB(2, 2, 64);
    } else {
        if (k == 64)
            hipLaunchKernelGGL((gemm_k64_kernel<1, 1>), grid_for(32), dim3(256), 0, c->cur, C, ldc,
                               cstride, P, ldp, pstride, Q, qsj, qsk, qstride, m, n, mode, fuse_j0,
                               dinv, dstride, info);
        else
            BQ_GEMM_SUB(1, 1, 32);
    }
#undef BQ_GEMM_SUB
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// the sweeps' products (few rows, a long k).  A kernel of their own with eight k-steps of
// fragment loads in flight (instead of gemm_sub_kernel's one) was measured and gained nothing:
// 36 us per launch at k = 512 either way -- the factor panel streams from HBM behind one
// block of prefetch, not from L2.
int launch_gemm_rows(bq_ctx *c, int cls, double *C, long ldc, const double *P, long ldp,
                     const double *Q, long qsj, long qsk, int m, int n, int k)
{
    // small products (posterior variance at C2 size): split-k tiles, gemm_splitk_kernel --
    // unless the 64 x 64 LDS-staged tiles already get half a chip of workgroups
    const bool lds = (qsj == 1 || (qsk == 1 && (qsj & 1) == 0)) && (m % 64) == 0 &&
                     gemm_lds_tile(c, m, n, k, 0, 1) != 0;
    if (!lds && (m % 32) == 0 && (n % 32) == 0 && (k % 64) == 0 && k <= 2048 &&
        (long)(m / 32) * (n / 32) <= 4L * c->cus) {
        Bracket br(c, cls, 2.0 * (double)m * n * k);
        hipLaunchKernelGGL(gemm_splitk_kernel, dim3(m / 32, n / 32), dim3(256), 0, c->cur, C, ldc,
                           P, ldp, Q, qsj, qsk, k);
        HIPCHK(c, hipGetLastError());
        return BQ_OK;
    }
    return launch_gemm(c, cls, C, ldc, 0, P, ldp, 0, Q, qsj, qsk, 0, m, n, k, 0, 1);
}

// one step of the row sweep over a large resident factor in one launch (rows_fused_kernel):
// job a in 32 x 32 split-k tiles + C(m x n) -= P(m x k) Q^T in 64 x 64 LDS-staged tiles
// (qt: Q k-contiguous, Q(j, k) at Q[j ldq + k]; else Q(j, k) at Q[j + k ldq])
int launch_rows_fused(bq_ctx *c, int mrows, const RowsJob &a, double *C, long ldc, const double *P,
                      long ldp, const double *Q, long ldq, int n, int k, bool qt, double work)
{
    Bracket br(c, BQ_K_GEMM, work);
    const int ndx = mrows / 32, nd = ndx * a.ny;
    const int nu = (n > 0 && k > 0) ? (mrows / 64) * (n / 64) : 0;
    if (qt)
        hipLaunchKernelGGL(rows_fused_kernel<true>, dim3(nd + nu), dim3(256), BQ_L64_BYTES, c->cur,
                           a, nd, ndx, C, ldc, P, ldp, Q, ldq, mrows, n, k);
    else
        hipLaunchKernelGGL(rows_fused_kernel<false>, dim3(nd + nu), dim3(256), BQ_L64_BYTES, c->cur,
                           a, nd, ndx, C, ldc, P, ldp, Q, ldq, mrows, n, k);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// one forward step of the row sweep over a resident factor in one launch (rows_step_kernel)
int launch_rows_step(bq_ctx *c, int mrows, const RowsJob &a, const RowsJob &b, double work)
{
    Bracket br(c, BQ_K_GEMM, work);
    hipLaunchKernelGGL(rows_step_kernel, dim3(mrows / 32, a.ny + b.ny), dim3(256), 0, c->cur, a, b);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

} // namespace bqh
