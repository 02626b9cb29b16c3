// k_gemm.hip -- the MFMA products C -= P Q^T (gemm.h) and their launchers.
#include "host.h"
#include "gemm.h"
#include "trsmsweep.h"

namespace bqh {

// function attributes of the LDS-staged kernel: 72 KiB of dynamic LDS per workgroup
int gemm_init(bq_ctx *c)
{
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_lds_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BQ_LDS_BYTES));
#define BQ_L64_ATTR(F_, B_)                                                                        \
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(F_),                              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, B_))
    BQ_L64_ATTR((gemm_lds64_kernel<false, 1>), BQ_L64_BYTES);
    BQ_L64_ATTR((gemm_lds64_kernel<true, 1>), BQ_L64_BYTES);
    BQ_L64_ATTR((rows_fused_kernel<false, 1>), BQ_L64_BYTES);
    BQ_L64_ATTR((rows_fused_kernel<true, 1>), BQ_L64_BYTES);
    BQ_L64_ATTR((rows_fused_kernel<false, 2>), BQ_L64_BYTES);
    BQ_L64_ATTR((rows_fused_kernel<true, 2>), BQ_L64_BYTES);
    BQ_L64_ATTR(gemm_trsm64_kernel, BQ_L64_BYTES);
    BQ_L64_ATTR(trsm_sweep_kernel, BQ_L64_BYTES);
    BQ_L64_ATTR(gemm_lds_seed_kernel<1>, BQ_LDS_BYTES);
    BQ_L64_ATTR(gemm_lds_seed_kernel<2>, BQ_LDS_BYTES);
    BQ_L64_ATTR(gemm_lds64_seed_kernel<1>, BQ_L64_BYTES);
    BQ_L64_ATTR(gemm_lds64_seed_kernel<2>, BQ_L64_BYTES);
#undef BQ_L64_ATTR
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
//
// Measured on MI355X with random operands and launches repeated for 150 ms (bq_probe_gemm; a
// product of zeros, or a single launch, runs at clocks a real sweep never sees -- it had the
// 64-tile ahead everywhere): a chip FULL of 128 x 128 tiles sustains about 1.1x the rate of one
// full of 64 x 64 tiles (k = 320, batch 100, m = 2816: 57.9 against 52.5 TFLOP/s; N = 8192
// alone: 59.7 / 57.3) -- half the LDS traffic per flop.  The 64-tile wins where the larger one
// cannot fill the chip (m = 4096 alone: 54.8 against 44.1) or pads ragged edges and the diagonal
// of a triangular update with more than that tenth (m = 1088 at batch 32: 48.8 / 46.3; m = 704
// at batch 128: 46.8 / 44.9).  Hence: 128 when there are at least 4.5 tiles per CU and they
// cover at most 1.12x the area of the 64-tiles; 64 when those give a workgroup per two CUs;
// else whatever fits.  That is the rule for a product that has the chip to itself
// (c->sharing == 0).  Otherwise:
//   * many small systems of a whole number of 128-tiles (m <= 512 at batch >= 64) stay with 128;
//   * while the two streams of a look-ahead share the chip (sharing == 1) every product with two
//     128-tiles per CU takes those.  Four 64-tile workgroups (88 VGPRs) fill a CU's register
//     file in quarters, and a retiring one never frees the 250+ VGPRs a panel_step_kernel
//     workgroup on the other stream needs: the panel chain starved until the update had drained
//     (N = 16384: 27.0 instead of 26.0 ms).  The 128-tile kernel holds a CU in halves;
//   * while the two halves of a batch share it (sharing == 2) the 64-tile goes first: one
//     half's update beside the other half's panel chain was faster in small workgroups whatever
//     its size (C5 shard 5.85 ms against 6.00 with the rule above and 6.17 with 128-tiles
//     first; 256 x C2: 4.68 / 4.79 / 4.85).
// BQ_GEMM_TILE=64|128 (read when a context is created) forces a tile where its kernel can run.
static int gemm_lds_tile(const bq_ctx *c, int m, int n, int k, int lower, int batch)
{
    if (!c->gemm_lds || (m % 64) || (n % 64) || (k % 32))
        return 0;
    const bool tri = lower && m == n;
    auto tiles = [&](int t) {
        const long gm = (m + t - 1) / t, gn = (n + t - 1) / t;
        const long per = tri ? gm * (gm + 1) / 2 : (lower ? gm * gn / 2 + 1 : gm * gn);
        return per * batch;
    };
    const long a = tiles(128), a64 = tiles(64);
    const bool can64 = c->gemm_lds64 && k >= 64, can128 = n >= 128;
    const int forced = c->gemm_tile;
    if (forced == 64 && can64)
        return 64;
    if (forced == 128 && can128)
        return 128;
    const bool small128 = tri && (m % 128) == 0 && m <= 512 && a >= 2L * c->cus;
    const bool full128 =
        c->sharing == 1 ? a >= 2L * c->cus
                        : c->sharing == 0 && 2 * a >= 9L * c->cus &&
                              4.0 * (double)a <= 1.12 * (double)a64;
    if (can128 && (small128 || full128))
        return 128;
    if (can64 && a64 >= c->cus / 2)
        return 64;
    return (can128 && a >= BQ_LDS_MIN_TILES) ? 128 : 0;
}

bool gemm_uses_lds(const bq_ctx *c, int m, int n, int k, int lower, int batch)
{
    return gemm_lds_tile(c, m, n, k, lower, batch) != 0;
}

int launch_gemm(bq_ctx *c, int cls, double *C, long ldc, long cstride, const double *P, long ldp,
                long pstride, const double *Q, long qsj, long qsk, long qstride, int m, int n,
                int k, int lower, int batch, int fuse_j0, double *dinv, long dstride, int *info,
                int ccut, const GramSeed *seed)
{
    // ccut > 0: columns >= ccut of C need no update (honoured by the LDS-staged kernel only)
    if (m <= 0 || n <= 0 || k <= 0)
        return BQ_OK;
    if ((m & 15) || (n & 15) || (k & 7))
        return fail(c, BQ_ERR_BAD_ARG, "gemm: m,n must be multiples of 16 and k of 8");
    if (seed) {
        // a product that cannot seed its own accumulators: its region of C is assembled first
        const bool f444s = qsj == 1 && (m % 64) == 0 && (n % 64) == 0;
        const int t = (f444s && fuse_j0 < 0) ? gemm_lds_tile(c, m, n, k, lower, batch) : 0;
        if (!((t == 128 || t == 64) && (seed->d == 1 || seed->d == 2))) {
            const int ncol = ccut > 0 ? std::min(n, ccut) : n;
            BQCHK(launch_assemble_region(c, *seed, C - seed->r - (long)seed->c * ldc, ldc, cstride, m,
                                         (ncol + 63) / 64 * 64, batch));
            seed = nullptr;
        }
    }
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
    // C left out of the assembly (seed): the LDS-staged kernels compute their tile of it from the
    // problem's points (d <= 2); anything else gets the region written first
    if (seed) {
        const bool can = (ldst == 128 || ldst == 64) && (seed->d == 1 || seed->d == 2);
        if (can) {
            const int cut = ccut > 0 ? ccut : 0x7fffffff;
            const dim3 g = grid_for(ldst);
#define BQ_GEMM_SEED(K_, BYTES_)                                                                   \
    hipLaunchKernelGGL(K_, g, dim3(256), BYTES_, c->cur, C, ldc, cstride, P, ldp, pstride, Q, qsk,  \
                       qstride, m, n, k, mode, cut, *seed)
            if (ldst == 128 && seed->d == 1)
                BQ_GEMM_SEED(gemm_lds_seed_kernel<1>, BQ_LDS_BYTES);
            else if (ldst == 128)
                BQ_GEMM_SEED(gemm_lds_seed_kernel<2>, BQ_LDS_BYTES);
            else if (seed->d == 1)
                BQ_GEMM_SEED(gemm_lds64_seed_kernel<1>, BQ_L64_BYTES);
            else
                BQ_GEMM_SEED(gemm_lds64_seed_kernel<2>, BQ_L64_BYTES);
#undef BQ_GEMM_SEED
            HIPCHK(c, hipGetLastError());
            return BQ_OK;
        }
    }
    if (ldst == 128) {
        dim3 g = grid_for(128);
        hipLaunchKernelGGL(gemm_lds_kernel, g, dim3(256), BQ_LDS_BYTES, c->cur, C, ldc, cstride, P,
                           ldp, pstride, Q, qsk, qstride, m, n, k, mode,
                           ccut > 0 ? ccut : 0x7fffffff);
    } else if (ldst == 64) {
        dim3 g = grid_for(64);
        hipLaunchKernelGGL((gemm_lds64_kernel<false, 1>), g, dim3(256), BQ_L64_BYTES, c->cur, C, ldc,
                           cstride, P, ldp, pstride, Q, qsk, qstride, m, n, k, mode,
                           ccut > 0 ? ccut : 0x7fffffff);
    } else if (qsk == 1 && (qsj & 1) == 0 && fuse_j0 < 0 && (m % 64) == 0 &&
               gemm_lds_tile(c, m, n, k, lower, batch) != 0 && c->gemm_lds64 &&
               tiles(64) >= c->cus / 2) {
        // Q given k-contiguous (the backward row sweep): the 64-tile kernel's transposed staging
        dim3 g = grid_for(64);
        hipLaunchKernelGGL((gemm_lds64_kernel<true, 1>), g, dim3(256), BQ_L64_BYTES, c->cur, C, ldc,
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

// C (m x n) -= P Q^T with the panel solve of C's first 64 columns fused in (gemm_trsm64_kernel):
// the batched panel solve's products (potrf.hip, enqueue_trsm_rec).  Lss / wrec: the factored
// diagonal block those 64 columns are solved against and its record of block inverses.
bool gemm_trsm_ok(const bq_ctx *c, int m, int n, int k)
{
    return c->gemm_lds64 && m > 0 && (m % 64) == 0 && n >= 64 && (n % 64) == 0 && k >= 32 &&
           (k % 32) == 0;
}

int launch_gemm_trsm(bq_ctx *c, double *C, long ldc, long cstride, const double *P, long ldp,
                     long pstride, const double *Q, long ldq, long qstride, int m, int n, int k,
                     const double *Lss, long ldl, long lstride, const double *wrec, long wstride,
                     int batch)
{
    if (!gemm_trsm_ok(c, m, n, k))
        return fail(c, BQ_ERR_BAD_ARG, "gemm_trsm: m, n multiples of 64 and k of 32");
    Bracket br(c, BQ_K_GEMM, (2.0 * (double)m * n * k + 64.0 * 64 * (double)m) * batch);
    hipLaunchKernelGGL(gemm_trsm64_kernel, dim3(m / 64, n / 64, batch), dim3(256), BQ_L64_BYTES,
                       c->cur, C, ldc, cstride, P, ldp, pstride, Q, ldq, qstride, m, n, k, Lss, ldl,
                       lstride, wrec, wstride);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// X (m x kb) <- X L11^-T for every row block in one launch (trsm_sweep_kernel)
int launch_trsm_sweep(bq_ctx *c, double *X, long ldx, long xstride, int m, const double *L11,
                      long ldl, long lstride, const double *rec, long rstride, int kb, int batch)
{
    if (m <= 0)
        return BQ_OK;
    if ((m & 63) || (kb & 63) || kb <= 0)
        return fail(c, BQ_ERR_BAD_ARG, "trsm_sweep: m and kb must be multiples of 64");
    Bracket br(c, BQ_K_TRSM, (double)m * kb * kb * batch);
    const int nrb = m / 64;
#ifdef BQ_TS_DBG
    // ablation build (make DEFS=-DBQ_TS_DBG OUT=../libbqhip_dbg.so; tools/r06_ablate.sh)
    static const int dbg = std::getenv("BQ_TS_DBG") ? std::atoi(std::getenv("BQ_TS_DBG")) : 0;
    hipLaunchKernelGGL(trsm_sweep_kernel, dim3(8 * nrb * ((batch + 7) / 8)), dim3(256), BQ_L64_BYTES,
                       c->cur, X, ldx, xstride, L11, ldl, lstride, rec, rstride, kb, nrb, batch, dbg);
#else
    hipLaunchKernelGGL(trsm_sweep_kernel, dim3(8 * nrb * ((batch + 7) / 8)), dim3(256), BQ_L64_BYTES,
                       c->cur, X, ldx, xstride, L11, ldl, lstride, rec, rstride, kb, nrb, batch);
#endif
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
    // a grid of at most two workgroups per CU: eight waves per workgroup.  The split-k job
    // tiles walk their k range in half the steps (the last step of an N = 4096 sweep, job tiles
    // only: 24 -> 16 us; a posterior variance at N = 1024: 0.067 -> 0.060 ms); the LDS tiles are
    // MFMA-bound either way
    const bool ks2 = c->gemm_ksplit && nd + nu <= 2 * c->cus && ((a.k1 + a.k2) % 128) == 0 &&
                     (nu == 0 || k >= 32);
#define BQ_ROWS_FUSED(QT_, KS_)                                                                    \
    hipLaunchKernelGGL((rows_fused_kernel<QT_, KS_>), dim3(nd + nu), dim3(256 * KS_),              \
                       BQ_L64_BYTES, c->cur, a, nd, ndx, C, ldc, P, ldp, Q, ldq, mrows, n, k)
    if (qt && ks2)
        BQ_ROWS_FUSED(true, 2);
    else if (qt)
        BQ_ROWS_FUSED(true, 1);
    else if (ks2)
        BQ_ROWS_FUSED(false, 2);
    else
        BQ_ROWS_FUSED(false, 1);
#undef BQ_ROWS_FUSED
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// one forward step of the row sweep over a resident factor in one launch (rows_step_kernel)
int launch_rows_step(bq_ctx *c, int mrows, const RowsJob &a, const RowsJob &b, double work)
{
    Bracket br(c, BQ_K_GEMM, work);
    const bool nw8 = c->gemm_ksplit && ((a.k1 + a.k2) % 128) == 0 &&
                     (b.ny == 0 || ((b.k1 + b.k2) % 128) == 0) &&
                     (long)(mrows / 32) * (a.ny + b.ny) <= 2L * c->cus;
    if (nw8)
        hipLaunchKernelGGL(rows_step_kernel<8>, dim3(mrows / 32, a.ny + b.ny), dim3(512), 0, c->cur,
                           a, b);
    else
        hipLaunchKernelGGL(rows_step_kernel<4>, dim3(mrows / 32, a.ny + b.ny), dim3(256), 0, c->cur,
                           a, b);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

} // namespace bqh
