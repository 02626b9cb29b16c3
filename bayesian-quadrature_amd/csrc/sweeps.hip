// sweeps.hip -- sweeps over a RESIDENT Cholesky factor: the explicit inverses of its diagonal
// blocks, single-vector sweeps (GEMV form) and row-form sweeps for many right-hand sides.
// No kernels of its own (k_gemm.hip, k_panel.hip, k_reduce.hip).
#include "host.h"

namespace bqh {

// ---------------------------------------------------------------------------
// Row sweeps over a RESIDENT factor (cho_solve, alpha, posterior variance, the bordered
// acquisition update): X <- X L^-T (forward) and X <- X L^-1 (backward), right-hand sides as
// the rows of X (mrows x npad, ld ldx).
//
// With 64-column steps a sweep is 2 npad / 64 dependent launches and nothing else -- 2 ms at
// N = 4096 for 134 MB of factor.  The steps are therefore B = 256 or 512 columns wide and a
// step's triangular solve is a product with the explicit inverse of its B x B diagonal block
// (MAGMA's trtri-based trsm; the 16 x 16 inverses of the panel solve one level up):
//     forward   Y_J = X_J W_J^T,   X[:, J+B:] -= Y_J L[J+B:, J]^T
//     backward  Y_J = X_J W_J,     X[:, :J]   -= Y_J L[J, :J]
// -- two MFMA GEMM launches per B columns, input and output in two buffers so that no step
// copies.  NR = -W^T of every diagonal block (npad x B doubles, block J at NR + J B, ld B) is
// built once per factor by the 64-column sweep itself applied to -I, batched over the blocks.
// cond(W_J) <= cond(L) = sqrt(cond(K)): 1e-12 relative at the worst-conditioned configs.
// ---------------------------------------------------------------------------
WideInv wide_views(const double *base, int npad)
{
    WideInv w;
    const size_t n = wide_doubles(npad);
    w.nr = base;
    w.nt = base + n;
    w.tt = base + 2 * n;
    w.uu = base + 3 * n;
    w.t = base + 4 * n;
    w.B = wide_block(npad);
    return w;
}

// dw: the per-64-block records of diag_winv_kernel (npad / 64 of them)
int compute_wide_inverses(bq_ctx *c, const double *L, long ldl, int npad, const double *dw,
                          double *nr)
{
    const int B = wide_block(npad);
    HIPCHK(c, hipMemsetAsync(nr, 0, sizeof(double) * wide_doubles(npad), c->stream));
    BQCHK(launch_neg_identity(c, nr, B, npad));
    const int nfull = npad / B, rem = npad - nfull * B;
    for (int part = 0; part < 2; ++part) {
        const int batch = part == 0 ? nfull : (rem ? 1 : 0), bs = part == 0 ? B : rem;
        const int J0 = part == 0 ? 0 : nfull * B;
        if (batch == 0)
            continue;
        double *X = nr + (size_t)J0 * B; // bs x bs per block, ld B
        const double *Ld = L + J0 + (long)J0 * ldl;
        const long xs = (long)B * B, ls = (long)B * (1 + ldl), ds = (long)(B / 64) * BQ_DINV_HALF;
        for (int jb = 0; jb < bs; jb += 64) {
            const double *L11 = Ld + jb + (long)jb * ldl;
            BQCHK(launch_trsm_blk(c, X + (long)jb * B, B, xs, bs, L11, ldl, ls,
                                  dw + (long)((J0 + jb) / 64) * BQ_DINV_HALF, ds, batch));
            const int rest = bs - jb - 64;
            if (rest > 0)
                BQCHK(launch_gemm(c, BQ_K_GEMM, X + (long)(jb + 64) * B, B, xs, X + (long)jb * B, B,
                                  xs, L11 + 64, 1, ldl, ls, bs, rest, 64, 0, batch));
        }
    }
    // NT = the blocks' transposes, behind NR
    double *nt = nr + wide_doubles(npad), *tt = nt + wide_doubles(npad),
           *uu = tt + wide_doubles(npad);
    for (int part = 0; part < 2; ++part) {
        const int batch = part == 0 ? nfull : (rem ? 1 : 0), bs = part == 0 ? B : rem;
        if (batch == 0)
            continue;
        const size_t off = part == 0 ? 0 : (size_t)nfull * B * B;
        BQCHK(launch_transpose_blocks(c, nr + off, nt + off, B, (long)B * B, bs / 64, bs / 64, batch));
    }
    // The couplings of neighbouring blocks for the one-launch steps of trsv.h:
    //   T_J = W_J L[J, J-B]  (bJ x B; kept transposed)   and   U_J = L[J+B, J] W_J  (bn x B).
    const int nblk = nfull + (rem ? 1 : 0);
    if (nblk > 1) {
        // T before its transposition: a full B x B per block, behind UU
        double *tmp = uu + wide_doubles(npad);
        HIPCHK(c, hipMemsetAsync(tmp, 0, sizeof(double) * (size_t)nblk * B * B, c->stream));
        HIPCHK(c, hipMemsetAsync(uu, 0, sizeof(double) * wide_doubles(npad), c->stream));
        const long bb = (long)B * B, ls = (long)B * (1 + ldl);
        // full blocks J = B .. (nfull - 1) B, then the partial last one
        if (nfull > 1)
            BQCHK(launch_gemm(c, BQ_K_GEMM, tmp + bb, B, bb, nt + bb, B, bb, L + B, ldl, 1, ls, B, B,
                              B, 0, nfull - 1));
        if (rem) {
            const long J = (long)nfull * B;
            BQCHK(launch_gemm(c, BQ_K_GEMM, tmp + nfull * bb, B, 0, nt + J * B, B, 0,
                              L + J + (J - B) * ldl, ldl, 1, 0, rem, B, rem, 0, 1));
        }
        for (int part = 0; part < 2; ++part) {
            const int batch = part == 0 ? nfull - 1 : (rem ? 1 : 0), bs = part == 0 ? B : rem;
            if (batch <= 0)
                continue;
            const size_t off = (part == 0 ? 1 : (size_t)nfull) * bb;
            BQCHK(launch_transpose_blocks(c, tmp + off, tt + off, B, bb, bs / 64, B / 64, batch));
        }
        // U_J for the blocks with a full neighbour below, then the one above the partial block
        if (nfull > 1)
            BQCHK(launch_gemm(c, BQ_K_GEMM, uu, B, bb, L + B, ldl, ls, nr, 1, B, bb, B, B, B, 0,
                              nfull - 1));
        if (rem) {
            const long J = (long)(nfull - 1) * B;
            BQCHK(launch_gemm(c, BQ_K_GEMM, uu + J * B, B, 0, L + J + B + J * ldl, ldl, 0,
                              nr + J * B, 1, B, 0, rem, B, B, 0, 1));
        }
    }
    return BQ_OK;
}

// One right-hand side: x (npad, consumed) -> y = L^-1 x, one launch per B columns (trsv.h)
int enqueue_forward_vec(bq_ctx *c, double *x, double *y, const double *L, long ldl, int npad,
                        WideInv w, double *ws)
{
    if (ws && trsv_flow_ok(c, npad, w.B))
        return launch_trsv_flow(c, true, L, ldl, npad, w.B, w.nr, w.tt, x, y, ws);
    for (int J = 0; J < npad; J += w.B) {
        const int bJ = std::min(w.B, npad - J);
        const int nupd = J > 0 ? (npad - J - bJ) / 64 : 0;
        BQCHK(launch_trsv_fwd(c, L, ldl, J, bJ, w.B, nupd, w.nr + (size_t)J * w.B,
                              w.tt + (size_t)J * w.B, x, y,
                              (double)bJ * bJ + 2.0 * w.B * (J > 0 ? bJ + 64.0 * nupd : 0)));
    }
    return BQ_OK;
}

// x (npad, consumed) -> y = L^-T x
int enqueue_backward_vec(bq_ctx *c, double *x, double *y, const double *L, long ldl, int npad,
                         WideInv w, double *ws)
{
    if (ws && trsv_flow_ok(c, npad, w.B))
        return launch_trsv_flow(c, false, L, ldl, npad, w.B, w.nt, w.uu, x, y, ws);
    const int last = (npad - 1) / w.B * w.B;
    for (int J = last; J >= 0; J -= w.B) {
        const int bJ = std::min(w.B, npad - J);
        const int bn = J < last ? std::min(w.B, npad - J - w.B) : 0;
        const int nupd = bn > 0 ? J / 64 : 0;
        BQCHK(launch_trsv_bwd(c, L, ldl, J, bJ, w.B, bn, nupd, w.nt + (size_t)J * w.B,
                              w.uu + (size_t)J * w.B, x, y,
                              (double)bJ * bJ + 2.0 * bn * (bJ + 64.0 * nupd)));
    }
    return BQ_OK;
}

// X <- X L^-T in place, 64 columns per step from the 16 x 16 block inverses (the panel solve
// of the factorisation): twice the launches of the wide steps, but every product is with the
// inverse of a 16 x 16 block only.  For right-hand sides that nearly lie in the span of the
// factor's own columns -- the borders of the acquisition update, whose Schur complement
// k0 - |L^-1 k|^2 cancels to 1e-7 -- the 64- to 512-wide explicit inverses lose cond(L_JJ)
// (3e-10 against 2e-12 on test_acquisition_and_posterior_vs_extended_precision).
int enqueue_forward_rows_blk(bq_ctx *c, double *X, long ldx, int mrows, const double *L, long ldl,
                             int npad, const double *dw)
{
    for (int jb = 0; jb < npad; jb += 64) {
        const double *L11 = L + jb + (long)jb * ldl;
        BQCHK(launch_trsm_blk(c, X + (long)jb * ldx, ldx, 0, mrows, L11, ldl, 0,
                              dw + (long)(jb / 64) * BQ_DINV_HALF, 0, 1));
        const int rest = npad - jb - 64;
        if (rest > 0)
            BQCHK(launch_gemm(c, BQ_K_GEMM, X + (long)(jb + 64) * ldx, ldx, 0, X + (long)jb * ldx,
                              ldx, 0, L11 + 64, 1, ldl, 0, mrows, rest, 64, 0, 1));
    }
    return BQ_OK;
}

// whether the one-launch steps of rows_step_kernel (32 x 32 split-k tiles for everything) serve
// a row sweep: systems too small to give 64 x 64 LDS-staged tiles half a chip of workgroups
static bool rows_small(const bq_ctx *c, int mrows, int npad, const WideInv &w)
{
    // (below 2048 rows the split-k tiles stay ahead: the posterior at 1000 points over N = 1024
    // took 77 us with them, 90 us with the fused large-system steps -- round 3, 256-column steps;
    // 66 us with round 5's 512-column steps)
    return (mrows % 32) == 0 && (w.B % 64) == 0 &&
           (long)(mrows / 32) * (npad / 32) <= 4L * c->cus &&
           ((mrows % 64) != 0 || !c->gemm_lds64 || npad < 2048 ||
            (long)(mrows / 64) * (npad / 64) < c->cus / 2);
}

// A large system's sweep, one launch per step (rows_fused_kernel).  With T_J = W_J L[J, J-B]
// (forward; U_J = L[J+B, J] W_J backward) a step's small product -- the diagonal block's solve,
//     Y_J = X_J W_J^T - Y_{J-B} T_J^T,
// latency-bound -- and its big one -- Y_{J-B} applied to everything beyond block J, MFMA-bound --
// depend on the PREVIOUS step only, not on each other: they share a launch, and a step costs
// the longer of the two instead of their sum plus a launch gap.  (On two streams with events
// the small product waited for workgroup slots behind the big one: slower than in sequence.)
static int rows_fused(bq_ctx *c, bool forward, double *Xin, double *Xout, long ldx, int mrows,
                      const double *L, long ldl, int npad, const WideInv &w)
{
    const int last = (npad - 1) / w.B * w.B;
    for (int step = 0; step * w.B < npad; ++step) {
        const int J = forward ? step * w.B : last - step * w.B;
        const int bJ = std::min(w.B, npad - J);
        const int Jp = forward ? J - w.B : J + w.B; // the block solved one step earlier
        const bool first = step == 0;
        const int bp = first ? 0 : std::min(w.B, npad - Jp);
        RowsJob a{};
        a.C = Xout + (long)J * ldx;
        a.ldc = ldx;
        a.P1 = Xin + (long)J * ldx;
        a.ldp1 = ldx;
        a.Q1 = (forward ? w.nt : w.nr) + (size_t)J * w.B;
        a.qsj1 = 1;
        a.qsk1 = w.B;
        a.k1 = bJ;
        a.ny = bJ / 32;
        a.write = 1;
        // (an unused second pair points at valid memory: k2 = 0 never dereferences it)
        a.P2 = a.P1, a.Q2 = a.Q1, a.ldp2 = ldx, a.qsj2 = 1, a.qsk2 = w.B;
        double *Cu = Xin;
        const double *Pu = Xin, *Qu = L;
        long ldq = ldl;
        int nu = 0;
        if (!first) {
            a.P2 = Xout + (long)Jp * ldx;
            a.k2 = bp;
            Pu = a.P2;
            if (forward) {
                a.Q2 = w.t + (size_t)J * w.B; // T_J (bJ x B), rows contiguous
                nu = npad - J - bJ;           // Xin[:, J + bJ:] -= Y_Jp L[J + bJ:, Jp ..)^T
                Cu = Xin + (long)(J + bJ) * ldx;
                Qu = L + J + bJ + (long)Jp * ldl;
            } else {
                a.Q2 = w.uu + (size_t)J * w.B; // U_J (bp x B): Q2(j, k) = U_J[k, j]
                a.qsj2 = w.B;
                a.qsk2 = 1;
                nu = J;                        // Xin[:, 0:J] -= Z_Jp L[Jp .., 0:J]  (Q k-contiguous)
                Qu = L + Jp;
            }
        }
        // The last steps' updates are few 64 x 64 tiles of one k loop each (22 us however few):
        // there the update goes out as 32 x 32 split-k tiles too (rows_step_kernel), a quarter of
        // a tile's k loop per wave.  rows_tail: from how many LDS tiles down in the forward sweep
        // (0: never); the backward sweep's Q is strided across a tile's columns there and gains
        // only from a quarter of that on (N = 4096, 256 rows, step times in us, forward 30 26 26 25
        // 17 -> 29 23 21 17 16, backward 28 26 26 25 19 -> 28 26 26 19 19; tools/rows_tail_check.py).
        const long lds_tiles = (long)(mrows / 64) * (nu / 64);
        if (!first && nu > 0 && c->rows_tail > 0 &&
            lds_tiles <= (forward ? c->rows_tail : c->rows_tail / 4) &&
            (mrows % 32) == 0 && (nu % 32) == 0 && (bp % 16) == 0) {
            RowsJob b = a;
            b.C = Cu;
            b.P1 = Pu;
            b.ldp1 = ldx;
            b.Q1 = Qu;
            b.qsj1 = forward ? 1 : ldq;
            b.qsk1 = forward ? ldq : 1;
            b.k1 = bp;
            b.k2 = 0;
            b.P2 = b.P1, b.Q2 = b.Q1, b.ldp2 = ldx, b.qsj2 = b.qsj1, b.qsk2 = b.qsk1;
            b.ny = nu / 32;
            b.write = 0;
            BQCHK(launch_rows_step(c, mrows, a, b,
                                   2.0 * mrows * ((double)bJ * (a.k1 + a.k2) + (double)nu * bp)));
            continue;
        }
        BQCHK(launch_rows_fused(c, mrows, a, Cu, ldx, Pu, ldx, Qu, ldq, nu, bp, !forward,
                                2.0 * mrows * ((double)bJ * (a.k1 + a.k2) + (double)nu * bp)));
    }
    return BQ_OK;
}

// Xout <- Xin L^-T; Xin is overwritten with partial sums
int enqueue_forward_rows(bq_ctx *c, double *Xin, double *Xout, long ldx, int mrows,
                         const double *L, long ldl, int npad, WideInv w)
{
    // small systems: one launch per step (rows_step_kernel), every entry of Xout written
    if (rows_small(c, mrows, npad, w)) {
        for (int J = 0; J < npad; J += w.B) {
            const int bJ = std::min(w.B, npad - J), rest = npad - J - bJ;
            RowsJob a{}, b{};
            a.C = Xout + (long)J * ldx;
            a.ldc = ldx;
            a.P1 = Xin + (long)J * ldx;
            a.ldp1 = ldx;
            a.Q1 = w.nt + (size_t)J * w.B;
            a.qsj1 = 1;
            a.qsk1 = w.B;
            a.k1 = bJ;
            a.ny = bJ / 32;
            a.write = 1;
            // (unused operand pairs point at valid memory: k2 = 0 never dereferences them)
            a.P2 = a.P1, a.Q2 = a.Q1, a.ldp2 = ldx, a.qsj2 = 1, a.qsk2 = w.B;
            b = a;
            b.ny = 0;
            // Block 1 has no update pending when the sweep starts (the coupling to block 0 is
            // folded into T_1), so its own product X_1 W_1^T rides in step 0's launch -- which has
            // no update job of its own -- and step 1 only subtracts Y_0 T_1^T: half the dependent
            // k-steps on the chain (posterior at N = 1024, M = 256: 27 -> 23 us of sweep)
            if (J == 0 && rest > 0) {
                const int b1 = std::min(w.B, rest);
                b.C = Xout + (long)w.B * ldx;
                b.P1 = Xin + (long)w.B * ldx;
                b.Q1 = w.nt + (size_t)w.B * w.B;
                b.k1 = b1;
                b.ny = b1 / 32;
                b.P2 = b.P1, b.Q2 = b.Q1;
            }
            if (J > 0) {
                if (J == w.B) {
                    a.P1 = Xout + (long)(J - w.B) * ldx;
                    a.Q1 = w.t + (size_t)J * w.B;
                    a.k1 = w.B;
                    a.write = 0;
                    a.P2 = a.P1, a.Q2 = a.Q1;
                } else {
                    a.P2 = Xout + (long)(J - w.B) * ldx;
                    a.Q2 = w.t + (size_t)J * w.B;
                    a.k2 = w.B;
                }
                if (rest > 0) {
                    b.C = Xin + (long)(J + bJ) * ldx;
                    b.P1 = a.P2;
                    b.Q1 = L + J + bJ + (long)(J - w.B) * ldl;
                    b.qsj1 = 1;
                    b.qsk1 = ldl;
                    b.k1 = w.B;
                    b.k2 = 0;
                    b.ny = rest / 32;
                    b.write = 0;
                }
            }
            BQCHK(launch_rows_step(
                c, mrows, a, b,
                2.0 * mrows * ((double)bJ * (a.k1 + a.k2) + (double)b.ny * 32 * b.k1)));
        }
        return BQ_OK;
    }
    if (c->gemm_lds64 && (mrows % 64) == 0 && (w.B % 64) == 0 && (ldl & 1) == 0)
        return rows_fused(c, true, Xin, Xout, ldx, mrows, L, ldl, npad, w);
    HIPCHK(c, hipMemsetAsync(Xout, 0, sizeof(double) * (size_t)ldx * npad, c->cur));
    for (int J = 0; J < npad; J += w.B) {
        const int bJ = std::min(w.B, npad - J);
        // (NR[k, j] read through its transposed copy: unit stride across the output columns)
        BQCHK(launch_gemm_rows(c, BQ_K_TRSM, Xout + (long)J * ldx, ldx, Xin + (long)J * ldx, ldx,
                               w.nt + (size_t)J * w.B, 1, w.B, mrows, bJ, bJ));
        const int rest = npad - J - bJ;
        if (rest > 0)
            BQCHK(launch_gemm_rows(c, BQ_K_GEMM, Xin + (long)(J + bJ) * ldx, ldx,
                                   Xout + (long)J * ldx, ldx, L + J + bJ + (long)J * ldl, 1, ldl,
                                   mrows, rest, bJ));
    }
    return BQ_OK;
}

// Xout <- Xin L^-1 (the L^T sweep of dpotrs in row form); Xin is overwritten
int enqueue_backward_rows(bq_ctx *c, double *Xin, double *Xout, long ldx, int mrows,
                          const double *L, long ldl, int npad, WideInv w)
{
    if (!rows_small(c, mrows, npad, w) && c->gemm_lds64 && (mrows % 64) == 0 &&
        (w.B % 64) == 0 && (ldl & 1) == 0)
        return rows_fused(c, false, Xin, Xout, ldx, mrows, L, ldl, npad, w);
    HIPCHK(c, hipMemsetAsync(Xout, 0, sizeof(double) * (size_t)ldx * npad, c->cur));
    const int last = (npad - 1) / w.B * w.B;
    for (int J = last; J >= 0; J -= w.B) {
        const int bJ = std::min(w.B, npad - J);
        BQCHK(launch_gemm_rows(c, BQ_K_TRSM, Xout + (long)J * ldx, ldx, Xin + (long)J * ldx, ldx,
                               w.nr + (size_t)J * w.B, 1, w.B, mrows, bJ, bJ));
        if (J > 0) // Xin[:, 0:J] -= Xout[:, J:J+bJ] L[J:J+bJ, 0:J]
            BQCHK(launch_gemm_rows(c, BQ_K_GEMM, Xin, ldx, Xout + (long)J * ldx, ldx, L + J, ldl, 1,
                                   mrows, J, bJ));
    }
    return BQ_OK;
}

// (L L^T) X = B for nrhs host columns through the row-form sweeps: the columns go up as they
// are, a device transposition puts them into the rows of the sweep buffers (mpad x npad,
// X_dev[r, j] = B[j, r]) and the solution back
int solve_rows_host(bq_ctx *c, const double *L, long ldl, int n, int npad, WideInv w,
                    const double *B, int64_t nrhs, double *X)
{
    const int mpad = (int)roundup(nrhs, 64);
    DevBuf Bd, Xd, X2;
    HIPCHK(c, Bd.alloc(sizeof(double) * (size_t)n * nrhs));
    HIPCHK(c, Xd.alloc(sizeof(double) * (size_t)mpad * npad));
    HIPCHK(c, X2.alloc(sizeof(double) * (size_t)mpad * npad));
    HIPCHK(c, hipMemcpyAsync(Bd.p, B, Bd.bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(Xd.p, 0, Xd.bytes, c->stream));
    BQCHK(launch_transpose_pad(c, Bd.d(), (long)n, n, (int)nrhs, Xd.d(), (long)mpad));
    BQCHK(enqueue_forward_rows(c, Xd.d(), X2.d(), mpad, mpad, L, ldl, npad, w));
    BQCHK(enqueue_backward_rows(c, X2.d(), Xd.d(), mpad, mpad, L, ldl, npad, w));
    BQCHK(launch_transpose_pad(c, Xd.d(), (long)mpad, (int)nrhs, n, Bd.d(), (long)n));
    HIPCHK(c, hipMemcpyAsync(X, Bd.p, Bd.bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

} // namespace bqh
