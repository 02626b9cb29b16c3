// types.h -- plain structs and constants shared by the kernels and the host side of
// libbqhip.so (no device code: every translation unit includes it).
#pragma once
#include <stdint.h>

#define BQ_MAXD 8

// Per-problem scratch of the 64-column panel step: 64 reciprocal pivots of the current
// diagonal block, then the inverses of its four 16 x 16 diagonal sub-blocks (column-major,
// 256 doubles each) -- written by potf2f_body, read by trsm_blk_kernel.
#define BQ_DINV_HALF (64 + 4 * 256)
// two halves: the one-launch slab step (slab.h) writes the next block's half while this
// block's is still being read
#define BQ_DINV_STRIDE (2 * BQ_DINV_HALF)

// Gaussian kernel parameters of one batch element:
//   k(p,q) = c * exp( sum_k nh[k] (p_k - q_k)^2 ),  c = h^2 / prod(sqrt(2 pi) w_k),
//   nh[k] = -1 / (2 w_k^2);  s2 = s^2 is added on the diagonal of Kxx.
struct GaussParams {
    double c;
    double s2;
    double nh[BQ_MAXD];
};

// Layout of one bordered system (gram.h: assemble_kernel has the picture)
struct Layout {
    int n, npad, M, yrow, ntot; // yrow < 0: no y row
};

// A product C -= P Q^T whose C has not been written yet (round 6): instead of loading its tile of
// C the kernel computes the tile's entries of the bordered system -- assemble_tile's values, bit
// for bit -- from the problem's points.  r, c: the global row / column of C(0, 0) in that system.
struct GramSeed {
    const double *pts; // d x ntot per problem
    long pstride;
    const double *y;
    long ystride;
    const GaussParams *gp;
    int gpstride;
    Layout L;
    int d, r, c;
};

// Read-out of a bordered system folded into the one-launch sweep (slab.h): the diagonal factors
// add their share of log|K| to scal[4b + 1] as they go, and the LAST step's tiles -- the Schur
// complement of the border -- store what finalize_kernel would read from it (no launch of its own).
struct SlabOut {
    double *scal, *mean, *var; // scal: 4 per problem {logml, logdet, qf, -}; mean / var: mstride per problem
    long mstride;
    int n, npad, M, yrow;
};

// batched active-sampling systems (moments.h: assemble_esm_kernel)
struct EsmLayout {
    int ns, nsc, npad, ntot; // points [0, nsc] (nsc+1 of them), border rows npad, npad+1
};

// closed-form Gaussian integrals (moments.h): exp(logc - |linv (p - mu)|^2 / 2)
template <int D>
struct GaussForm {
    double mu[D];        // subtracted from the point(s) to form z
    double linv[D * D];  // row-major lower-triangular inverse Cholesky factor
    double logc;         // -(D log 2pi + log|C|) / 2
};

// one job of rows_step_kernel (gemm.h)
struct RowsJob {
    double *C;
    long ldc;
    const double *P1, *Q1, *P2, *Q2;
    long ldp1, qsj1, qsk1, ldp2, qsj2, qsk2;
    int k1, k2;
    int ny;    // tile columns of this job
    int write; // 1: C = -(products); 0: C -= products
};
