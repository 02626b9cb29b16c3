// gram.h -- Gaussian-kernel Gram matrices and the bordered GP system (HBM-write bound)
// Part of the libbqhip.so kernel set; compiled into k_gram.hip / k_panel.hip (host.h lists the units).
#pragma once
#include "common.h"
#include "seed.h"

// ---------------------------------------------------------------------------
// Full symmetric Gram, K[i,j] = k(x_i,x_j) + s2 [i==j], any n (the path for sizes that are
// not whole 64 x 64 blocks; gram_tri_kernel below otherwise).
// Block = 256 threads = a 128(i) x 64(j) tile: lane pairs two consecutive rows
// (one 16-byte store), a wave stores 1 KiB of one column per instruction, the
// four waves take 16 columns each.  x is d x n.
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void gram_sym_kernel(const double *__restrict__ x, long xstride,
                                                       const GaussParams *__restrict__ gp,
                                                       int gpstride, double *__restrict__ K,
                                                       long ldk, long kstride, int n)
{
    const int b = blockIdx.z;
    x += (long)b * xstride;
    K += (long)b * kstride;
    const GaussParams g = gp[(long)b * gpstride];
    const int t = threadIdx.x;
    const int i = blockIdx.x * 128 + (t & 63) * 2;
    const int jbase = blockIdx.y * 64 + (t >> 6) * 16;
    if (i >= n)
        return;
    const bool two = (i + 1 < n);
    double xi0[D], xi1[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        xi0[k] = x[k + (long)i * D];
        xi1[k] = two ? x[k + (long)(i + 1) * D] : 0.0;
    }
    const bool vec = two && ((ldk & 1) == 0);
#pragma unroll 4
    for (int jj = 0; jj < 16; ++jj) {
        const int j = jbase + jj;
        if (j >= n)
            break;
        double xj[D];
#pragma unroll
        for (int k = 0; k < D; ++k)
            xj[k] = x[k + (long)j * D];
        double v0 = g.c * exp_gauss(gauss_q<D>(xi0, xj, g));
        double v1 = g.c * exp_gauss(gauss_q<D>(xi1, xj, g));
        if (i == j)
            v0 += g.s2;
        if (i + 1 == j)
            v1 += g.s2;
        double *dst = K + i + (long)j * ldk;
        if (vec) {
            double2_t v = {v0, v1};
            *reinterpret_cast<double2_t *>(dst) = v;
        } else {
            dst[0] = v0;
            if (two)
                dst[1] = v1;
        }
    }
}

// ---------------------------------------------------------------------------
// The same matrix with every exp evaluated ONCE: a workgroup owns the 64 x 64 block (bi, bj),
// bi >= bj, of the lower triangle and stores it and its transpose.  A lane holds 2 x 2
// micro-tiles -- the two orientations of a micro-tile are both 16-byte pairs, so neither store
// needs a shuffle -- and the 64 lanes of a wave form an 8 x 8 patch: a store instruction
// writes 8 full 128-byte lines in either orientation (8 lanes x 16 B along the contiguous
// direction, 8 columns).  n % 64 == 0, ldk even.  grid (T (T + 1) / 2, 1, batch), T = n / 64.
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void gram_tri_kernel(const double *__restrict__ x, long xstride,
                                                       const GaussParams *__restrict__ gp,
                                                       int gpstride, double *__restrict__ K,
                                                       long ldk, long kstride, int n)
{
    const int b = blockIdx.z;
    x += (long)b * xstride;
    K += (long)b * kstride;
    const GaussParams g = gp[(long)b * gpstride];
    int bi, bj;
    tri_decode(blockIdx.x, bi, bj);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 7, lj = lane >> 3;
    // wave w sweeps the 16-column strip w of the block in four 16-row steps
    const int j0 = 64 * bj + 16 * wave + 2 * lj;
    double xj0[D], xj1[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        xj0[k] = x[k + (long)j0 * D];
        xj1[k] = x[k + (long)(j0 + 1) * D];
    }
#pragma unroll
    for (int si = 0; si < 4; ++si) {
        const int i0 = 64 * bi + 16 * si + 2 * li;
        double xi0[D], xi1[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            xi0[k] = x[k + (long)i0 * D];
            xi1[k] = x[k + (long)(i0 + 1) * D];
        }
        double v00 = g.c * exp_gauss(gauss_q<D>(xi0, xj0, g));
        double v10 = g.c * exp_gauss(gauss_q<D>(xi1, xj0, g));
        double v01 = g.c * exp_gauss(gauss_q<D>(xi0, xj1, g));
        double v11 = g.c * exp_gauss(gauss_q<D>(xi1, xj1, g));
        if (i0 == j0) {
            v00 += g.s2;
            v11 += g.s2;
        }
        const double2_t c0 = {v00, v10}, c1 = {v01, v11};
        *reinterpret_cast<double2_t *>(K + i0 + (long)j0 * ldk) = c0;
        *reinterpret_cast<double2_t *>(K + i0 + (long)(j0 + 1) * ldk) = c1;
        if (bi != bj) {
            const double2_t r0 = {v00, v01}, r1 = {v10, v11};
            *reinterpret_cast<double2_t *>(K + j0 + (long)i0 * ldk) = r0;
            *reinterpret_cast<double2_t *>(K + j0 + (long)(i0 + 1) * ldk) = r1;
        }
    }
}

// Rectangular Gram K[i,j] = k(x1_i, x2_j), n1 x n2, ld = ldk.
template <int D>
__global__ __launch_bounds__(256) void gram_cross_kernel(const double *__restrict__ x1, int n1,
                                                         const double *__restrict__ x2, int n2,
                                                         GaussParams g, double *__restrict__ K,
                                                         long ldk)
{
    const int t = threadIdx.x;
    const int i = blockIdx.x * 64 + (t & 63);
    const int jbase = blockIdx.y * 64 + (t >> 6) * 16;
    if (i >= n1)
        return;
    double xi[D];
#pragma unroll
    for (int k = 0; k < D; ++k)
        xi[k] = x1[k + (long)i * D];
    for (int jj = 0; jj < 16; ++jj) {
        const int j = jbase + jj;
        if (j >= n2)
            break;
        double xj[D];
#pragma unroll
        for (int k = 0; k < D; ++k)
            xj[k] = x2[k + (long)j * D];
        K[i + (long)j * ldk] = g.c * exp_gauss(gauss_q<D>(xi, xj, g));
    }
}

// The same with the zero padding written by the kernel itself: K is n1p x n2p (multiples of 64),
// rows >= n1 and columns >= n2 are zeros -- the right-hand sides of a posterior's forward sweep
// (fit.hip, bq_gp_predict) without a memset in front.  x1 may be mapped host memory (the
// prediction points straight out of the caller's staging buffer: every workgroup reads its 64
// points once).  Real entries: the arithmetic of gram_cross_kernel.  grid (n1p / 64, n2p / 64).
template <int D>
__global__ __launch_bounds__(256) void gram_cross_pad_kernel(const double *__restrict__ x1, int n1,
                                                             const double *__restrict__ x2, int n2,
                                                             GaussParams g, double *__restrict__ K,
                                                             long ldk)
{
    const int t = threadIdx.x;
    const int i = blockIdx.x * 64 + (t & 63);
    const int jbase = blockIdx.y * 64 + (t >> 6) * 16;
    const bool row = i < n1;
    double xi[D];
#pragma unroll
    for (int k = 0; k < D; ++k)
        xi[k] = row ? x1[k + (long)i * D] : 0.0;
    // (the kernel is latency, not work: a thread's points of the second set are requested in
    // batches of eight -- clamped at the end, so that no load sits under a branch -- before the
    // first exp that needs one; the sixteen exps are independent of one another)
#pragma unroll
    for (int h8 = 0; h8 < 16; h8 += 8) {
        double xj[8][D];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = min(jbase + h8 + jj, n2 - 1);
#pragma unroll
            for (int k = 0; k < D; ++k)
                xj[jj][k] = x2[k + (long)j * D];
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = jbase + h8 + jj;
            const double e = g.c * exp_gauss(gauss_q<D>(xi, xj[jj], g));
            K[i + (long)j * ldk] = (row && j < n2) ? e : 0.0;
        }
    }
}

// ---------------------------------------------------------------------------
// Bordered GP system, lower triangle only (tiles strictly above the diagonal
// are skipped).  Index space of size ntot (multiple of 64):
//   [0, n)              samples x            -> Kxx + s2 I
//   [n, npad)           identity padding     -> delta_ij
//   [npad, npad + M)    prediction points xo -> K(xo, x), K(xo, xo)
//   yrow = npad + M     (if has_y) the row y^T, zero elsewhere, zero diagonal
//   (yrow, ntot)        identity padding
// pts is d x ntot with x at [0,n) and xo at [npad, npad+M); other columns are
// never read.  After eliminating the first npad columns, the Schur complement
// holds the posterior covariance, -mean in row yrow and -y'K^-1 y at
// (yrow, yrow); see DESIGN.md section 2.
// ---------------------------------------------------------------------------
// (struct Layout: types.h)

// One 128 x 64 tile of the bordered system (pts, y, A: this problem's).  S0 != nullptr: the
// tile's entries of rows >= 64 of column block 0 -- the unsolved first panel of the one-launch
// slab sweep -- go to the sweep's scratch column as well (slab.h).
// ib, jb: the tile's first row / column (default: from the block index); ilim, jlim: rows /
// columns from there on are not written (a region of the system, assemble_region_kernel)
template <int D>
__device__ __forceinline__ void assemble_tile(const double *__restrict__ pts,
                                              const double *__restrict__ y, const GaussParams &g,
                                              double *__restrict__ A, long lda, const Layout &L,
                                              double *__restrict__ S0, long lds, int ib = -1,
                                              int jb = -1, int ilim = 0x7fffffff,
                                              int jlim = 0x7fffffff)
{
    const int t = threadIdx.x;
    if (ib < 0) {
        ib = blockIdx.x * 128;
        jb = blockIdx.y * 64;
    }
    if (jb > ib + 127) // whole tile strictly above the diagonal
        return;
    const int i = ib + (t & 63) * 2;
    const int jbase = jb + (t >> 6) * 16;
    if (i >= L.ntot || i >= ilim)
        return;
    bool pi[2];
    double xi[2][D];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ii = i + r;
        pi[r] = (ii < L.n) || (ii >= L.npad && ii < L.npad + L.M);
#pragma unroll
        for (int k = 0; k < D; ++k)
            xi[r][k] = pi[r] ? pts[k + (long)ii * D] : 0.0;
    }
    for (int jj = 0; jj < 16; ++jj) {
        const int j = jbase + jj;
        if (j >= L.ntot || j >= jlim)
            break;
        const bool pj = (j < L.n) || (j >= L.npad && j < L.npad + L.M);
        double xj[D];
#pragma unroll
        for (int k = 0; k < D; ++k)
            xj[k] = pj ? pts[k + (long)j * D] : 0.0;
        double v[2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
            v[r] = bordered_entry<D>(i + r, j, pi[r], pj, xi[r], xj, g, L, y);
        double2_t vv = {v[0], v[1]};
        *reinterpret_cast<double2_t *>(A + i + (long)j * lda) = vv; // ntot, lda even
        if (S0 && jb == 0 && i >= 64)
            *reinterpret_cast<double2_t *>(S0 + i + (long)j * lds) = vv;
    }
}

template <int D>
__global__ __launch_bounds__(256) void assemble_kernel(const double *__restrict__ pts,
                                                       long pstride, const double *__restrict__ y,
                                                       long ystride,
                                                       const GaussParams *__restrict__ gp,
                                                       int gpstride, double *__restrict__ A,
                                                       long lda, long astride, Layout L)
{
    const int b = blockIdx.z;
    assemble_tile<D>(pts + (long)b * pstride, y + (long)b * ystride, gp[(long)b * gpstride],
                     A + (long)b * astride, lda, L, nullptr, 0);
}

// Rows [r, r + m) x columns [c, c + n) of the bordered system (lower part: tiles strictly above
// the diagonal skipped), for a product whose C was left out of the assembly and whose kernel
// cannot seed its accumulators itself (launch_gemm).  r, c, m, n multiples of 64.
// grid (ceil(m / 128), n / 64, batch).
template <int D>
__global__ __launch_bounds__(256) void assemble_region_kernel(const double *__restrict__ pts,
                                                              long pstride,
                                                              const double *__restrict__ y,
                                                              long ystride,
                                                              const GaussParams *__restrict__ gp,
                                                              int gpstride, double *__restrict__ A,
                                                              long lda, long astride, Layout L, int r,
                                                              int c, int m, int n)
{
    const int b = blockIdx.z;
    assemble_tile<D>(pts + (long)b * pstride, y + (long)b * ystride, gp[(long)b * gpstride],
                     A + (long)b * astride, lda, L, nullptr, 0, r + (int)blockIdx.x * 128,
                     c + (int)blockIdx.y * 64, r + m, c + n);
}
