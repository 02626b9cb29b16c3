// seed.h -- one entry of the bordered GP system, and a product's accumulators seeded with it
// Part of the libbqhip.so kernel set; included by gram.h (the assembly) and gemm.h (the products
// whose C the assembly left out: round 6).
#pragma once
#include "common.h"

// Entry (ii, j) of the bordered system from the two points (xi, xj; pi, pj: whether row / column
// carry a point at all): THE arithmetic of the assembly -- the seeded products (gram_seed_neg)
// call the same function, so a tile computed there has the bits of one stored here.
template <int D>
__device__ __forceinline__ double bordered_entry(int ii, int j, bool pi, bool pj,
                                                 const double (&xi)[D], const double (&xj)[D],
                                                 const GaussParams &g, const Layout &L,
                                                 const double *__restrict__ y)
{
    if (pi && pj) {
        double val = g.c * exp_gauss(gauss_q<D>(xi, xj, g));
        if (ii == j && ii < L.n)
            val += g.s2;
        return val;
    }
    if (ii == L.yrow)
        return (j < L.n) ? y[j] : 0.0;
    return (ii == j) ? 1.0 : 0.0;
}

// The accumulators of a (16 TM) x (16 TN) wave tile in the rotated-quad layout (gemm.h, Tile444)
// seeded with MINUS the bordered system's entries instead of minus a loaded C:
//   acc[tm][tn][s] = -A(sd.r + row0 + 16 tm + l15, sd.c + col0 + 16 tn + 4 ((blk - s) & 3) + l4).
template <int D, int TM, int TN>
__device__ __forceinline__ void gram_seed_neg(double (&acc)[TM][TN][4], const GramSeed &sd, int b,
                                              int row0, int col0, int lane)
{
    const int l15 = lane & 15, l4 = lane >> 4, blk = (lane >> 2) & 3;
    const double *__restrict__ pts = sd.pts + (long)b * sd.pstride;
    const double *__restrict__ y = sd.y + (long)b * sd.ystride;
    const GaussParams g = sd.gp[(long)b * sd.gpstride];
    const Layout L = sd.L;
    bool pi[TM];
    double xi[TM][D];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int ii = sd.r + row0 + 16 * tm + l15;
        pi[tm] = (ii < L.n) || (ii >= L.npad && ii < L.npad + L.M);
#pragma unroll
        for (int k = 0; k < D; ++k)
            xi[tm][k] = pi[tm] ? pts[k + (long)ii * D] : 0.0;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int j = sd.c + col0 + 16 * tn + 4 * ((blk - s) & 3) + l4;
            const bool pj = (j < L.n) || (j >= L.npad && j < L.npad + L.M);
            double xj[D];
#pragma unroll
            for (int k = 0; k < D; ++k)
                xj[k] = pj ? pts[k + (long)j * D] : 0.0;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                acc[tm][tn][s] = -bordered_entry<D>(sd.r + row0 + 16 * tm + l15, j, pi[tm], pj,
                                                    xi[tm], xj, g, L, y);
        }
}

