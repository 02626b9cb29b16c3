// slab.h -- one launch per 64-column step of a small factorisation (outer block 64)
// Part of the libbqhip.so kernel set; included through kernels.h.
#pragma once
#include "common.h"
#include "gemm.h"
#include "potf2.h"

// ---------------------------------------------------------------------------
// Small systems (one 64-column slab per outer step: C2, the reference's own problem
// sizes, the batched active-sampling systems) are a chain of dependent launches; every
// step used to be three of them -- trailing update (+ fused diagonal factor), panel
// solve -- and the chain, not the arithmetic, is the time.  Here a step is ONE launch and
// nothing in it waits on another workgroup:
//
//   the workgroup of trailing tile (bx, by) solves the panel rows it needs ITSELF --
//   row block bx and row block by, 64 x 64 each, on the matrix cores from the 16 x 16
//   block inverses potf2 left behind (the trsm_blk_kernel scheme, 40 MFMAs per wave) --,
//   updates its tile with them, and workgroup (0, 0) then factors the next diagonal block.
//
// The solve is recomputed by every workgroup that needs the rows (at most 2 x 80 MFMAs
// per wave, nothing next to a launch).  The unsolved panel cannot be overwritten in place
// while other workgroups still read it, so it lives in a scratch column: the step reads
// its panel from Sin and the workgroups of tile column 0 -- the next panel -- write their
// updated tiles to Sout (ping-pong) instead of A; workgroup (bx, 0) also writes the
// solved rows, the final L, to A.  The reciprocal pivots / block inverses ping-pong the
// same way (workgroup 0 writes the next block's while others may still read this one's).
//
//   A      ntot x ntot, column-major; L_jj at (j0, j0) factored by the previous launch
//   Sin    unsolved panel column j0 for rows >= j0 + 64, absolute row index, ld lds
//   Sout   receives the next panel column (rows >= j0 + 128) unless `last`
//   din    64 reciprocal pivots + 4 block inverses of L_jj; dout: the same for the next block
//   factor_next: workgroup (0, 0) factors the diagonal block at j0 + 64 after updating it
// grid: (T (T + 1) / 2, 1, batch), T = (ntot - j0 - 64) / 64; block 256.
// ---------------------------------------------------------------------------

// Solve 16 rows of the panel: returns X^T blocks x[c] (D / B-fragment form: x[c][r] of lane l
// is X[row l & 15][16 c + (l >> 4) + 4 r]).  la / w: A fragments of L_jj's off-diagonal
// blocks and of the (negated) block inverses, shared by every solve of the wave.
__device__ __forceinline__ void slab_solve16(const double *__restrict__ S, long lds,
                                             const double (&la)[4][3][4],
                                             const double (&wneg)[4][4], double4_t (&x)[4])
{
    double t[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            t[c][r] = S[(long)(16 * c + 4 * r) * lds];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double4_t acc = {-t[c][0], -t[c][1], -t[c][2], -t[c][3]};
#pragma unroll
        for (int bb = 0; bb < c; ++bb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(la[c][bb][r], x[bb][r], acc, 0, 0, 0);
        double4_t xc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r)
            xc = __builtin_amdgcn_mfma_f64_16x16x4f64(wneg[c][r], acc[r], xc, 0, 0, 0);
        x[c] = xc;
    }
}

__global__ __launch_bounds__(256) void slab_step_kernel(double *__restrict__ A, long lda,
                                                        long astride,
                                                        const double *__restrict__ Sin,
                                                        double *__restrict__ Sout, long lds,
                                                        long sstride, int ntot, int j0,
                                                        const double *__restrict__ din,
                                                        double *__restrict__ dout, long dstride,
                                                        int factor_next, int last,
                                                        int *__restrict__ info)
{
    // Q rows of the tile, [k][row] (A-fragment reads are contiguous over rows)
    __shared__ __attribute__((aligned(16))) double Qs[64 * 64];
    // workgroup 0: the updated diagonal block on its way to the factor, and the factor's ring
    __shared__ __attribute__((aligned(16))) double Ts[64 * 64];
    __shared__ __attribute__((aligned(16))) double ring[4 * 4 * 64 + 64];
    __shared__ int sbad[4];
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int bx, by;
    tri_decode(blockIdx.x, bx, by);
    A += (long)b * astride;
    Sin += (long)b * sstride;
    Sout += (long)b * sstride;
    din += (long)b * dstride;
    dout += (long)b * dstride;
    const int r0 = j0 + 64;
    const int Rb = r0 + 64 * bx, Cb = r0 + 64 * by;
    const int l15 = lane & 15, l4 = lane >> 4;

    // A fragments of L_jj (blocks below its diagonal) and of the negated block inverses
    double la[4][3][4], wneg[4][4];
    {
        const double *L11 = A + j0 + (long)j0 * lda + l15 + (long)l4 * lda;
        const double *W = din + 64 + l15 + 16 * l4;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                wneg[c][r] = -W[256 * c + 64 * r];
#pragma unroll
        for (int c = 1; c < 4; ++c)
#pragma unroll
            for (int bb = 0; bb < c; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    la[c][bb][r] = L11[16 * c + (long)(16 * bb + 4 * r) * lda];
    }
    // Q rows (row block by) -> LDS; P rows (row block bx) stay in registers
    double4_t xp[4];
    if (bx != by) {
        double4_t xq[4];
        slab_solve16(Sin + Cb + 16 * wave + l15 + (long)l4 * lds, lds, la, wneg, xq);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Qs[(16 * c + l4 + 4 * r) * 64 + 16 * wave + l15] = xq[c][r];
    }
    slab_solve16(Sin + Rb + 16 * wave + l15 + (long)l4 * lds, lds, la, wneg, xp);
    if (bx == by) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Qs[(16 * c + l4 + 4 * r) * 64 + 16 * wave + l15] = xp[c][r];
    }
    // the solved rows are the factor: tile column 0 owns the write
    if (by == 0) {
        double *Lw = A + Rb + 16 * wave + l15 + (long)(j0 + l4) * lda;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Lw[(long)(16 * c + 4 * r) * lda] = xp[c][r];
    }
    // C tile rows 16 wave .. +15, four 16-column blocks: D[n][i] = sum_k Q[n][k] P[i][k]
    // (A operand = Q fragment from LDS, B operand = P as it stands), C -= D^T
    const double *Cin = A + Rb + 16 * wave + l15 + (long)(Cb + l4) * lda;
    double4_t acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[cb][r] = -Cin[(long)(16 * cb + 4 * r) * lda];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double pf = xp[c][r];
            const double *qrow = Qs + (16 * c + l4 + 4 * r) * 64 + l15;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(qrow[16 * cb], pf, acc[cb], 0, 0, 0);
        }
    // tile column 0 is the next panel: it goes to the scratch column (the diagonal tile,
    // which workgroup 0 factors in place, and the Schur complement of the last step stay in A)
    if (blockIdx.x == 0 && factor_next) {
        // the next diagonal block never touches memory between its update and its factor
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ts[16 * wave + l15 + 64 * (16 * cb + l4 + 4 * r)] = -acc[cb][r];
        __syncthreads();
        potf2_64x4_body(A + r0 + (long)r0 * lda, lda, r0, dout, info + b, ring, sbad, Ts, 64);
        return;
    }
    const bool to_s = by == 0 && bx > 0 && !last;
    double *Cout = to_s ? Sout + Rb + 16 * wave + l15 + (long)l4 * lds
                        : A + Rb + 16 * wave + l15 + (long)(Cb + l4) * lda;
    const long ldo = to_s ? lds : lda;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        if (bx == by && cb > wave)
            continue; // above the diagonal of a diagonal tile
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Cout[(long)(16 * cb + 4 * r) * ldo] = -acc[cb][r];
    }
}

// panel column j0 of A (rows >= j0 + 64) -> scratch column, absolute row index
__global__ void slab_stage_kernel(const double *__restrict__ A, long lda, long astride,
                                  double *__restrict__ S, long lds, long sstride, int ntot, int j0)
{
    const int b = blockIdx.z;
    const int i = j0 + 64 + blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (i < ntot)
        S[(long)b * sstride + i + (long)j * lds] = A[(long)b * astride + i + (long)(j0 + j) * lda];
}
