// slab.h -- one launch per 64-column step of a small factorisation (outer block 64)
// Part of the libbqhip.so kernel set; compiled into k_panel.hip (host.h lists the units).
#pragma once
#include "common.h"
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
__device__ __forceinline__ void slab_load16(const double *__restrict__ S, long lds,
                                            double (&t)[4][4])
{
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            t[c][r] = S[(long)(16 * c + 4 * r) * lds];
}

__device__ __forceinline__ void slab_solve16(const double (&t)[4][4], const double (&la)[4][3][4],
                                             const double (&wneg)[4][4], double4_t (&x)[4])
{
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

// The ten lower 16 x 16 blocks (rb, cb) of a diagonal tile dealt to the four waves:
//   wave 0: (0,0) (1,0) (1,1)   wave 1: (2,0) (2,1) (2,2)   wave 2: (3,0) (3,1)   wave 3: (3,2) (3,3)
// Every wave issues THREE MFMAs per k-step: waves 2 and 3 repeat their second block into an
// accumulator that is never stored.  (A third MFMA under `if (wave < 2)` was miscompiled into
// a wrong block (3, 1): an MFMA ignores EXEC, so it must never sit under anything but a scalar
// branch, and the redundant one costs nothing -- the other two waves issue three anyway.)
// doubles of LDS behind the tile for the fragments of L_jj (6 blocks) and W (4 blocks)
#define BQ_SLAB_FW (10 * 256)
#define DIAG_NB(w) ((w) < 2 ? 3 : 2)
#define DIAG_RB(w, s) ((w) == 0 ? ((s) > 0 ? 1 : 0) : ((w) == 1 ? 2 : 3))
#define DIAG_CB(w, s) ((w) == 0 ? ((s) == 2 ? 1 : 0) : (((w) == 3 ? 2 : 0) + ((w) >= 2 && (s) == 2 ? 1 : (s))))

// STAMP: a profiling instantiation (tools/c2_timeline.py) whose workgroup 0 records s_memtime
// at its phase boundaries; the shipped launches use STAMP = false and carry no stamp code.
// NW = 8 (launched when a step's workgroups have a CU each): 512 threads; waves 4-7 take no part
// in the tile update -- they pass its barriers -- and in workgroup 0 join the diagonal factor as
// its second wave per SIMD (potf2f_body<8>: the chain 16.5 k -> 13.2 k cycles).
template <bool STAMP, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void slab_step_kernel(double *__restrict__ A, long lda,
                                                        long astride,
                                                        const double *__restrict__ Sin,
                                                        double *__restrict__ Sout, long lds,
                                                        long sstride, int ntot, int j0,
                                                        const double *__restrict__ din,
                                                        double *__restrict__ dout, long dstride,
                                                        int factor_next, int last,
                                                        int *__restrict__ info, int col0,
                                                        long long *stamps, SlabOut out)
{
#define BQ_SSTAMP(k, dep)                                                                          \
    if (STAMP) {                                                                                   \
        asm volatile("" ::"v"(dep));                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0)                                \
            stamps[k] = (long long)__builtin_amdgcn_s_memtime();                                   \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
    BQ_SSTAMP(0, 0)
    // One region, 52 KiB -- a workgroup of this kernel must fit into what ONE retiring workgroup
    // of the 64-tile product frees on a CU (36 KiB + the 16 KiB four of them leave over), or it
    // starves behind a trailing update running on the other stream.  First 4096 doubles: the Q
    // rows of the tile, [k][row] (A-fragment reads are contiguous over rows); afterwards, in
    // workgroup 0, the updated diagonal block on its way to the factor and the factor's panel
    // slots.  Behind them: the fragments of L_jj and W (BQ_SLAB_FW doubles), later the factor's
    // diagonal sub-blocks.
    __shared__ __attribute__((aligned(16))) double plds[4096 + BQ_SLAB_FW];
    double *const Qs = plds;
    double *const Ts = plds; // the block sits where the factor's panel slots will be
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
    if (NW == 8 && wave >= 4) {
        // the barriers of the tile update below, in their order: fragments staged; Q rows
        // written; and for workgroup 0 with a next factor: Q rows read; diagonal block in LDS
        __syncthreads();
        __syncthreads();
        if (blockIdx.x == 0 && factor_next) {
            __syncthreads();
            __syncthreads();
            potf2_body<8>(A + r0 + (long)r0 * lda, lda, col0 + r0, dout, info + b, plds, Ts, 64,
                          (long long *)nullptr, out.scal ? out.scal + 4 * b + 1 : nullptr);
        }
        return;
    }

    // Every global load of the step is issued here, before the first MFMA: the fragments of
    // L_jj (blocks below its diagonal) and of the negated block inverses, the unsolved panel
    // rows of both row blocks and the C tile.  (Loaded where they are first used, the tile
    // waited behind the factor's stores, which may alias it, and the launch paid three
    // memory round trips one after the other.)
    double la[4][3][4], wneg[4][4], tp[4][4], tq[4][4];
    double4_t acc[4];
    {
        // The fragments of L_jj and W are the same for all four waves: the workgroup fetches
        // the six 16 x 16 blocks below L_jj's diagonal and the four block inverses ONCE into
        // LDS (10 doubles per thread; the region is the diagonal factor's, free until the
        // tile update is over) instead of 64 doubles per lane from L2 -- a CU takes in about
        // 40 B per cycle and the fragment loads were half of the launch's 190 KB.
        double *Fs = plds + 4096;          // block (c, bb), c > bb, at 256 ((c (c - 1)) / 2 + bb)
        double *Ws = plds + 4096 + 6 * 256; // W blocks as in global memory
        {
            const int t = threadIdx.x, ti = t & 15, tk = t >> 4;
            const double *L11 = A + j0 + (long)j0 * lda + ti + (long)tk * lda;
            double f[6], wv[4];
#pragma unroll
            for (int c = 1; c < 4; ++c)
#pragma unroll
                for (int bb = 0; bb < c; ++bb)
                    f[(c * (c - 1)) / 2 + bb] = L11[16 * c + (long)(16 * bb) * lda];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                wv[q] = din[64 + 256 * q + t];
            if (bx != by)
                slab_load16(Sin + Cb + 16 * wave + l15 + (long)l4 * lds, lds, tq);
            slab_load16(Sin + Rb + 16 * wave + l15 + (long)l4 * lds, lds, tp);
            // C tile (negated: the MFMAs add Q P^T).  Off-diagonal tiles: rows 16 wave .. +15,
            // four 16-column blocks.  Diagonal tiles need only their ten lower 16 x 16 blocks
            // and deal them 3 / 3 / 2 / 2 to the waves (DIAG_RB / DIAG_CB): 48 MFMAs on the
            // longest wave instead of 64 -- for workgroup (0, 0) this update sits between the
            // launch's start and the diagonal factor.
            if (bx == by) {
                acc[2] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int sb = 0; sb < 3; ++sb)
                    if (sb < DIAG_NB(wave)) {
                        const double *Cin = A + Rb + 16 * DIAG_RB(wave, sb) + l15 +
                                            (long)(Cb + 16 * DIAG_CB(wave, sb) + l4) * lda;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[sb][r] = -Cin[(long)(4 * r) * lda];
                    }
            } else {
                const double *Cin = A + Rb + 16 * wave + l15 + (long)(Cb + l4) * lda;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[cb][r] = -Cin[(long)(16 * cb + 4 * r) * lda];
            }
#pragma unroll
            for (int q = 0; q < 6; ++q)
                Fs[256 * q + t] = f[q];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Ws[256 * q + t] = -wv[q];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                wneg[c][r] = Ws[256 * c + 64 * r + l15 + 16 * l4];
#pragma unroll
        for (int c = 1; c < 4; ++c)
#pragma unroll
            for (int bb = 0; bb < c; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    la[c][bb][r] = Fs[256 * ((c * (c - 1)) / 2 + bb) + l15 + 16 * (l4 + 4 * r)];
    }
    BQ_SSTAMP(1, la[3][2][3] + wneg[3][3])
    // Q rows (row block by) -> LDS; P rows (row block bx) stay in registers
    double4_t xp[4];
    if (bx != by) {
        double4_t xq[4];
        slab_solve16(tq, la, wneg, xq);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Qs[(16 * c + l4 + 4 * r) * 64 + 16 * wave + l15] = xq[c][r];
    }
    slab_solve16(tp, la, wneg, xp);
    if (bx == by) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Qs[(16 * c + l4 + 4 * r) * 64 + 16 * wave + l15] = xp[c][r];
    }
    BQ_SSTAMP(2, xp[3][3])
    // the solved rows are the factor: tile column 0 owns the write
    if (by == 0) {
        double *Lw = A + Rb + 16 * wave + l15 + (long)(j0 + l4) * lda;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Lw[(long)(16 * c + 4 * r) * lda] = xp[c][r];
    }
    // D[n][i] = sum_k Q[n][k] P[i][k] (A operand = Q fragment from LDS, B operand = P as it
    // stands), C -= D^T
    __syncthreads();
    BQ_SSTAMP(3, acc[1][3] + acc[0][0])
    if (bx == by) {
        // both operands from LDS (Q = P here): block (rb, cb) += X_cb X_rb^T in D^T form
        const int rb0 = DIAG_RB(wave, 0), rb1 = DIAG_RB(wave, 1), rb2 = DIAG_RB(wave, 2);
        const int cb0 = DIAG_CB(wave, 0), cb1 = DIAG_CB(wave, 1), cb2 = DIAG_CB(wave, 2);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double *qrow = Qs + (16 * c + l4 + 4 * r) * 64 + l15;
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(qrow[16 * cb0], qrow[16 * rb0], acc[0],
                                                              0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(qrow[16 * cb1], qrow[16 * rb1], acc[1],
                                                              0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(qrow[16 * cb2], qrow[16 * rb2],
                                                              acc[2], 0, 0, 0);
            }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double pf = xp[c][r];
                const double *qrow = Qs + (16 * c + l4 + 4 * r) * 64 + l15;
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[cb] =
                        __builtin_amdgcn_mfma_f64_16x16x4f64(qrow[16 * cb], pf, acc[cb], 0, 0, 0);
            }
    }
    BQ_SSTAMP(4, acc[1][3] + acc[0][0])
    // tile column 0 is the next panel: it goes to the scratch column (the diagonal tile,
    // which workgroup 0 factors in place, and the Schur complement of the last step stay in A)
    if (blockIdx.x == 0 && factor_next) {
        // the next diagonal block never touches memory between its update and its factor
        // (only the ten lower blocks: the factor never reads a lane above its column's own);
        // it lands where the Q rows were: every wave must be through with them
        __syncthreads();
#pragma unroll
        for (int sb = 0; sb < 3; ++sb)
            if (sb < DIAG_NB(wave)) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Ts[16 * DIAG_RB(wave, sb) + l15 +
                       64 * (16 * DIAG_CB(wave, sb) + l4 + 4 * r)] = -acc[sb][r];
            }
        __syncthreads();
        // (col0: the global column of this sweep's first column, for the failure report)
        potf2_body<NW>(A + r0 + (long)r0 * lda, lda, col0 + r0, dout, info + b, plds, Ts, 64,
                       (STAMP && blockIdx.z == 0) ? stamps + 5 : nullptr,
                       out.scal ? out.scal + 4 * b + 1 : nullptr);
        return;
    }
    const bool to_s = by == 0 && bx > 0 && !last;
    double *Cout = to_s ? Sout + Rb + 16 * wave + l15 + (long)l4 * lds
                        : A + Rb + 16 * wave + l15 + (long)(Cb + l4) * lda;
    const long ldo = to_s ? lds : lda;
    // The last step's tiles are the Schur complement S of the border: with the read-out folded
    // in (SlabOut) the lane that holds S[c, c], S[yrow, c] or S[yrow, yrow] stores var_i, mean_i
    // and -- log|K| is complete since the previous launch -- the log-ML (reduce.h,
    // finalize_kernel, is the stand-alone form).  row / col: this sweep's numbering = the
    // system's (col0 = 0 for every caller that sets `out`).
#define BQ_SLAB_EMIT(ROW_, COL_, S_)                                                               \
    {                                                                                              \
        const int row_ = (ROW_), col_ = (COL_);                                                    \
        const double s_ = (S_);                                                                    \
        if (row_ == col_ && row_ >= out.npad && row_ < out.npad + out.M && out.var)                \
            out.var[(long)b * out.mstride + (row_ - out.npad)] = s_;                               \
        if (row_ == out.yrow) {                                                                    \
            if (col_ >= out.npad && col_ < out.npad + out.M && out.mean)                           \
                out.mean[(long)b * out.mstride + (col_ - out.npad)] = -s_;                         \
            if (col_ == out.yrow) {                                                                \
                const double qf_ = -s_, ld_ = out.scal[4 * b + 1];                                 \
                out.scal[4 * b + 2] = qf_;                                                         \
                out.scal[4 * b + 0] =                                                              \
                    -0.5 * qf_ - 0.5 * ld_ - 0.5 * (double)out.n * 1.8378770664093453;             \
            }                                                                                      \
        }                                                                                          \
    }
    const bool emit = last && out.scal != nullptr;
    if (bx == by) {
#pragma unroll
        for (int sb = 0; sb < 3; ++sb)
            if (sb < DIAG_NB(wave)) {
                double *Cd = A + Rb + 16 * DIAG_RB(wave, sb) + l15 +
                             (long)(Cb + 16 * DIAG_CB(wave, sb) + l4) * lda;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Cd[(long)(4 * r) * lda] = -acc[sb][r];
                if (emit) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        BQ_SLAB_EMIT(Rb + 16 * DIAG_RB(wave, sb) + l15,
                                     Cb + 16 * DIAG_CB(wave, sb) + l4 + 4 * r, -acc[sb][r])
                }
            }
    } else {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Cout[(long)(16 * cb + 4 * r) * ldo] = -acc[cb][r];
        if (emit) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    BQ_SLAB_EMIT(Rb + 16 * wave + l15, Cb + 16 * cb + l4 + 4 * r, -acc[cb][r])
        }
    }
#undef BQ_SLAB_EMIT
#undef BQ_SSTAMP
}

// The assembly of the bordered system with the first launch of the slab sweep folded in: the
// tiles of column block 0 also fill the sweep's scratch column, and the workgroup of tile
// (0, 0) clears the failure flag and factors the leading 64 x 64 block once its own stores are
// out -- one launch, one memset and 12 us less per pass of a small problem.
// NW = 8 (one problem or a few: the grid has a CU per workgroup): 512 threads, waves 4-7 only
// join the diagonal factor of tile (0, 0) (potf2f_body<8>).
template <int D, int NW = 4>
__global__ __launch_bounds__(64 * NW) void assemble_first_kernel(
    const double *__restrict__ pts, long pstride, const double *__restrict__ y, long ystride,
    const GaussParams *__restrict__ gp, int gpstride, double *__restrict__ A, long lda,
    long astride, Layout L, double *__restrict__ S0, long lds, long sstride,
    double *__restrict__ dinv, long dstride, int *__restrict__ info, double *__restrict__ scal)
{
    __shared__ __attribute__((aligned(16))) double plds[BQ_POTF2_LDS_DOUBLES];
    const int b = blockIdx.z;
    A += (long)b * astride;
    if (NW == 4 || threadIdx.x < 256)
        assemble_tile<D>(pts + (long)b * pstride, y + (long)b * ystride, gp[(long)b * gpstride],
                         A, lda, L, S0 + (long)b * sstride, lds);
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (threadIdx.x == 0) {
            info[b] = 0;
            if (scal)
                scal[4 * b + 1] = 0.0; // log|K|: the factors add to it (SlabOut)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // the block's 64 x 64 entries (rows 0..63 of this tile) are in memory
        __builtin_amdgcn_s_setprio(3);
        potf2_body<NW>(A, lda, 0, dinv + (long)b * dstride, info + b, plds, nullptr, 0, nullptr,
                       scal ? scal + 4 * b + 1 : nullptr, L.n);
    }
}

// first launch of a slab sweep: workgroup 0 factors the leading diagonal block, the others
// copy panel column 0 (rows >= 64) into the scratch column -- one launch instead of two
__global__ __launch_bounds__(256) void slab_first_kernel(double *__restrict__ A, long lda,
                                                         long astride, double *__restrict__ S,
                                                         long lds, long sstride, int ntot,
                                                         double *__restrict__ dinv, long dstride,
                                                         int *__restrict__ info, int col0)
{
    __shared__ __attribute__((aligned(16))) double plds[BQ_POTF2_LDS_DOUBLES];
    const int b = blockIdx.z;
    A += (long)b * astride;
    if (blockIdx.x == 0) {
        __builtin_amdgcn_s_setprio(3);
        potf2_body(A, lda, col0, dinv + (long)b * dstride, info + b, plds);
        return;
    }
    // 64 rows x 64 columns per workgroup: thread t copies row (t & 63) of 16 columns
    const int i = 64 * (int)blockIdx.x + (threadIdx.x & 63);
    const int jb = (threadIdx.x >> 6) * 16;
    if (i < ntot) {
        double *dst = S + (long)b * sstride + i + (long)jb * lds;
        const double *src = A + i + (long)jb * lda;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            dst[(long)j * lds] = src[(long)j * lda];
    }
}

// ---------------------------------------------------------------------------
// The same idea for the 64-column steps INSIDE a wide panel (outer block 128 / 256): one
// launch per step instead of two (left-looking slab update + fused diagonal factor, panel
// solve).  Step s of the panel that starts at column K0 (j0 = K0 + 64 s), one workgroup per
// 64-row block below the diagonal block of slab s:
//   * solve its own rows of slab s (from the scratch column Sin; the first step of a panel
//     reads them from A) and write them, the final L, to A;
//   * if the panel has a slab s + 1: solve the rows of row block s + 1 as well (every
//     workgroup for itself, through LDS) and bring its rows of slab s + 1 up to date with
//     ALL slabs 0 .. s of the panel (left-looking, k = 64 (s + 1): the earlier slabs' L from
//     A, slab s from registers / LDS); the result goes to the scratch column Sout, the input
//     of step s + 1 -- except in workgroup 0, whose rows ARE row block s + 1: its result is
//     the next diagonal block, handed through LDS to the 4-wave potf2.
// Nothing waits on another workgroup.  grid: ((ntot - j0 - 64) / 64, 1, batch); block 256.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void panel_step_kernel(double *__restrict__ A, long lda,
                                                         long astride,
                                                         const double *__restrict__ Sin,
                                                         double *__restrict__ Sout, long lds,
                                                         long sstride, int K0, int j0,
                                                         const double *__restrict__ din,
                                                         double *__restrict__ dout, long dstride,
                                                         int has_next, int first,
                                                         double *__restrict__ SL,
                                                         int *__restrict__ info)
{
    // 40 KiB (see slab_step_kernel): the solved Q rows, then -- workgroup 0 -- the next diagonal
    // block and the factor's slots in the same 4096 doubles
    __shared__ __attribute__((aligned(16))) double plds[BQ_POTF2_LDS_DOUBLES];
    double *const Qs = plds;
    double *const Ts = plds;
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    A += (long)b * astride;
    Sin += (long)b * sstride;
    Sout += (long)b * sstride;
    din += (long)b * dstride;
    dout += (long)b * dstride;
    const int r0 = j0 + 64;                 // row block s + 1 = the next diagonal block
    const int Rb = r0 + 64 * blockIdx.x;    // this workgroup's rows
    const int l15 = lane & 15, l4 = lane >> 4;
    SL += (long)b * 4096;
    // The first step of a panel reads its slab straight from A (no staging launch).  Every
    // workgroup reads its own rows before it overwrites them; the one block that others read
    // too -- row block s + 1, for their own solve -- is written by workgroup 0 to the side
    // buffer SL instead, and workgroup 0 of the NEXT step (where those rows are the diagonal
    // block's and nobody reads them) moves it into place.
    const double *Sp = first ? A + (long)j0 * lda : Sin;
    const long ldsp = first ? lda : lds;
    if (!first && j0 - K0 == 64 && blockIdx.x == 0) {
        double *dst = A + j0 + (long)(j0 - 64) * lda;
        for (int e = threadIdx.x; e < 4096; e += 256)
            dst[(e & 63) + (long)(e >> 6) * lda] = SL[e];
    }

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
    double4_t xp[4];
    if (has_next && blockIdx.x != 0) {
        double4_t xq[4];
        double tq[4][4];
        slab_load16(Sp + r0 + 16 * wave + l15 + (long)l4 * ldsp, ldsp, tq);
        slab_solve16(tq, la, wneg, xq);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Qs[(16 * c + l4 + 4 * r) * 64 + 16 * wave + l15] = xq[c][r];
    }
    {
        double tp[4][4];
        slab_load16(Sp + Rb + 16 * wave + l15 + (long)l4 * ldsp, ldsp, tp);
        slab_solve16(tp, la, wneg, xp);
    }
    {
        // (a lone workgroup has no readers to protect, and no next step to move the block)
        const bool side = first && has_next && blockIdx.x == 0 && gridDim.x > 1;
        double *Lw = side ? SL + 16 * wave + l15 + 64 * l4
                          : A + Rb + 16 * wave + l15 + (long)(j0 + l4) * lda;
        const long ldw = side ? 64 : lda;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Lw[(long)(16 * c + 4 * r) * ldw] = xp[c][r];
    }
    if (!has_next)
        return;
    if (blockIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Qs[(16 * c + l4 + 4 * r) * 64 + 16 * wave + l15] = xp[c][r];
    }
    // my rows of slab s + 1 (columns r0 .. r0 + 63), still as the last trailing update left them
    const double *Cin = A + Rb + 16 * wave + l15 + (long)(r0 + l4) * lda;
    double4_t acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[cb][r] = -Cin[(long)(16 * cb + 4 * r) * lda];
    // slabs 0 .. s-1 of the panel: both operands from A (P: my rows, Q: row block s + 1)
    {
        const double *Pg = A + Rb + 16 * wave + l15 + (long)(K0 + l4) * lda;
        const double *Qg = A + r0 + l15 + (long)(K0 + l4) * lda;
        // 16 columns (4 k-steps) per trip, the next trip's 20 fragments in flight meanwhile
        const int kend = j0 - K0; // multiple of 64
        double pa[4], qa[4][4], pb[4], qb[4][4];
#define BQ_PANEL_LOAD(PF, QF, KK)                                                                  \
    _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                  \
    {                                                                                              \
        PF[t] = Pg[(long)((KK) + 4 * t) * lda];                                                    \
        _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) QF[t][cb] =                               \
            Qg[16 * cb + (long)((KK) + 4 * t) * lda];                                              \
    }
#define BQ_PANEL_MFMA(PF, QF)                                                                      \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) \
        acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(QF[t][cb], PF[t], acc[cb], 0, 0, 0);
        if (kend > 0) {
            BQ_PANEL_LOAD(pa, qa, 0)
            for (int kk = 0; kk < kend; kk += 32) {
                BQ_PANEL_LOAD(pb, qb, kk + 16)
                __builtin_amdgcn_sched_barrier(0);
                BQ_PANEL_MFMA(pa, qa)
                __builtin_amdgcn_sched_barrier(0);
                const int kn = kk + 32 < kend ? kk + 32 : kk; // clamped: values unused past the end
                BQ_PANEL_LOAD(pa, qa, kn)
                __builtin_amdgcn_sched_barrier(0);
                BQ_PANEL_MFMA(pb, qb)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef BQ_PANEL_LOAD
#undef BQ_PANEL_MFMA
    }
    __syncthreads();
    // slab s: P from registers, Q from LDS
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
    if (blockIdx.x == 0) {
        __syncthreads(); // Ts is where the Q rows were
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ts[16 * wave + l15 + 64 * (16 * cb + l4 + 4 * r)] = -acc[cb][r];
        __syncthreads();
        potf2_body(A + r0 + (long)r0 * lda, lda, r0, dout, info + b, plds, Ts, 64);
        return;
    }
    double *Cout = Sout + Rb + 16 * wave + l15 + (long)l4 * lds;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Cout[(long)(16 * cb + 4 * r) * lds] = -acc[cb][r];
}

// ---------------------------------------------------------------------------
// The whole kb x kb diagonal block of an outer block factored by ONE workgroup per matrix
// (potrf.hip, enqueue_potrf_dfirst: a batch's diagonal factor D).  The one-launch steps above
// spend a chip on it -- every tile's workgroup solves the panel rows it needs itself, 253 VGPRs,
// hundreds of workgroups per step -- which is right for a chain that has the chip to itself and
// wrong for one that is supposed to hide beside the previous block's trailing update: there its
// workgroups displace the update's.  Here the factor of a matrix occupies one workgroup (eight
// waves) from its first column to its last:
//     per 64-column slab: the 64 x 64 factor (potf2f_body<8>, leaves its record behind);
//     the rows below, 16 per wave-task, on the matrix cores (slab_solve16);
//     the trailing tiles of the block, 32 x 32 per wave-task, k = 64 (all fragments requested
//     up front), lower triangle only;
// phases separated by workgroup barriers -- the block (1.6 MB at kb = 448) lives in L2, the same
// CU wrote what it reads.  64 matrices = 64 workgroups; ~20 us per slab.
// grid (batch), block 512.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wg_update32(double *__restrict__ C, long ldc,
                                            const double *__restrict__ P,
                                            const double *__restrict__ Q, long ldp, int row0,
                                            int col0, int lane)
{
    const int l15 = lane & 15, l4 = lane >> 4;
    double pa[16][2], qa[16][2], cold[2][2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const double *pp = P + row0 + 16 * h + l15 + (long)l4 * ldp;
        const double *qq = Q + col0 + 16 * h + l15 + (long)l4 * ldp;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            pa[ks][h] = pp[(long)(4 * ks) * ldp];
            qa[ks][h] = qq[(long)(4 * ks) * ldp];
        }
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                cold[tm][tn][rr] =
                    C[(row0 + 16 * tm + l15) + (long)(col0 + 16 * tn + l4 + 4 * rr) * ldc];
    __builtin_amdgcn_sched_barrier(0);
    double4_t acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                acc[tm][tn] =
                    __builtin_amdgcn_mfma_f64_16x16x4f64(qa[ks][tn], pa[ks][tm], acc[tm][tn], 0, 0, 0);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            if (col0 + 16 * tn >= row0 + 16 * tm + 16)
                continue; // (wave-uniform: a 16 x 16 block above the diagonal)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                C[(row0 + 16 * tm + l15) + (long)(col0 + 16 * tn + l4 + 4 * rr) * ldc] =
                    cold[tm][tn][rr] - acc[tm][tn][rr];
        }
}

__global__ __launch_bounds__(512) void potrf_wg_kernel(double *__restrict__ A, long lda,
                                                       long astride, int kb,
                                                       double *__restrict__ rec, long rstride,
                                                       int *__restrict__ info, int col0)
{
    __shared__ __attribute__((aligned(16))) double plds[BQ_POTF2_LDS_DOUBLES];
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x;
    A += (long)b * astride;
    rec += (long)b * rstride;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ns = kb >> 6;
    for (int s = 0; s < ns; ++s) {
        double *Ass = A + 64 * s + (long)(64 * s) * lda;
        double *rs = rec + (long)s * BQ_DINV_HALF;
        potf2_body<8>(Ass, lda, col0 + 64 * s, rs, info + b, plds);
        // (explicit: what this workgroup stored is in memory before its other waves read it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nr = ns - s - 1;
        if (nr == 0)
            break;
        const int r0 = 64 * (s + 1);
        {
            double la[4][3][4], wneg[4][4];
            const double *L11 = Ass + l15 + (long)l4 * lda;
            const double *W = rs + 64 + l15 + 16 * l4;
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
            for (int task = wave; task < 4 * nr; task += 8) {
                double *Xr = A + r0 + 16 * task + l15 + (long)(64 * s + l4) * lda;
                double t[4][4];
                double4_t x[4];
                slab_load16(Xr, lda, t);
                slab_solve16(t, la, wneg, x);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Xr[(long)(16 * c + 4 * r) * lda] = x[c][r];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        {
            const double *Pn = A + r0 + (long)(64 * s) * lda; // the solved panel, rows from r0
            double *Cn = A + r0 + (long)r0 * lda;
            const int ntask = 4 * (nr * (nr + 1) / 2);
            for (int task = wave; task < ntask; task += 8) {
                int ti, tj;
                tri_decode(task >> 2, ti, tj);
                const int q = task & 3;
                const int row0 = 64 * ti + 32 * (q & 1), cl0 = 64 * tj + 32 * (q >> 1);
                if (cl0 > row0)
                    continue; // the upper 32 x 32 block of a diagonal tile
                wg_update32(Cn, lda, Pn, Pn, lda, row0, cl0, lane);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}
