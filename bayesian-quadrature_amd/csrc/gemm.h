// gemm.h -- C -= P Q^T on v_mfma_f64_16x16x4_f64 (panel and trailing updates)
// Part of the libbqhip.so kernel set; compiled into k_gemm.hip (host.h lists the units).
#pragma once
#include "common.h"
#include "potf2.h"
#include "seed.h"


// one wave tile of C -= P Q^T (see gemm_sub_kernel); C, P, Q already point at the batch element
template <int TM, int TN>
__device__ __forceinline__ void gemm_sub_tile(double *__restrict__ C, long ldc,
                                              const double *__restrict__ P, long ldp,
                                              const double *__restrict__ Q, long qsj, long qsk,
                                              int m, int n, int k, int lower, int row0, int col0,
                                              int lane)
{
    const int l15 = lane & 15, l4 = lane >> 4;

    // clamp fragment rows at the edge (m, n multiples of 16 but maybe not of
    // the wave tile): out-of-range MFMA tiles are computed on clamped rows and
    // dropped at the store.
    const double *pp[TM];
    const double *qq[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        int r = row0 + tm * 16;
        if (r >= m)
            r = row0;
        pp[tm] = P + r + l15 + (long)l4 * ldp;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        int c = col0 + tn * 16;
        if (c >= n)
            c = col0;
        qq[tn] = Q + (long)(c + l15) * qsj + (long)l4 * qsk;
    }

    double4_t acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = (double4_t){0.0, 0.0, 0.0, 0.0};

    double pa[TM], qa[TN], pb[TM], qb[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
        pa[tm] = pp[tm][0];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
        qa[tn] = qq[tn][0];
    const long pstep = 4 * ldp, qstep = 4 * qsk;
    const int ksteps = k >> 2; // even (k is a multiple of 8): the body below has no branch
    for (int ks = 0; ks < ksteps; ks += 2) {
        // fragments of step ks+1 are requested before the MFMAs of step ks issue,
        // those of step ks+2 before the MFMAs of step ks+1.  The scheduling
        // barriers keep that order: without them the scheduler sinks each load
        // group down to its first use and the prefetch distance collapses to zero.
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
            pb[tm] = pp[tm][(long)(ks + 1) * pstep];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            qb[tn] = qq[tn][(long)(ks + 1) * qstep];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] =
                    __builtin_amdgcn_mfma_f64_16x16x4f64(qa[tn], pa[tm], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const long o2 = (ks + 2 < ksteps) ? (long)(ks + 2) : (long)ks; // clamped, value unused
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
            pa[tm] = pp[tm][o2 * pstep];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            qa[tn] = qq[tn][o2 * qstep];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] =
                    __builtin_amdgcn_mfma_f64_16x16x4f64(qb[tn], pb[tm], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    // D^T tile: D[jj][ii], jj = l4 + 4 r (column of C), ii = l15 (row of C)
    // Two passes, all loads before any store: written as `*dst -= acc` the compiler must
    // assume the store of one element aliases the load of the next and serialises 64
    // memory round trips per lane (measured: 14 % of the trailing update).
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int r = row0 + tm * 16;
            if (r >= m)
                continue;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int c = col0 + tn * 16;
                if (c >= n)
                    continue;
                if (lower && c >= r + 16)
                    continue;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    double *dst = C + (r + l15) + (long)(c + l4 + 4 * rr) * ldc;
                    if (pass == 0)
                        acc[tm][tn][rr] = *dst - acc[tm][tn][rr];
                    else
                        *dst = acc[tm][tn][rr];
                }
            }
        }
}

// ---------------------------------------------------------------------------
// The same wave tile on v_mfma_f64_4x4x4_4b_f64.  On gfx950 the 16x16x4 form sustains
// only ~48 TFLOP/s (62 % of the 78.6 peak: its MFMA_BUSY is 64 cycles but it cannot be
// issued back to back), the four-block 4x4x4 form ~70 (tools/probe_mfma.py).  Operand
// map, probed on the box (tools/probe_mfma444.py): A lane = i + 4 blk + 16 k,
// B lane = j + 4 blk + 16 k, D lane = j + 4 blk + 16 i  (D_blk = A_blk B_blk, 4x4x4 each).
//
// Both operands are the ordinary 16 x 4 fragments (64 distinct values, one load each):
// block blk of the B operand holds rows 4 blk .. 4 blk + 3 of the P fragment, block blk
// of the A operand columns 4 blk .. 4 blk + 3 of the Q fragment.  Rotating the Q
// fragment by s quads (DPP row_ror:4s) makes block blk meet column quad (blk - s) mod 4,
// so four MFMAs (s = 0..3) cover a 16 x 16 tile of C:
//   acc[tm][tn][s], lane l -> C[r + l%16, c + 4 (((l/4)%4 - s) mod 4) + l/16]
// i.e. an accumulator register still covers whole 128-byte lines of C.  Per k-step of
// 4 a (16 TM) x (16 TN) wave tile takes TM + TN fragment loads, 3 TN rotations and
// 4 TM TN MFMAs.  (Feeding the four-block form with replicated 4-column fragments
// instead needs 2.5x the load instructions and is bound by the L1: 36 TFLOP/s; one
// rotation of each operand instead of three of Q measures the same, with 32-byte
// store runs.)  Q must be unit-stride in its row index (qsj == 1) and m, n multiples
// of the wave tile (true for every padded system here).
// ---------------------------------------------------------------------------
// C tile access of a (16 TM) x (16 TN) wave tile held as acc[TM][TN][4] in the rotated-quad
// layout (gemm444_tile / gemm_lds_kernel).  Written as `*dst -= acc` at the end of the
// kernel, the compiler must assume that the store of one element aliases the load of the
// next and serialises 64 memory round trips per lane (measured: 14 % of the N=16384
// trailing update).  Instead the accumulators START as -C (every load in flight at once,
// straight into the accumulator registers, overlapping the first operand fetch; blocks
// above the diagonal are read and dropped so there is no branch between the loads), the
// MFMAs add P Q^T, and the epilogue only stores -acc = C - P Q^T.
// Addresses: wave-uniform tile origin (SGPR base) + four 32-bit per-lane offsets, one per
// rotation + immediates -- no per-element 64-bit address arithmetic.
template <int TM, int TN> struct Tile444 {
    char *tile;
    long tnstep;
    unsigned voff[4];
    int row0, col0;
    __device__ __forceinline__ Tile444(double *C, long ldc, int row0_, int col0_, int lane)
    {
        const int l15 = lane & 15, l4 = lane >> 4, blk = (lane >> 2) & 3;
        row0 = __builtin_amdgcn_readfirstlane(row0_);
        col0 = __builtin_amdgcn_readfirstlane(col0_);
        tile = reinterpret_cast<char *>(C + row0 + (long)col0 * ldc);
        tnstep = 16 * ldc * (long)sizeof(double);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            voff[s] = (unsigned)((l15 + (long)(l4 + 4 * ((blk - s) & 3)) * ldc) * (long)sizeof(double));
    }
    __device__ __forceinline__ double *at(int tm, int tn, int s) const
    {
        return reinterpret_cast<double *>((tile + (tn * tnstep + tm * 128)) + voff[s]);
    }
    __device__ __forceinline__ void load_neg(double (&acc)[TM][TN][4]) const
    {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[tm][tn][s] = *at(tm, tn, s);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[tm][tn][s] = -acc[tm][tn][s];
    }
    __device__ __forceinline__ void store_neg(const double (&acc)[TM][TN][4], int lower) const
    {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                if (lower && col0 + tn * 16 >= row0 + tm * 16 + 16)
                    continue;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    *at(tm, tn, s) = -acc[tm][tn][s];
            }
    }
};

template <int TM, int TN>
__device__ __forceinline__ void gemm444_tile(double *__restrict__ C, long ldc,
                                             const double *__restrict__ P, long ldp,
                                             const double *__restrict__ Q, long ldq, int k,
                                             int lower, int row0, int col0, int lane)
{
    const int l15 = lane & 15, l4 = lane >> 4;
    const double *pp = P + row0 + l15 + (long)l4 * ldp;
    const double *qq = Q + col0 + l15 + (long)l4 * ldq;

    double acc[TM][TN][4];
    const Tile444<TM, TN> ct(C, ldc, row0, col0, lane);
    ct.load_neg(acc);

    double pa[TM], qa[TN], pb[TM], qb[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
        pa[tm] = pp[16 * tm];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
        qa[tn] = qq[16 * tn];
    const long pstep = 4 * ldp, qstep = 4 * ldq;
    const int ksteps = k >> 2; // even

#define BQ_444_ROT                                                                                 \
    const double q1 = row_ror_quads<1>(q0), q2 = row_ror_quads<2>(q0), q3 = row_ror_quads<3>(q0);
#define BQ_444_KOFF(ks) (long)(ks)
#define BQ_444_STEP(PF, QF)                                                                        \
    _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)                                              \
    {                                                                                              \
        const double q0 = QF[tn];                                                                  \
        BQ_444_ROT                                                                                 \
        _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                                          \
        {                                                                                          \
            acc[tm][tn][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(q0, PF[tm], acc[tm][tn][0], 0, 0, 0); \
            acc[tm][tn][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(q1, PF[tm], acc[tm][tn][1], 0, 0, 0); \
            acc[tm][tn][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(q2, PF[tm], acc[tm][tn][2], 0, 0, 0); \
            acc[tm][tn][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(q3, PF[tm], acc[tm][tn][3], 0, 0, 0); \
        }                                                                                          \
    }

    for (int ks = 0; ks < ksteps; ks += 2) {
        const double *p1 = pp + BQ_444_KOFF(ks + 1) * pstep, *q1p = qq + BQ_444_KOFF(ks + 1) * qstep;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
            pb[tm] = p1[16 * tm];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            qb[tn] = q1p[16 * tn];
        __builtin_amdgcn_sched_barrier(0);
        BQ_444_STEP(pa, qa)
        __builtin_amdgcn_sched_barrier(0);
        const long o2 = BQ_444_KOFF((ks + 2 < ksteps) ? (ks + 2) : ks); // clamped, value unused
        const double *p2 = pp + o2 * pstep, *q2p = qq + o2 * qstep;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
            pa[tm] = p2[16 * tm];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            qa[tn] = q2p[16 * tn];
        __builtin_amdgcn_sched_barrier(0);
        BQ_444_STEP(pb, qb)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef BQ_444_STEP
#undef BQ_444_ROT
#undef BQ_444_KOFF

    ct.store_neg(acc, lower);
}

// Fused diagonal factor: when fuse_j0 >= 0 the launch also factors the leading 64x64
// block of C (the next diagonal block of the Cholesky) right after updating it.
// Workgroup 0 owns every workgroup tile that intersects that block, updates them,
// and runs potf2f_body on the result; the other workgroups of the block exit.
// This removes one dependent launch (and the block's trip through L2) per 64 columns.
template <int TM, int TN, int MF> // MF = 0: v_mfma_f64_16x16x4_f64, 1: v_mfma_f64_4x4x4_4b_f64
__global__ __launch_bounds__(256, 2) void gemm_sub_kernel(double *__restrict__ C, long ldc,
                                                          long cstride, const double *__restrict__ P,
                                                          long ldp, long pstride,
                                                          const double *__restrict__ Q, long qsj,
                                                          long qsk, long qstride, int m, int n,
                                                          int k, int lower, int fuse_j0,
                                                          double *__restrict__ dinv, long dstride,
                                                          int *__restrict__ info)
{
    __shared__ __attribute__((aligned(16))) double plds[BQ_POTF2_LDS_DOUBLES]; // the fused diagonal factor's LDS
    // the small-tile forms are the panel's own updates: on the look-ahead stream they share
    // CUs with the bulk trailing update and sit on the critical path, so they issue first
    if (TM < 4)
        __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int bx = blockIdx.x, by = blockIdx.y;
    if (lower == 2)
        tri_decode(blockIdx.x, bx, by);
    C += (long)b * cstride;
    P += (long)b * pstride;
    Q += (long)b * qstride;
    constexpr int WT = 32 * TM; // rows (and, TM == TN, columns) of a workgroup tile
    if (fuse_j0 >= 0 && bx * WT < 64 && by * (32 * TN) < 64) {
        if (bx != 0 || by != 0)
            return; // inside the diagonal block: workgroup 0 does it
        constexpr int NS = (WT >= 64) ? 1 : 64 / WT;
        for (int sx = 0; sx < NS; ++sx)
            for (int sy = 0; sy <= sx; ++sy) {
                const int row0 = (sx * 2 + (wave & 1)) * (TM * 16);
                const int col0 = (sy * 2 + (wave >> 1)) * (TN * 16);
                if (row0 < m && col0 < n && !(lower && col0 >= row0 + TM * 16)) {
                    if (MF == 1)
                        gemm444_tile<TM, TN>(C, ldc, P, ldp, Q, qsk, k, lower, row0, col0, lane);
                    else
                        gemm_sub_tile<TM, TN>(C, ldc, P, ldp, Q, qsj, qsk, m, n, k, lower, row0,
                                              col0, lane);
                }
            }
        __syncthreads(); // the updated block is visible to the whole workgroup
        potf2_body(C, ldc, fuse_j0, dinv + (long)b * dstride, info + b, plds);
        return;
    }
    const int row0 = (bx * 2 + (wave & 1)) * (TM * 16);
    const int col0 = (by * 2 + (wave >> 1)) * (TN * 16);
    if (row0 >= m || col0 >= n)
        return;
    if (lower && col0 >= row0 + TM * 16)
        return;
    if (MF == 1)
        gemm444_tile<TM, TN>(C, ldc, P, ldp, Q, qsk, k, lower, row0, col0, lane);
    else
        gemm_sub_tile<TM, TN>(C, ldc, P, ldp, Q, qsj, qsk, m, n, k, lower, row0, col0, lane);
}

// ---------------------------------------------------------------------------
// The same product for k == 64 exactly (the trailing / panel update of small
// systems, outer block 64): all 16 k-steps of fragments are requested up front
// and the MFMAs drain them as they land, so a tile costs one memory round trip
// instead of sixteen.  TM, TN <= 2.
// ---------------------------------------------------------------------------
template <int TM, int TN>
__device__ __forceinline__ void gemm_k64_tile(double *__restrict__ C, long ldc,
                                              const double *__restrict__ P, long ldp,
                                              const double *__restrict__ Q, long qsj, long qsk,
                                              int m, int n, int lower, int row0, int col0, int lane)
{
    const int l15 = lane & 15, l4 = lane >> 4;
    double pa[16][TM], qa[16][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        int r = row0 + tm * 16;
        if (r >= m)
            r = row0;
        const double *pp = P + r + l15 + (long)l4 * ldp;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            pa[ks][tm] = pp[(long)ks * 4 * ldp];
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        int c = col0 + tn * 16;
        if (c >= n)
            c = col0;
        const double *qq = Q + (long)(c + l15) * qsj + (long)l4 * qsk;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            qa[ks][tn] = qq[(long)ks * 4 * qsk];
    }
    // C is read while the fragments are in flight
    double cold[TM][TN][4];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                int r = row0 + tm * 16, c = col0 + tn * 16;
                if (r >= m) r = row0;
                if (c >= n) c = col0;
                cold[tm][tn][rr] = C[(r + l15) + (long)(c + l4 + 4 * rr) * ldc];
            }
    // every load above is issued before the first MFMA (the scheduler otherwise
    // interleaves them to save registers and serialises the round trips)
    __builtin_amdgcn_sched_barrier(0);
    double4_t acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[ks][tn], pa[ks][tm],
                                                                   acc[tm][tn], 0, 0, 0);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int r = row0 + tm * 16;
        if (r >= m)
            continue;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int c = col0 + tn * 16;
            if (c >= n)
                continue;
            if (lower && c >= r + 16)
                continue;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                C[(r + l15) + (long)(c + l4 + 4 * rr) * ldc] = cold[tm][tn][rr] - acc[tm][tn][rr];
        }
    }
}

template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_k64_kernel(double *__restrict__ C, long ldc,
                                                       long cstride, const double *__restrict__ P,
                                                       long ldp, long pstride,
                                                       const double *__restrict__ Q, long qsj,
                                                       long qsk, long qstride, int m, int n,
                                                       int lower, int fuse_j0,
                                                       double *__restrict__ dinv, long dstride,
                                                       int *__restrict__ info)
{
    __shared__ __attribute__((aligned(16))) double plds[BQ_POTF2_LDS_DOUBLES]; // the fused diagonal factor's LDS
    __builtin_amdgcn_s_setprio(3); // panel-internal update: see gemm_sub_kernel
    const int b = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int bx = blockIdx.x, by = blockIdx.y;
    if (lower == 2)
        tri_decode(blockIdx.x, bx, by);
    C += (long)b * cstride;
    P += (long)b * pstride;
    Q += (long)b * qstride;
    constexpr int WT = 32 * TM;
    if (fuse_j0 >= 0 && bx * WT < 64 && by * (32 * TN) < 64) { // see gemm_sub_kernel
        if (bx != 0 || by != 0)
            return;
        constexpr int NS = (WT >= 64) ? 1 : 64 / WT;
        for (int sx = 0; sx < NS; ++sx)
            for (int sy = 0; sy <= sx; ++sy) {
                const int row0 = (sx * 2 + (wave & 1)) * (TM * 16);
                const int col0 = (sy * 2 + (wave >> 1)) * (TN * 16);
                if (row0 < m && col0 < n && !(lower && col0 >= row0 + TM * 16))
                    gemm_k64_tile<TM, TN>(C, ldc, P, ldp, Q, qsj, qsk, m, n, lower, row0, col0,
                                          lane);
            }
        __syncthreads();
        potf2_body(C, ldc, fuse_j0, dinv + (long)b * dstride, info + b, plds);
        return;
    }
    const int row0 = (bx * 2 + (wave & 1)) * (TM * 16);
    const int col0 = (by * 2 + (wave >> 1)) * (TN * 16);
    if (row0 >= m || col0 >= n)
        return;
    if (lower && col0 >= row0 + TM * 16)
        return;
    gemm_k64_tile<TM, TN>(C, ldc, P, ldp, Q, qsj, qsk, m, n, lower, row0, col0, lane);
}

// ---------------------------------------------------------------------------
// Trailing update, LDS-staged: C(m x n) -= P Q^T, workgroup tile 128 x 128, wave tile
// 64 x 64, v_mfma_f64_4x4x4_4b_f64.  The register-streaming kernel above reads every
// fragment from global memory in two waves and spends VALU issue slots on the quad
// rotations and on 64-bit load addresses (0.57 VALU instructions per MFMA, PMC); VALU
// and the f64 MFMA do not co-execute, so that is what it is bound by.  Here a
// workgroup stages 16 k-columns of its P and Q row blocks in LDS with LDS-DMA
// (global_load_lds_dwordx4: one wave instruction moves one k row of 128 doubles, no
// staging registers, no ds_write), a chunk ahead of the MFMAs, and every wave reads its
// fragments from LDS with immediate offsets: 4 P fragments and, instead of rotating, the
// 16 pre-rotated views of its 4 Q fragments -- a rotated view is just another address
// pattern of the same 16 doubles.  Fragments of k-step s+1 are read while the MFMAs of
// k-step s run.  The inner loop has no VALU work at all.
// LDS: 2 buffers x (16 P rows + 16 Q rows) x 1152 B = 72 KiB per workgroup, two
// workgroups per CU.  One barrier per chunk (256 MFMAs per wave).
// Requires qsj == 1, m and n multiples of 64, k a multiple of 32.
// ---------------------------------------------------------------------------
#define BQ_LDS_KC 16                 // k columns per chunk
#define BQ_LDS_ROW (128 * 8 + 128)   // bytes per staged k row (128 doubles + pad)
#define BQ_LDS_STAGE (2 * BQ_LDS_KC * BQ_LDS_ROW)
#define BQ_LDS_BYTES (2 * BQ_LDS_STAGE)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_cvoid_t;

// SD > 0: C has not been written yet -- the accumulators start as minus the bordered system's own
// entries (gram_seed_neg<SD>, points of dimension SD) instead of minus a loaded tile
template <int SD>
__device__ __forceinline__ void gemm_lds_body(double *__restrict__ C, long ldc, long cstride,
                                              const double *__restrict__ P, long ldp, long pstride,
                                              const double *__restrict__ Q, long ldq, long qstride,
                                              int m, int n, int k, int lower, int ncut,
                                              const GramSeed *sd)
{
    // ncut: columns >= ncut of C are left alone (the border x border block of a bordered
    // system, which nothing reads: see plan_readout_kernel)
    // [buffer][P rows 0..15 | Q rows 16..31][128 doubles + pad]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.z;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6); // scalar: LDS-DMA bases stay in SGPRs
    int bx = blockIdx.x, by = blockIdx.y;
    if (lower == 2)
        tri_decode(blockIdx.x, bx, by);
    C += (long)b * cstride;
    P += (long)b * pstride;
    Q += (long)b * qstride;
    const int R0 = bx * 128, C0 = by * 128;
    if (C0 >= ncut)
        return; // (the whole workgroup, before any barrier)
    const int wr = (wave & 1) * 64, wc = (wave >> 1) * 64; // this wave's 64 x 64 sub-tile
    const int row0 = R0 + wr, col0 = C0 + wc;
    const bool active = row0 < m && col0 < n && col0 < ncut && !(lower && col0 >= row0 + 64);

    // staging: a chunk is 32 k rows (16 of P, 16 of Q); wave w moves rows 8w .. 8w+7, one
    // LDS-DMA per row, lane i carrying rows 2i, 2i+1 of the operand's 128-row block
    // (clamped at the matrix edge: the clamped rows feed accumulators that are not stored)
    const bool stq = wave >= 2;
    const long sld = stq ? ldq : ldp;
    const double *gsrc = stq ? Q + min(C0 + 2 * lane, n - 2) + (long)(8 * (wave - 2)) * ldq
                             : P + min(R0 + 2 * lane, m - 2) + (long)(8 * wave) * ldp;
    const int srow = wave * 8 * BQ_LDS_ROW; // wave-uniform LDS offset of this wave's first row

    const int l15 = lane & 15, l4 = lane >> 4;
    // fragment read offsets inside a stage buffer: k row l4 of a k-step
    const unsigned char *pview = smem + l4 * BQ_LDS_ROW + (wr + l15) * 8;
    const unsigned char *qview[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) // view rotated by s quads: lane reads column (l15 - 4 s) mod 16
        qview[s] = smem + (BQ_LDS_KC + l4) * BQ_LDS_ROW + (wc + ((l15 - 4 * s) & 15)) * 8;

    double acc[4][4][4];

#define BQ_LDS_FILL(BUF_, CH_)                                                                     \
    {                                                                                              \
        const double *g_ = gsrc + (long)(CH_) * BQ_LDS_KC * sld;                                   \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) __builtin_amdgcn_global_load_lds(            \
            (global_cvoid_t *)(g_ + (long)r * sld),                                                \
            (lds_void_t *)(smem + (BUF_) * BQ_LDS_STAGE + srow + r * BQ_LDS_ROW), 16, 0, 0);       \
    }
    // one chunk from buffer BUF_: (1) every wave's LDS-DMA of this chunk has landed and
    // every wave is done with the other buffer; (2) LDS-DMA of the next chunk into it;
    // (3) 16 sub-steps (k-step st, column block tn) of 16 MFMAs; the 4 rotated Q views of
    // sub-step j+1 (and the 4 P fragments of the next k-step) are read during sub-step j
#define BQ_LDS_READ_P(BUF_, ST_, PF)                                                               \
    _Pragma("unroll") for (int tm = 0; tm < 4; ++tm) PF[tm] = *reinterpret_cast<const double *>(   \
        pview + (BUF_) * BQ_LDS_STAGE + (ST_) * 4 * BQ_LDS_ROW + tm * 128);
#define BQ_LDS_READ_Q(BUF_, ST_, TN_, QF)                                                          \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) QF[s] = *reinterpret_cast<const double *>(       \
        qview[s] + (BUF_) * BQ_LDS_STAGE + (ST_) * 4 * BQ_LDS_ROW + (TN_) * 128);
#define BQ_LDS_LANDED(QF, PF)                                                                      \
    asm volatile("" ::"v"(QF[0]), "v"(QF[1]), "v"(QF[2]), "v"(QF[3]), "v"(PF[0]), "v"(PF[1]),      \
                 "v"(PF[2]), "v"(PF[3]));
#define BQ_LDS_CHUNK(BUF_, CH_)                                                                    \
    {                                                                                              \
        __syncthreads();                                                                           \
        if ((CH_) + 1 < nchunk)                                                                    \
            BQ_LDS_FILL(1 - (BUF_), (CH_) + 1)                                                     \
        if (active) {                                                                              \
            BQ_LDS_READ_P(BUF_, 0, pf[0])                                                          \
            BQ_LDS_READ_Q(BUF_, 0, 0, qf[0])                                                       \
            _Pragma("unroll") for (int j = 0; j < 16; ++j)                                         \
            {                                                                                      \
                const int st = j >> 2, tn = j & 3;                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                 \
                /* the fragments of THIS sub-step are waited for here, before the next ones are  */ \
                /* requested: hipcc waits with lgkmcnt(0) whatever is in flight, and placed after */ \
                /* the new requests that wait exposed their full latency every sub-step           */ \
                BQ_LDS_LANDED(qf[j & 1], pf[st & 1])                                               \
                __builtin_amdgcn_sched_barrier(0);                                                 \
                if (j < 15)                                                                        \
                    BQ_LDS_READ_Q(BUF_, (j + 1) >> 2, (j + 1) & 3, qf[(j + 1) & 1])                \
                if (tn == 3 && j < 15)                                                             \
                    BQ_LDS_READ_P(BUF_, st + 1, pf[(st + 1) & 1])                                  \
                __builtin_amdgcn_sched_barrier(0);                                                 \
                _Pragma("unroll") for (int tm = 0; tm < 4; ++tm)                                   \
                    _Pragma("unroll") for (int s = 0; s < 4; ++s) acc[tm][tn][s] =                 \
                        __builtin_amdgcn_mfma_f64_4x4x4f64(qf[j & 1][s], pf[st & 1][tm],           \
                                                           acc[tm][tn][s], 0, 0, 0);               \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    }

    double pf[2][4], qf[2][4];
    const int nchunk = k / BQ_LDS_KC; // even
    BQ_LDS_FILL(0, 0)
    const Tile444<4, 4> ct(C, ldc, row0, col0, lane);
    if (active) {
        if (SD > 0)
            gram_seed_neg<(SD > 0 ? SD : 1), 4, 4>(acc, *sd, b, row0, col0, lane);
        else
            ct.load_neg(acc);
    }
    for (int ch = 0; ch < nchunk; ch += 2) {
        BQ_LDS_CHUNK(0, ch)
        BQ_LDS_CHUNK(1, ch + 1)
    }
#undef BQ_LDS_READ_P
#undef BQ_LDS_READ_Q
#undef BQ_LDS_LANDED
#undef BQ_LDS_CHUNK
#undef BQ_LDS_FILL

    if (!active)
        return;
    ct.store_neg(acc, lower);
}

__global__ __launch_bounds__(256, 2) void gemm_lds_kernel(double *__restrict__ C, long ldc,
                                                          long cstride, const double *__restrict__ P,
                                                          long ldp, long pstride,
                                                          const double *__restrict__ Q, long ldq,
                                                          long qstride, int m, int n, int k,
                                                          int lower, int ncut)
{
    gemm_lds_body<0>(C, ldc, cstride, P, ldp, pstride, Q, ldq, qstride, m, n, k, lower, ncut, nullptr);
}

// ... its tile of C computed, not loaded (the first product over a region the assembly left out)
template <int SD>
__global__ __launch_bounds__(256, 2) void gemm_lds_seed_kernel(
    double *__restrict__ C, long ldc, long cstride, const double *__restrict__ P, long ldp,
    long pstride, const double *__restrict__ Q, long ldq, long qstride, int m, int n, int k,
    int lower, int ncut, GramSeed sd)
{
    gemm_lds_body<SD>(C, ldc, cstride, P, ldp, pstride, Q, ldq, qstride, m, n, k, lower, ncut, &sd);
}


// ---------------------------------------------------------------------------
// The same product with a 64 x 64 workgroup tile (four waves of 32 x 32): four times the
// workgroups of gemm_lds_kernel for products whose 128 x 128 tiles cannot fill the chip -- the
// row sweeps over a resident factor with a few hundred right-hand sides (a 256-row operand
// gives the 128-tile kernel ONE workgroup per CU whatever the other dimension, and a step then
// costs one 65 us workgroup lifetime however little work it has), the late trailing updates and
// the panel-internal products of a batch.  36 KiB of LDS and <= 128 VGPRs: four workgroups
// per CU, sixteen waves to hide each other's barriers.
// Staging: an LDS-DMA instruction moves 1 KiB = TWO k rows of a 64-row operand block; DMA row
// d (1152 B with the pad) holds k rows d and d + 8 of the chunk, so that the four k rows of a
// k-step sit in four consecutive DMA rows of one half -- the bank pattern of the 128-row layout
// (rows 1152 B apart, conflict-free ds_read_b64).  Chunk = 16 k columns = 8 DMA rows of P +
// 8 of Q; wave w stages DMA rows 4 (w & 1) .. + 3 of P (w < 2) or Q.
// Requires ldq unit stride in the row index, m and n multiples of 64, k of 32.
// QT: Q is given k-contiguous -- Q(j, k) at Q[j ldq + k], the form of the backward row sweep's
// operand L[J.., 0..J)^T, which the register-streaming kernels read with a stride of ldl between
// lanes (24 TFLOP/s against 45 for the forward sweep).  A DMA lane then carries k rows 2 d,
// 2 d + 1 of ONE operand row (16 contiguous bytes), a DMA row the k-row pair d of all 64 rows,
// interleaved; the fragment views only change their address pattern (lanes l4 = 0, 1 of a
// k-step read the two halves of consecutive 16-byte pairs: 256 contiguous bytes, no conflict).
// ---------------------------------------------------------------------------
#define BQ_L64_STAGE (16 * BQ_LDS_ROW)
#define BQ_L64_BYTES (2 * BQ_L64_STAGE)

// the workgroup tile (bx, by); C, P, Q point at the batch element
// KS = 2: the workgroup has EIGHT waves, two groups of four that split every chunk's k range
// (group g takes k-steps 2g, 2g + 1 of the four) and meet in LDS at the end: the form the tile
// takes inside rows_fused_kernel's eight-wave workgroups (whose job tiles gain from eight waves;
// this tile neither gains nor loses).
// TRSM (the batched panel solve, potrf.hip: enqueue_trsm_rec): the tiles of column block 0 -- the
// 64-column slab that is solved next -- do not store C - P Q^T but (C - P Q^T) L_ss^-T, the
// trsm_blk_kernel scheme applied to the tile on its way out (through LDS into that kernel's
// 16-rows-per-wave operand form): one launch and one pass over the slab less per 64 columns.
// Lss: the slab's factored 64 x 64 diagonal block, wrec: its record of block inverses.
template <bool QT, int KS = 1, bool TRSM = false, int SD = 0>
__device__ __forceinline__ void gemm_lds64_body(unsigned char *smem, double *__restrict__ C,
                                                long ldc, const double *__restrict__ P, long ldp,
                                                const double *__restrict__ Q, long ldq, int m,
                                                int n, int k, int lower, int ncut, int bx, int by,
                                                const double *__restrict__ Lss = nullptr,
                                                long ldl = 0,
                                                const double *__restrict__ wrec = nullptr,
                                                const GramSeed *sd = nullptr, int sb = 0)
{
    const int t = threadIdx.x, lane = t & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wave = wave8 & 3, grp = wave8 >> 2; // place in the 2 x 2 wave layout; k group
    const int R0 = bx * 64, C0 = by * 64;
    if (C0 >= ncut)
        return; // (the whole workgroup, before any barrier)
    const int wr = (wave & 1) * 32, wc = (wave >> 1) * 32;
    const int row0 = R0 + wr, col0 = C0 + wc;
    const bool active = row0 < m && col0 < n && col0 < ncut && !(lower && col0 >= row0 + 32);

    // staging: 8 DMA rows of P and 8 of Q per chunk, dealt to the 4 KS waves
    constexpr int NDMA = 4 / KS; // DMA rows per wave and chunk
    const bool stq = wave8 >= 2 * KS;
    const int dma0 = NDMA * (wave8 & (2 * KS - 1)); // this wave's first DMA row of its operand
    // lane i: operand rows 2 (i & 31), + 1 of k row (dma row) + 8 (i >> 5);
    // QT operand: operand row i, k rows 2 (dma row), + 1
    const double *gsrc;
    long sld;    // source step from one DMA row to the next
    long schunk; // ... and from one chunk to the next
    if (QT && stq) {
        gsrc = Q + (long)min(C0 + lane, n - 1) * ldq + 2 * dma0;
        sld = 2;
        schunk = 16;
    } else {
        sld = stq ? ldq : ldp;
        gsrc = (stq ? Q + min(C0 + 2 * (lane & 31), n - 2) : P + min(R0 + 2 * (lane & 31), m - 2)) +
               (long)(dma0 + 8 * (lane >> 5)) * sld;
        schunk = 16 * sld;
    }
    const int srow = ((stq ? 8 : 0) + dma0) * BQ_LDS_ROW;

    const int l15 = lane & 15, l4 = lane >> 4;
    // (KS = 2: group g's k-steps are 2g, 2g + 1 -- the half of the DMA rows at + 512 B, and for
    // a k-contiguous Q the DMA rows from 4g on: folded into the view bases)
    const unsigned char *pview = smem + l4 * BQ_LDS_ROW + (wr + l15) * 8 + (KS == 2 ? grp * 512 : 0);
    const unsigned char *qview[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        qview[s] = QT ? smem + (8 + (l4 >> 1) + (KS == 2 ? 4 * grp : 0)) * BQ_LDS_ROW +
                            (wc + ((l15 - 4 * s) & 15)) * 16 + (l4 & 1) * 8
                      : smem + (8 + l4) * BQ_LDS_ROW + (wc + ((l15 - 4 * s) & 15)) * 8 +
                            (KS == 2 ? grp * 512 : 0);

    double acc[2][2][4];

#define BQ_L64_FILL(BUF_, CH_)                                                                     \
    {                                                                                              \
        const double *g_ = gsrc + (long)(CH_) * schunk;                                            \
        _Pragma("unroll") for (int r = 0; r < NDMA; ++r) __builtin_amdgcn_global_load_lds(         \
            (global_cvoid_t *)(g_ + (long)r * sld),                                                \
            (lds_void_t *)(smem + (BUF_) * BQ_L64_STAGE + srow + r * BQ_LDS_ROW), 16, 0, 0);       \
    }
    // k-step ST_ of a chunk: k rows 4 ST_ + l4 = DMA rows 4 (ST_ & 1) + l4 of half ST_ >> 1
#define BQ_L64_OFF(BUF_, ST_) ((BUF_) * BQ_L64_STAGE + 4 * ((ST_) & 1) * BQ_LDS_ROW + ((ST_) >> 1) * 512)
#define BQ_L64_READ_P(BUF_, ST_, PF)                                                               \
    _Pragma("unroll") for (int tm = 0; tm < 2; ++tm) PF[tm] = *reinterpret_cast<const double *>(   \
        pview + BQ_L64_OFF(BUF_, ST_) + tm * 128);
    // (QT: k-step ST_ = DMA rows 2 ST_, 2 ST_ + 1; a 16-row fragment block is 256 B apart)
#define BQ_L64_QOFF(BUF_, ST_, TN_)                                                                \
    (QT ? (BUF_) * BQ_L64_STAGE + 2 * (ST_) * BQ_LDS_ROW + (TN_) * 256                             \
        : BQ_L64_OFF(BUF_, ST_) + (TN_) * 128)
#define BQ_L64_READ_Q(BUF_, ST_, TN_, QF)                                                          \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) QF[s] = *reinterpret_cast<const double *>(       \
        qview[s] + BQ_L64_QOFF(BUF_, ST_, TN_));
#define BQ_L64_CHUNK(BUF_, CH_)                                                                    \
    {                                                                                              \
        __syncthreads();                                                                           \
        if ((CH_) + 1 < nchunk)                                                                    \
            BQ_L64_FILL(1 - (BUF_), (CH_) + 1)                                                     \
        if (active) {                                                                              \
            BQ_L64_READ_P(BUF_, 0, pf[0])                                                          \
            BQ_L64_READ_Q(BUF_, 0, 0, qf[0])                                                       \
            _Pragma("unroll") for (int j = 0; j < 8 / KS; ++j)                                     \
            {                                                                                      \
                const int st = j >> 1, tn = j & 1;                                                 \
                __builtin_amdgcn_sched_barrier(0);                                                 \
                /* (this sub-step's fragments are waited for BEFORE the next are requested:       */ \
                /* see gemm_lds_kernel)                                                            */ \
                asm volatile("" ::"v"(qf[j & 1][0]), "v"(qf[j & 1][1]), "v"(qf[j & 1][2]),         \
                             "v"(qf[j & 1][3]), "v"(pf[st & 1][0]), "v"(pf[st & 1][1]));           \
                __builtin_amdgcn_sched_barrier(0);                                                 \
                if (j < 8 / KS - 1)                                                                \
                    BQ_L64_READ_Q(BUF_, (j + 1) >> 1, (j + 1) & 1, qf[(j + 1) & 1])                \
                if (tn == 1 && j < 8 / KS - 1)                                                     \
                    BQ_L64_READ_P(BUF_, st + 1, pf[(st + 1) & 1])                                  \
                __builtin_amdgcn_sched_barrier(0);                                                 \
                _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                   \
                    _Pragma("unroll") for (int s = 0; s < 4; ++s) acc[tm][tn][s] =                 \
                        __builtin_amdgcn_mfma_f64_4x4x4f64(qf[j & 1][s], pf[st & 1][tm],           \
                                                           acc[tm][tn][s], 0, 0, 0);               \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    }

    double pf[2][2], qf[2][4];
    const int nchunk = k / 16; // even
    const Tile444<2, 2> ct(C, ldc, row0, col0, lane);
    // (KS = 2 measured with four stages and counted vmcnt waits as well: no faster -- with the
    // LDS-DMA removed a chunk takes 0.54 us, with it 0.52-0.55: the k loop is MFMA-bound at 80 %
    // of a CU's rate whether four waves walk it or eight; what the eight gain is the job tiles')
    BQ_L64_FILL(0, 0)
    if (active && (KS == 1 || grp == 0)) {
        if (SD > 0) // (the tile computed from the problem's points: see gemm_lds_body)
            gram_seed_neg<(SD > 0 ? SD : 1), 2, 2>(acc, *sd, sb, row0, col0, lane);
        else
            ct.load_neg(acc);
    } else if (KS == 2) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[a][b2][s] = 0.0;
    }
    for (int ch = 0; ch < nchunk; ch += 2) {
        BQ_L64_CHUNK(0, ch)
        BQ_L64_CHUNK(1, ch + 1)
    }
#undef BQ_L64_READ_P
#undef BQ_L64_READ_Q
#undef BQ_L64_CHUNK
#undef BQ_L64_FILL
#undef BQ_L64_OFF
#undef BQ_L64_QOFF

    if (KS == 2) {
        // group 1's partial sums -> LDS (16 doubles per lane, 32 KiB: the staging buffers are
        // free once every wave is past its last chunk) -> group 0
        double *red = reinterpret_cast<double *>(smem) + wave * 1024 + lane;
        __syncthreads();
        if (grp == 1 && active) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        red[64 * (8 * a + 4 * b2 + s)] = acc[a][b2][s];
        }
        __syncthreads();
        if (grp == 1 || !active)
            return;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[a][b2][s] += red[64 * (8 * a + 4 * b2 + s)];
        ct.store_neg(acc, lower);
        return;
    }
    if (TRSM && by == 0) {
        // (lower == 0 and whole tiles here: every wave is active)
        double *Ts = reinterpret_cast<double *>(smem); // the updated tile, column-major 64 x 64
        __syncthreads();                               // the staging buffers are free
        {
            const int blk = (lane >> 2) & 3;
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        Ts[wr + 16 * tm + l15 + 64 * (wc + 16 * tn + l4 + 4 * ((blk - s) & 3))] =
                            -acc[tm][tn][s];
        }
        __syncthreads();
        // rows 16 wave .. + 15 of the tile: X_c = (T_c - sum_{b<c} X_b L_cb^T) W_cc^T (trsm.h)
        const double *L11 = Lss + l15 + (long)l4 * ldl;
        const double *W = wrec + 64 + l15 + 16 * l4;
        const double *Tw = Ts + 16 * wave + l15 + 64 * l4;
        double *Xr = C + R0 + 16 * wave + l15 + (long)(C0 + l4) * ldc;
        double4_t x[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double4_t a4 = {-Tw[64 * (16 * c)], -Tw[64 * (16 * c + 4)], -Tw[64 * (16 * c + 8)],
                            -Tw[64 * (16 * c + 12)]};
#pragma unroll
            for (int bb = 0; bb < c; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(
                        L11[16 * c + (long)(16 * bb + 4 * r) * ldl], x[bb][r], a4, 0, 0, 0);
            double4_t xc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r)
                xc = __builtin_amdgcn_mfma_f64_16x16x4f64(-W[256 * c + 64 * r], a4[r], xc, 0, 0, 0);
            x[c] = xc;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Xr[(long)(16 * c + 4 * r) * ldc] = xc[r];
        }
        return;
    }
    if (!active)
        return;
    ct.store_neg(acc, lower);
}

// the product of the batched panel solve: C (m x n) -= P Q^T with the solve of the first 64
// columns of C fused in (gemm_lds64_body<.., TRSM>); m, n multiples of 64, k of 32
__global__ __launch_bounds__(256, 4) void gemm_trsm64_kernel(
    double *__restrict__ C, long ldc, long cstride, const double *__restrict__ P, long ldp,
    long pstride, const double *__restrict__ Q, long ldq, long qstride, int m, int n, int k,
    const double *__restrict__ Lss, long ldl, long lstride, const double *__restrict__ wrec,
    long wstride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.z;
    gemm_lds64_body<false, 1, true>(smem, C + (long)b * cstride, ldc, P + (long)b * pstride, ldp,
                                    Q + (long)b * qstride, ldq, m, n, k, 0, 0x7fffffff,
                                    blockIdx.x, blockIdx.y, Lss + (long)b * lstride, ldl,
                                    wrec + (long)b * wstride);
}

template <bool QT, int KS = 1>
__global__ __launch_bounds__(256 * KS, 4 / KS) void gemm_lds64_kernel(
    double *__restrict__ C, long ldc, long cstride, const double *__restrict__ P, long ldp,
    long pstride, const double *__restrict__ Q, long ldq, long qstride, int m, int n, int k,
    int lower, int ncut)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.z;
    int bx = blockIdx.x, by = blockIdx.y;
    if (lower == 2)
        tri_decode(blockIdx.x, bx, by);
    gemm_lds64_body<QT, KS>(smem, C + (long)b * cstride, ldc, P + (long)b * pstride, ldp,
                            Q + (long)b * qstride, ldq, m, n, k, lower, ncut, bx, by);
}

template <int SD>
__global__ __launch_bounds__(256, 4) void gemm_lds64_seed_kernel(
    double *__restrict__ C, long ldc, long cstride, const double *__restrict__ P, long ldp,
    long pstride, const double *__restrict__ Q, long ldq, long qstride, int m, int n, int k,
    int lower, int ncut, GramSeed sd)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.z;
    int bx = blockIdx.x, by = blockIdx.y;
    if (lower == 2)
        tri_decode(blockIdx.x, bx, by);
    gemm_lds64_body<false, 1, false, SD>(smem, C + (long)b * cstride, ldc, P + (long)b * pstride,
                                         ldp, Q + (long)b * qstride, ldq, m, n, k, lower, ncut, bx,
                                         by, nullptr, 0, nullptr, &sd, b);
}


// ---------------------------------------------------------------------------
// C (m x n) -= P (m x k) Q for the SMALL products of the row sweeps over a resident factor
// (posterior variance at C2 size: m = 256 prediction points, n <= 768, k = 256): the 64 x 64
// tiles of gemm_sub_kernel leave 16-48 workgroups that each walk 64 dependent k-steps with
// one step of prefetch -- 16 us per launch, seven launches per prediction.  Here a workgroup
// owns a 32 x 32 tile and its four waves SPLIT k (a quarter each, four k-steps of fragments
// in flight), meet in LDS, and wave w subtracts 16 x 16 block w: 4x the workgroups, 1/4 of
// the dependent steps.  Works on transposes like the other kernels (A operand = Q fragment,
// B operand = P fragment, so that a D register holds 16 consecutive rows of C).
// m, n multiples of 32, k of 16, k <= 2048.  grid (m / 32, n / 32).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_splitk_kernel(double *__restrict__ C, long ldc,
                                                          const double *__restrict__ P, long ldp,
                                                          const double *__restrict__ Q, long qsj,
                                                          long qsk, int k)
{
    __shared__ double red[4][4][4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int kq = k >> 2, kb = wave * kq;
    const double *pp = P + (long)blockIdx.x * 32 + l15 + (long)(kb + l4) * ldp;
    const double *qq = Q + ((long)blockIdx.y * 32 + l15) * qsj + (long)(kb + l4) * qsk;
    double4_t acc[2][2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
            acc[jb][ib] = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int ks = 0; ks < kq; ks += 16) {
        double a[4][2], b[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                a[u][h] = pp[16 * h + (long)(ks + 4 * u) * ldp];
                b[u][h] = qq[(long)(16 * h) * qsj + (long)(ks + 4 * u) * qsk];
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
                    acc[jb][ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[u][jb], a[u][ib],
                                                                       acc[jb][ib], 0, 0, 0);
    }
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[wave][2 * jb + ib][r][lane] = acc[jb][ib][r];
    __syncthreads();
    const int jb = wave >> 1, ib = wave & 1;
    double *cp = C + (long)blockIdx.x * 32 + 16 * ib + l15 +
                 ((long)blockIdx.y * 32 + 16 * jb + l4) * ldc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double s = (red[0][wave][r][lane] + red[1][wave][r][lane]) +
                         (red[2][wave][r][lane] + red[3][wave][r][lane]);
        cp[(long)(4 * r) * ldc] -= s;
    }
}

// ---------------------------------------------------------------------------
// One forward step of the row sweep over a resident factor in ONE launch, for the small
// systems gemm_splitk_kernel serves.  With T_J = W_J L[J, J-B] (trsv.h) the step's two products
//     Y_J            = X_J W_J^T - Y_{J-B} T_J^T          (job a: written, k = bJ + B)
//     X[:, J + bJ:] -= Y_{J-B} L[J + bJ:, J-B .. J)^T     (job b: accumulated, k = B)
// depend only on the previous step, not on each other: a posterior variance at N = 1024 is 4
// launches instead of 7.  A job: C tile 32 x 32 per workgroup, the four waves split the
// concatenated k range [P1 Q1 | P2 Q2] (k1, k2 multiples of 64), meet in LDS, wave w owns
// 16 x 16 block w.  grid (rows / 32, a.ny + b.ny).
// ---------------------------------------------------------------------------
// (struct RowsJob: types.h)
// one 32 x 32 tile (bx, by) of a job; red: 32 KiB of LDS
// NW = 8: eight waves split the k range (k1 + k2 a multiple of 128); waves 4-7 hand their
// sums to waves 0-3 through the same 32 KiB before the four-way reduction
template <int NW = 4>
__device__ __forceinline__ void rows_job_tile(const RowsJob &j, int bx, int by,
                                              double (*red)[4][4][64])
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int kq = (j.k1 + j.k2) / NW;
    double4_t acc[2][2];
#pragma unroll
    for (int jb2 = 0; jb2 < 2; ++jb2)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
            acc[jb2][ib] = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int ks = wave * kq; ks < (wave + 1) * kq; ks += 16) {
        // a 16-column chunk lies in one of the two operand pairs (k1 is a multiple of 16)
        const bool p1 = ks < j.k1;
        const int kk = p1 ? ks : ks - j.k1;
        const double *pp = (p1 ? j.P1 : j.P2) + (long)bx * 32 + l15 +
                           (long)(kk + l4) * (p1 ? j.ldp1 : j.ldp2);
        const long qsj = p1 ? j.qsj1 : j.qsj2, qsk = p1 ? j.qsk1 : j.qsk2;
        const long ldp = p1 ? j.ldp1 : j.ldp2;
        const double *qq = (p1 ? j.Q1 : j.Q2) + ((long)by * 32 + l15) * qsj + (long)(kk + l4) * qsk;
        double a[4][2], b[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                a[u][h] = pp[16 * h + (long)(4 * u) * ldp];
                b[u][h] = qq[(long)(16 * h) * qsj + (long)(4 * u) * qsk];
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int jb2 = 0; jb2 < 2; ++jb2)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
                    acc[jb2][ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[u][jb2], a[u][ib],
                                                                        acc[jb2][ib], 0, 0, 0);
    }
    if (NW == 8) {
        if (wave >= 4) {
#pragma unroll
            for (int jb2 = 0; jb2 < 2; ++jb2)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[wave - 4][2 * jb2 + ib][r][lane] = acc[jb2][ib][r];
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int jb2 = 0; jb2 < 2; ++jb2)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[jb2][ib][r] += red[wave][2 * jb2 + ib][r][lane];
        }
        __syncthreads();
    }
    if (NW == 4 || wave < 4) {
#pragma unroll
        for (int jb2 = 0; jb2 < 2; ++jb2)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[wave][2 * jb2 + ib][r][lane] = acc[jb2][ib][r];
    }
    __syncthreads();
    if (NW == 8 && wave >= 4)
        return;
    const int jq = wave >> 1, iq = wave & 1;
    double *cp = j.C + (long)bx * 32 + 16 * iq + l15 + ((long)by * 32 + 16 * jq + l4) * j.ldc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double s = (red[0][wave][r][lane] + red[1][wave][r][lane]) +
                         (red[2][wave][r][lane] + red[3][wave][r][lane]);
        if (j.write)
            cp[(long)(4 * r) * j.ldc] = -s;
        else
            cp[(long)(4 * r) * j.ldc] -= s;
    }
}

template <int NW = 4>
__global__ __launch_bounds__(64 * NW) void rows_step_kernel(RowsJob ja, RowsJob jb)
{
    __shared__ double red[4][4][4][64];
    // (two calls: selecting the argument struct dynamically parks a copy of it in scratch)
    if ((int)blockIdx.y < ja.ny)
        rows_job_tile<NW>(ja, blockIdx.x, blockIdx.y, red);
    else
        rows_job_tile<NW>(jb, blockIdx.x, blockIdx.y - ja.ny, red);
}

// ---------------------------------------------------------------------------
// One step of the row sweep over a LARGE resident factor in one launch: the small, latency-
// bound product of the step (job `ja`, the diagonal block's solve with the T_J / U_J coupling:
// 32 x 32 split-k tiles as in rows_step_kernel) in the first nd workgroups, and the previous
// block's solution applied to everything beyond (C -= P Q^T on the 64 x 64 LDS-staged tiles of
// gemm_lds64_kernel) in the rest.  The two depend on the previous step only, not on each other;
// in one grid the short job's workgroups are dispatched first and finish beside the update's --
// on a stream of their own they waited for slots behind them.  grid (nd + (m / 64) (n / 64)).
// ---------------------------------------------------------------------------
// KS = 2: eight waves per workgroup, both kinds of job split k twice as far (for steps whose
// grid gives a CU at most two workgroups: the step's time is then ONE workgroup's k loop).
template <bool QT, int KS = 1>
__global__ __launch_bounds__(256 * KS, 4 / KS) void rows_fused_kernel(
    RowsJob ja, int nd, int ndx, double *__restrict__ C, long ldc, const double *__restrict__ P,
    long ldp, const double *__restrict__ Q, long ldq, int m, int n, int k)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int bid = blockIdx.x;
    if (bid < nd) {
        rows_job_tile<4 * KS>(ja, bid % ndx, bid / ndx,
                              reinterpret_cast<double (*)[4][4][64]>(smem));
        return;
    }
    const int tix = bid - nd, mt = m / 64;
    gemm_lds64_body<QT, KS>(smem, C, ldc, P, ldp, Q, ldq, m, n, k, 0, 0x7fffffff, tix % mt,
                            tix / mt);
}
