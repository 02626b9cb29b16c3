// gemm.h -- C -= P Q^T on v_mfma_f64_16x16x4_f64 (panel and trailing updates)
// Part of the libbqhip.so kernel set; included through kernels.h.
#pragma once
#include "common.h"

// 1-D grid over the lower-triangular workgroup tiles of a square update:
// t -> (bx, by), by <= bx, row by row, so no empty workgroups are launched (at
// N=16384 the 2-D grid's early-exit workgroups cost 8 % of the trailing update)
__device__ __forceinline__ void tri_decode(int t, int &bx, int &by)
{
    bx = (int)((__builtin_sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((bx + 1) * (bx + 2) / 2 <= t)
        ++bx;
    while (bx * (bx + 1) / 2 > t)
        --bx;
    by = t - bx * (bx + 1) / 2;
}

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
                *dst -= acc[tm][tn][rr];
            }
        }
    }
}

// Fused diagonal factor: when fuse_j0 >= 0 the launch also factors the leading 64x64
// block of C (the next diagonal block of the Cholesky) right after updating it.
// Workgroup 0 owns every workgroup tile that intersects that block, updates them,
// and runs potf2_64x4_body on the result; the other workgroups of the block exit.
// This removes one dependent launch (and the block's trip through L2) per 64 columns.
template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void gemm_sub_kernel(double *__restrict__ C, long ldc,
                                                          long cstride, const double *__restrict__ P,
                                                          long ldp, long pstride,
                                                          const double *__restrict__ Q, long qsj,
                                                          long qsk, long qstride, int m, int n,
                                                          int k, int lower, int fuse_j0,
                                                          double *__restrict__ dinv, long dstride,
                                                          int *__restrict__ info)
{
    __shared__ __attribute__((aligned(16))) double ring[3 * 4 * 64];
    __shared__ int sbad[4];
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
                if (row0 < m && col0 < n && !(lower && col0 >= row0 + TM * 16))
                    gemm_sub_tile<TM, TN>(C, ldc, P, ldp, Q, qsj, qsk, m, n, k, lower, row0, col0,
                                          lane);
            }
        __syncthreads(); // the updated block is visible to the whole workgroup
        potf2_64x4_body(C, ldc, fuse_j0, dinv + (long)b * dstride, info + b, ring, sbad);
        return;
    }
    const int row0 = (bx * 2 + (wave & 1)) * (TM * 16);
    const int col0 = (by * 2 + (wave >> 1)) * (TN * 16);
    if (row0 >= m || col0 >= n)
        return;
    if (lower && col0 >= row0 + TM * 16)
        return;
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
    __shared__ __attribute__((aligned(16))) double ring[3 * 4 * 64];
    __shared__ int sbad[4];
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
        potf2_64x4_body(C, ldc, fuse_j0, dinv + (long)b * dstride, info + b, ring, sbad);
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
