// trsm.h -- the MFMA panel solve from 16 x 16 block inverses, and those inverses for a resident factor
// Part of the libbqhip.so kernel set; compiled into k_panel.hip (host.h lists the units).
#pragma once
#include "common.h"
#include "potf2.h"

// ---------------------------------------------------------------------------
// Panel solve on the matrix cores: X (m x 64) <- X L11^-T in four 16-column block steps,
//   X_c = (A_c - sum_{b<c} X_b L_cb^T) W_cc^T,   W_cc = L_cc^-1 (from potf2f_body),
// instead of 64 dependent column steps.  A wave owns 16 rows and works on transposes, so
// that every intermediate stays in MFMA operand form: with D = A B on
// v_mfma_f64_16x16x4_f64, register r of the D tile IS the B fragment of k-step r, so
// X_b^T (a D tile) feeds T_c^T -= L_cb X_b^T and X_c^T = W_cc T_c^T directly; L_cb and W_cc
// are A fragments read from global memory (lane l: row l & 15, k = (l >> 4) + 4 r).
// 40 MFMAs per wave.  m is a multiple of 16.  The inverse of a 16 x 16 diagonal block of
// a Cholesky factor is as well conditioned as the block (MAGMA's trtri-based trsm does
// the same with 128-wide blocks); the parity bars of tests/test_gpu_parity.py hold.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void trsm_blk_kernel(double *__restrict__ X, long ldx,
                                                       long xstride, int m,
                                                       const double *__restrict__ Lm, long ldl,
                                                       long lstride,
                                                       const double *__restrict__ dinv,
                                                       long dstride)
{
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * 64 + wave * 16;
    if (row0 >= m)
        return;
    const int l15 = lane & 15, l4 = lane >> 4;
    double *Xr = X + (long)b * xstride + row0 + l15 + (long)l4 * ldx;
    const double *L11 = Lm + (long)b * lstride + l15 + (long)l4 * ldl;
    const double *W = dinv + (long)b * dstride + 64 + l15 + 16 * l4;
    // t[c][r] = X[row0 + l15][16 c + l4 + 4 r]: B fragments of the transposed row block
    double t[4][4], w[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t[c][r] = Xr[(long)(16 * c + 4 * r) * ldx];
            w[c][r] = W[256 * c + 64 * r]; // W_c[l15][l4 + 4 r]
        }
    // la[c][b][r] = L11[16 c + l15][16 b + l4 + 4 r], c > b
    double la[4][3][4];
#pragma unroll
    for (int c = 1; c < 4; ++c)
#pragma unroll
        for (int bb = 0; bb < c; ++bb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                la[c][bb][r] = L11[16 * c + (long)(16 * bb + 4 * r) * ldl];
    double4_t x[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // acc = sum_b L_cb X_b^T - T_c^T
        double4_t acc = {-t[c][0], -t[c][1], -t[c][2], -t[c][3]};
#pragma unroll
        for (int bb = 0; bb < c; ++bb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(la[c][bb][r], x[bb][r], acc, 0, 0, 0);
        // X_c^T = -W_cc acc
        double4_t xc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < 4; ++r)
            xc = __builtin_amdgcn_mfma_f64_16x16x4f64(-w[c][r], acc[r], xc, 0, 0, 0);
        x[c] = xc;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Xr[(long)(16 * c + 4 * r) * ldx] = xc[r];
    }
}

// For a RESIDENT factor: the scratch record trsm_blk_kernel wants -- 64 reciprocal pivots and
// the inverses of the four 16 x 16 diagonal sub-blocks -- of every 64 x 64 diagonal block,
// BQ_DINV_HALF doubles per block, so that the row-form sweeps over the factor (predictions,
// solves) can use the MFMA panel solve too.  One workgroup per diagonal block, wave w inverts
// sub-block w exactly as potf2f_body does.
__global__ __launch_bounds__(256) void diag_winv_kernel(const double *__restrict__ Lm, long ldl,
                                                        double *__restrict__ dw)
{
    __shared__ __attribute__((aligned(16))) double blk[4][256];
    __shared__ double rd[64];
    const int jb = 64 * blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double *Lw = Lm + jb + 16 * w + (long)(jb + 16 * w) * ldl;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int e = lane + 64 * t; // element i = e & 15, column k = e >> 4
        blk[w][e] = Lw[(e & 15) + (long)(e >> 4) * ldl];
    }
    if (lane < 16)
        rd[16 * w + lane] = 1.0 / Lw[lane + (long)lane * ldl];
    __syncthreads();
    double *out = dw + (long)blockIdx.x * BQ_DINV_HALF;
    if (lane < 16) {
        out[16 * w + lane] = rd[16 * w + lane];
        double wc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double sacc[4] = {(i == lane) ? 1.0 : 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < i; ++k)
                sacc[k & 3] -= blk[w][i + 16 * k] * wc[k];
            wc[i] = ((sacc[0] + sacc[1]) + (sacc[2] + sacc[3])) * rd[16 * w + i];
        }
        double *Wb = out + 64 + 256 * w + 16 * lane;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            Wb[i] = wc[i];
    }
}

// ---------------------------------------------------------------------------
// C(m x n) -= P(m x k) * Q(n x k)^T on v_mfma_f64_16x16x4_f64.
//
// P element (i,kk) at P[i + kk*ldp].  Q element (j,kk) at Q[j*qsj + kk*qsk]:
//   (qsj,qsk) = (1, ldq)  -> Q^T product (Cholesky panel / trailing update)
//   (qsj,qsk) = (ldq, 1)  -> plain product with a k x n matrix (L^T sweep)
// A workgroup is 4 waves in a 2 x 2 arrangement; each wave owns a
// (16 TM) x (16 TN) tile built from TM x TN MFMA tiles and streams its A/B
// fragments straight from global memory (L2-resident panel) into registers,
// software-pipelined one k-step (4 columns) ahead.  The MFMA is issued as
// D^T = Q_frag * P_frag^T so that the 16 lanes that share a D register row
// cover 16 consecutive ROWS of C: the read-modify-write of C then moves whole
// 128-byte lines (C is column-major).
//   f64 16x16x4 operand map: lane l supplies A[l&15][l>>4] and B[l>>4][l&15];
//   D register r of lane l is D[(l>>4) + 4 r][l & 15]   (probed on gfx950 by
//   bq_probe_mfma_layout; tests/test_gpu_probe.py asserts it).
// lower != 0: skip wave tiles that lie strictly above the diagonal of C
// (C square, trailing update); m, n multiples of 16 TM / 16 TN are not
// required, out-of-range wave tiles exit, but m and n must be multiples of 16
// and k a multiple of 8.
// ---------------------------------------------------------------------------

// the 64 x 64 diagonal factor as a launch of its own (potf2.h has the body)
template <int NW = 4>
__global__ __launch_bounds__(64 * NW) void potf2_kernel(double *__restrict__ A, long lda,
                                                        long astride, int j0,
                                                        double *__restrict__ dinv, long dstride,
                                                        int *__restrict__ info)
{
    __shared__ __attribute__((aligned(16))) double lds[BQ_POTF2_LDS_DOUBLES];
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.z;
    potf2_body<NW>(A + (long)b * astride + j0 + (long)j0 * lda, lda, j0,
                   dinv + (long)b * dstride, info + b, lds);
}
