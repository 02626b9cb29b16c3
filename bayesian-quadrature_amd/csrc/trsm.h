// trsm.h -- panel solves X <- X L11^-T / X L11^-1 (row-per-lane and quad-lane variants)
// Part of the libbqhip.so kernel set; included through kernels.h.
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------
// Panel solve, one row per lane, 64 columns in registers; one wave per block.
//   TRANS = true : X <- X * L11^-T   (forward substitution; Cholesky panel,
//                                     forward solves with rows = right-hand sides)
//   TRANS = false: X <- X * L11^-1   (backward substitution; the L^T sweep)
// X = rows of the panel (leading dimension ldx), L11 = 64x64 lower block
// (leading dimension ldl) with reciprocal diagonal dinv[64].  L11 is staged
// once into LDS (TRANS: as stored; else transposed) and its entries are read
// back as wave-wide broadcasts, two per ds_read_b128.
// ---------------------------------------------------------------------------
template <bool TRANS>
__global__ __launch_bounds__(64) void trsm_rows_kernel(double *__restrict__ X, long ldx,
                                                       long xstride, int m,
                                                       const double *__restrict__ Lm, long ldl,
                                                       long lstride,
                                                       const double *__restrict__ dinv,
                                                       long dstride)
{
    // T[p][j]: multiplier of x_p in the update of x_j, rows of 64 doubles
    __shared__ __attribute__((aligned(16))) double T[64 * 64];
    __shared__ __attribute__((aligned(16))) double di[64];
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.z;
    const int lane = threadIdx.x;
    const int row = blockIdx.x * 64 + lane;
    X += (long)b * xstride;
    const double *L11 = Lm + (long)b * lstride;
    if (TRANS) {
        // x_j -= L11[j][p] x_p (j > p): T[p][j] = L11[j + p ldl], coalesced copy
#pragma unroll 8
        for (int p = 0; p < 64; ++p)
            T[p * 64 + lane] = L11[lane + (long)p * ldl];
    } else {
        // x_j -= L11[p][j] x_p (j < p): T[p][j] = L11[p + j ldl]; lane = p keeps
        // the global read coalesced, the LDS write is strided (once per block)
#pragma unroll 8
        for (int j = 0; j < 64; ++j)
            T[lane * 64 + j] = L11[lane + (long)j * ldl];
    }
    di[lane] = dinv[(long)b * dstride + lane];
    const bool ok = row < m;
    double x[64];
    {
        const double *pr = X + (ok ? row : 0);
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            x[j] = *pr;
            pr += ldx;
        }
    }
    __syncthreads();
    // Row p of T is fetched one column step ahead of its use (T is static), so
    // the LDS latency hides behind the previous step's FMAs.
    double2_t cur[32], nxt[32];
    if (TRANS) {
        {
            const double2_t *t2 = reinterpret_cast<const double2_t *>(T);
#pragma unroll
            for (int kk = 0; kk < 32; ++kk)
                cur[kk] = t2[kk];
        }
#pragma unroll
        for (int p = 0; p < 64; ++p) {
            if (p < 63) {
                const double2_t *t2 = reinterpret_cast<const double2_t *>(T + (p + 1) * 64);
#pragma unroll
                for (int kk = (p + 2) >> 1; kk < 32; ++kk)
                    nxt[kk] = t2[kk];
            }
            const double xp = x[p] * di[p];
            x[p] = xp;
#pragma unroll
            for (int kk = (p + 1) >> 1; kk < 32; ++kk) {
                if (2 * kk >= p + 1)
                    x[2 * kk] -= cur[kk][0] * xp;
                x[2 * kk + 1] -= cur[kk][1] * xp;
            }
#pragma unroll
            for (int j = p + 1; j < 64; ++j)
                PIN(x[j]);
#pragma unroll
            for (int kk = (p + 2) >> 1; kk < 32; ++kk)
                cur[kk] = nxt[kk];
        }
    } else {
        {
            const double2_t *t2 = reinterpret_cast<const double2_t *>(T + 63 * 64);
#pragma unroll
            for (int kk = 0; kk < 32; ++kk)
                cur[kk] = t2[kk];
        }
#pragma unroll
        for (int p = 63; p >= 0; --p) {
            if (p > 0) {
                const double2_t *t2 = reinterpret_cast<const double2_t *>(T + (p - 1) * 64);
#pragma unroll
                for (int kk = 0; 2 * kk < p - 1; ++kk)
                    nxt[kk] = t2[kk];
            }
            const double xp = x[p] * di[p];
            x[p] = xp;
#pragma unroll
            for (int kk = 0; 2 * kk < p; ++kk) {
                x[2 * kk] -= cur[kk][0] * xp;
                if (2 * kk + 1 < p)
                    x[2 * kk + 1] -= cur[kk][1] * xp;
            }
#pragma unroll
            for (int j = 0; j < p; ++j)
                PIN(x[j]);
#pragma unroll
            for (int kk = 0; 2 * kk < p - 1; ++kk)
                cur[kk] = nxt[kk];
        }
    }
    if (ok) {
        double *pw = X + row; // opaque copy: see potf2_64_kernel
        asm volatile("" : "+v"(pw));
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            *pw = x[j];
            pw += ldx;
        }
    }
}

// ---------------------------------------------------------------------------
// Panel solve, FOUR lanes per row (a wave = 16 rows): the latency-oriented
// variant used when the panel is short (few rows per CU).  Lane l works on row
// l>>2 and on the 16 columns {8kk + 2g, 8kk + 2g + 1}, g = l&3, kk = 0..7, so a
// column step costs each lane at most 16 FMAs instead of 63; the solved entry
// x_p is handed to the other three lanes of the quad by DPP quad_perm.  The
// multipliers come from an LDS copy of L11 whose inapplicable entries (j <= p,
// or j >= p for the backward form) are stored as zeros, so the update needs no
// per-lane predicate; lanes with equal g read the same address (broadcast).
// ---------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ double quad_bcast_f64(double v)
{
    constexpr int ctrl = G | (G << 2) | (G << 4) | (G << 6); // quad_perm:[G,G,G,G]
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <bool TRANS, int P>
__device__ __forceinline__ void trsm_quad_step(double (&x)[8][2], const double *T, const double *di,
                                               int g, double2_t (&cur)[8], double2_t (&nxt)[8])
{
    constexpr int KK = P >> 3, GP = (P >> 1) & 3, SL = P & 1;
    constexpr int PN = TRANS ? P + 1 : P - 1; // next column step
    if (PN >= 0 && PN < 64) {
        const double2_t *t2 = reinterpret_cast<const double2_t *>(T + PN * 64);
        if (TRANS) {
#pragma unroll
            for (int kk = (PN >> 3); kk < 8; ++kk)
                nxt[kk] = t2[4 * kk + g];
        } else {
#pragma unroll
            for (int kk = 0; kk <= (PN >> 3); ++kk)
                nxt[kk] = t2[4 * kk + g];
        }
    }
    const double mine = x[KK][SL] * di[P];
    const double xp = quad_bcast_f64<GP>(mine);
    x[KK][SL] = (g == GP) ? xp : x[KK][SL];
    if (TRANS) {
#pragma unroll
        for (int kk = KK; kk < 8; ++kk) {
            x[kk][0] -= cur[kk][0] * xp;
            x[kk][1] -= cur[kk][1] * xp;
        }
#pragma unroll
        for (int kk = KK; kk < 8; ++kk) {
            PIN(x[kk][0]);
            PIN(x[kk][1]);
        }
    } else {
#pragma unroll
        for (int kk = 0; kk <= KK; ++kk) {
            x[kk][0] -= cur[kk][0] * xp;
            x[kk][1] -= cur[kk][1] * xp;
        }
#pragma unroll
        for (int kk = 0; kk <= KK; ++kk) {
            PIN(x[kk][0]);
            PIN(x[kk][1]);
        }
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
        cur[kk] = nxt[kk];
}

template <bool TRANS, int P>
struct TrsmQuadSteps {
    static __device__ __forceinline__ void run(double (&x)[8][2], const double *T, const double *di,
                                               int g, double2_t (&cur)[8], double2_t (&nxt)[8])
    {
        trsm_quad_step<TRANS, TRANS ? P : 63 - P>(x, T, di, g, cur, nxt);
        TrsmQuadSteps<TRANS, P + 1>::run(x, T, di, g, cur, nxt);
    }
};
template <bool TRANS>
struct TrsmQuadSteps<TRANS, 64> {
    static __device__ __forceinline__ void run(double (&)[8][2], const double *, const double *, int,
                                               double2_t (&)[8], double2_t (&)[8])
    {
    }
};

template <bool TRANS>
__global__ __launch_bounds__(64) void trsm_quad_kernel(double *__restrict__ X, long ldx,
                                                       long xstride, int m,
                                                       const double *__restrict__ Lm, long ldl,
                                                       long lstride,
                                                       const double *__restrict__ dinv,
                                                       long dstride)
{
    __shared__ __attribute__((aligned(16))) double T[64 * 64];
    __shared__ __attribute__((aligned(16))) double di[64];
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.z;
    const int lane = threadIdx.x;
    const int g = lane & 3;
    const int row = blockIdx.x * 16 + (lane >> 2);
    X += (long)b * xstride;
    const double *L11 = Lm + (long)b * lstride;
    if (TRANS) {
        // T[p][j] = L11[j][p] for j > p, else 0
#pragma unroll 8
        for (int p = 0; p < 64; ++p) {
            const double v = L11[lane + (long)p * ldl];
            T[p * 64 + lane] = (lane > p) ? v : 0.0;
        }
    } else {
        // T[p][j] = L11[p][j] for j < p, else 0 (lane = p: coalesced global read)
#pragma unroll 8
        for (int j = 0; j < 64; ++j) {
            const double v = L11[lane + (long)j * ldl];
            T[lane * 64 + j] = (j < lane) ? v : 0.0;
        }
    }
    di[lane] = dinv[(long)b * dstride + lane];
    const bool ok = row < m;
    double x[8][2];
    {
        const double *pr = X + (ok ? row : 0) + (long)(2 * g) * ldx;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            x[kk][0] = pr[0];
            x[kk][1] = pr[ldx];
            pr += 8 * ldx;
        }
    }
    __syncthreads();
    double2_t cur[8], nxt[8];
    {
        const double2_t *t2 = reinterpret_cast<const double2_t *>(T + (TRANS ? 0 : 63) * 64);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            cur[kk] = t2[4 * kk + g];
            nxt[kk] = cur[kk];
        }
    }
    TrsmQuadSteps<TRANS, 0>::run(x, T, di, g, cur, nxt);
    if (ok) {
        double *pw = X + row + (long)(2 * g) * ldx;
        asm volatile("" : "+v"(pw));
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            pw[0] = x[kk][0];
            pw[ldx] = x[kk][1];
            pw += 8 * ldx;
        }
    }
}

// ---------------------------------------------------------------------------
// Panel solve on the matrix cores: X (m x 64) <- X L11^-T in four 16-column block steps,
//   X_c = (A_c - sum_{b<c} X_b L_cb^T) W_cc^T,   W_cc = L_cc^-1 (from potf2_64x4_body),
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
// sub-block w exactly as potf2_64x4_body does.
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

// reciprocal diagonal of a resident factor: dinv[j] = 1 / L[j0+j, j0+j]
__global__ void diag_recip_kernel(const double *__restrict__ Lm, long ldl, long lstride, int n,
                                  double *__restrict__ dinv, long dstride)
{
    const int b = blockIdx.z;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n)
        dinv[(long)b * dstride + j] = 1.0 / Lm[(long)b * lstride + j + (long)j * ldl];
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
