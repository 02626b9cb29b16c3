// reduce.h -- read-out, reductions, fused prediction, small utility kernels
// Part of the libbqhip.so kernel set; compiled into k_reduce.hip (host.h lists the units).
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------
// Read-out after the bordered elimination (one block per problem):
//   logdet = 2 sum_{i<n} log L_ii,  qf = -S[yrow,yrow] = y' Kxx^-1 y,
//   logml  = -qf/2 - logdet/2 - n/2 log 2 pi,
//   mean_i = -S[yrow, i],  var_i = S[i,i]   (S = Schur complement at npad)
// scal[b*4 + {0,1,2}] = logml, logdet, qf.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finalize_kernel(const double *__restrict__ A, long lda,
                                                       long astride, Layout L,
                                                       double *__restrict__ scal,
                                                       double *__restrict__ mean,
                                                       double *__restrict__ var, long mstride)
{
    const int b = blockIdx.z;
    A += (long)b * astride;
    const int t = threadIdx.x;
    double s = 0.0;
    for (int i = t; i < L.n; i += 256)
        s += log(A[i + (long)i * lda]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off, 64);
    __shared__ double part[4];
    if ((t & 63) == 0)
        part[t >> 6] = s;
    __syncthreads();
    if (t == 0) {
        const double logdet = 2.0 * (part[0] + part[1] + part[2] + part[3]);
        double qf = 0.0, logml = 0.0;
        if (L.yrow >= 0) {
            qf = -A[L.yrow + (long)L.yrow * lda];
            logml = -0.5 * qf - 0.5 * logdet - 0.5 * (double)L.n * 1.8378770664093453; // log 2pi
        }
        scal[b * 4 + 0] = logml;
        scal[b * 4 + 1] = logdet;
        scal[b * 4 + 2] = qf;
    }
    for (int i = t; i < L.M; i += 256) {
        const long c = L.npad + i;
        if (var)
            var[(long)b * mstride + i] = A[c + c * lda];
        if (mean && L.yrow >= 0)
            mean[(long)b * mstride + i] = -A[L.yrow + c * lda];
    }
}

// per-row reductions of a solved border V (m x n, ld = ldv), z (n):
//   mean_i = sum_j V[i,j] z[j],   var_i = k0 - sum_j V[i,j]^2
// block = 16 rows x 64 column slices (1024 threads; a wave reads four columns x 128 bytes),
// eight loads of a slice in flight: with 64 rows x 4 slices and one load at a time a 256-row
// border took 78 us of a 200 us prediction.  grid: ceil(m / 16).
__global__ __launch_bounds__(1024) void rowdot_kernel(const double *__restrict__ V, long ldv,
                                                      int m, int n, const double *__restrict__ z,
                                                      long zs, double k0,
                                                      double *__restrict__ mean,
                                                      double *__restrict__ var)
{
    // (zs: stride of z -- the fit's z lives in a row of its factor)
    const int t = threadIdx.x, r = t & 15, sl = t >> 4;
    const int row = blockIdx.x * 16 + r;
    double sm = 0.0, sv = 0.0;
    if (row < m) {
        const double *p = V + row;
        int j = sl;
        // (sixteen columns per trip: at n = 1024 a thread's whole share is in flight at once --
        // the kernel is two memory round trips long, not bandwidth; 12.7 -> see LABBOOK round 5)
        for (; j + 64 * 15 < n; j += 64 * 16) {
            double v[16], zz[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                v[u] = p[(long)(j + 64 * u) * ldv];
                zz[u] = z ? z[(long)(j + 64 * u) * zs] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                sm = fma(v[u], zz[u], sm);
                sv = fma(v[u], v[u], sv);
            }
        }
        for (; j + 64 * 7 < n; j += 64 * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = p[(long)(j + 64 * u) * ldv];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (z)
                    sm = fma(v[u], z[(long)(j + 64 * u) * zs], sm);
                sv = fma(v[u], v[u], sv);
            }
        }
        for (; j < n; j += 64) {
            const double v = p[(long)j * ldv];
            if (z)
                sm = fma(v, z[(long)j * zs], sm);
            sv = fma(v, v, sv);
        }
    }
    __shared__ double pm[64][17], pv[64][17];
    pm[sl][r] = sm;
    pv[sl][r] = sv;
    __syncthreads();
    if (t < 16 && row < m) {
        double a = 0.0, q = 0.0;
        for (int w = 0; w < 64; ++w) {
            a += pm[w][r];
            q += pv[w][r];
        }
        if (mean)
            mean[row] = a;
        if (var)
            var[row] = k0 - q;
    }
}

// -I on the diagonal blocks of the wide-inverse array (block J of width B at nr + J B, ld B)
__global__ void neg_identity_kernel(double *nr, int B, int npad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x; // global row / column
    if (i < npad) {
        const int J = i / B * B, a = i - J;
        nr[(size_t)J * B + a + (size_t)a * B] = -1.0;
    }
}

// mean_i = sum_j k(xo_i, x_j) alpha_j : fused cross-Gram x GEMV, one wave per
// 64 outputs?  No: one block of 256 threads per output point slice would
// starve; use one wave per output point, lanes stride over j.
template <int D>
__global__ __launch_bounds__(256) void predict_mean_kernel(const double *__restrict__ xo, int M,
                                                           const double *__restrict__ x, int n,
                                                           const double *__restrict__ alpha,
                                                           GaussParams g, double *__restrict__ mean)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= M)
        return;
    double p[D];
#pragma unroll
    for (int k = 0; k < D; ++k)
        p[k] = xo[k + (long)i * D];
    double s = 0.0;
    for (int j = lane; j < n; j += 64) {
        double q[D];
#pragma unroll
        for (int k = 0; k < D; ++k)
            q[k] = x[k + (long)j * D];
        s += exp_gauss(gauss_q<D>(p, q, g)) * alpha[j];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off, 64);
    if (lane == 0)
        mean[i] = g.c * s;
}

// dst(rows x cols, ld ldd) <- src(rows x cols, ld lds); optional transpose
__global__ void copy2d_kernel(double *__restrict__ dst, long ldd, const double *__restrict__ src,
                              long lds, int rows, int cols, int transpose_src)
{
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j0 = blockIdx.y * 16 + (threadIdx.x >> 6) * 4;
    if (i >= rows)
        return;
    for (int j = j0; j < j0 + 4 && j < cols; ++j)
        dst[i + (long)j * ldd] = transpose_src ? src[j + (long)i * lds] : src[i + (long)j * lds];
}

// A <- identity on the padding square [n, ntot) and zero in the padding
// rows/cols of the lower triangle (used by the linalg drop-ins)
__global__ void pad_identity_kernel(double *__restrict__ A, long lda, int n, int ntot)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= ntot || j >= ntot)
        return;
    if (i >= n || j >= n)
        A[i + (long)j * lda] = (i == j) ? 1.0 : 0.0;
}

// 2 sum log diag, one block
__global__ __launch_bounds__(256) void logdet_kernel(const double *__restrict__ A, long lda, int n,
                                                     double *__restrict__ out)
{
    const int t = threadIdx.x;
    double s = 0.0;
    for (int i = t; i < n; i += 256)
        s += log(A[i + (long)i * lda]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off, 64);
    __shared__ double part[4];
    if ((t & 63) == 0)
        part[t >> 6] = s;
    __syncthreads();
    if (t == 0)
        out[0] = 2.0 * (part[0] + part[1] + part[2] + part[3]);
}

// dst[c + r ldd] = src[r + c lds] for r < rows, c < cols (64 x 64 tiles through LDS): the
// right-hand sides of a solve go into the row form of the sweeps and back on the device --
// the same loop on the host took 0.3 s for a 4096 x 4096 block.  grid (ceil(rows / 64),
// ceil(cols / 64)); whatever lies outside is left alone.
__global__ __launch_bounds__(256) void transpose_pad_kernel(const double *__restrict__ src,
                                                            long lds, int rows, int cols,
                                                            double *__restrict__ dst, long ldd)
{
    __shared__ double tile[64][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    for (int cc = wave; cc < 64; cc += 4)
        if (r0 + lane < rows && c0 + cc < cols)
            tile[cc][lane] = src[r0 + lane + (long)(c0 + cc) * lds];
    __syncthreads();
    for (int rr = wave; rr < 64; rr += 4)
        if (c0 + lane < cols && r0 + rr < rows)
            dst[c0 + lane + (long)(r0 + rr) * ldd] = tile[lane][rr];
}

// Read-out of a bordered elimination WITHOUT its border x border block (one launch per batch).
// After the npad columns are eliminated the border rows hold V = K(xo, x) L^-T and, as row
// yrow = npad + M, z = L^-1 y:
//   mean_i = v_i . z,   var_i = k(xo_i, xo_i) - |v_i|^2,   qf = |z|^2,
//   logdet = 2 sum log L_ii,   logml = -qf/2 - logdet/2 - n/2 log 2 pi
// -- the same numbers the Schur complement's diagonal and y row carry, so the trailing updates
// of a batch can skip that block (gemm_lds_kernel's ncut: 6-13 % of a C5 update's tiles).
// Block = 16 rows x 64 column slices like rowdot_kernel; rows npad .. npad + M (the y row is
// the last); the block that holds the y row also reduces the diagonal.  grid (ceil((M+1)/16),
// 1, batch).  scal[b*4 + {0,1,2}] = logml, logdet, qf.
__global__ __launch_bounds__(1024) void plan_readout_kernel(const double *__restrict__ A, long lda,
                                                            long astride, Layout L,
                                                            const GaussParams *__restrict__ gp,
                                                            double *__restrict__ scal,
                                                            double *__restrict__ mean,
                                                            double *__restrict__ var, long mstride)
{
    const int b = blockIdx.z;
    A += (long)b * astride;
    const int t = threadIdx.x, r = t & 15, sl = t >> 4;
    const int i = blockIdx.x * 16 + r; // border row index, M = the y row
    const double *zrow = A + L.yrow;
    double sm = 0.0, sv = 0.0;
    if (i <= L.M) {
        const double *p = A + L.npad + i;
        int j = sl;
        for (; j + 64 * 7 < L.npad; j += 64 * 8) {
            double v[8], z[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = p[(long)(j + 64 * u) * lda];
                z[u] = zrow[(long)(j + 64 * u) * lda];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sm = fma(v[u], z[u], sm);
                sv = fma(v[u], v[u], sv);
            }
        }
        for (; j < L.npad; j += 64) {
            const double v = p[(long)j * lda];
            sm = fma(v, zrow[(long)j * lda], sm);
            sv = fma(v, v, sv);
        }
    }
    __shared__ double pm[64][17], pv[64][17];
    __shared__ double part[16];
    pm[sl][r] = sm;
    pv[sl][r] = sv;
    __syncthreads();
    double a = 0.0, q = 0.0;
    if (t < 16 && i <= L.M) {
        for (int w = 0; w < 64; ++w) {
            a += pm[w][r];
            q += pv[w][r];
        }
        if (i < L.M) {
            if (mean)
                mean[(long)b * mstride + i] = a;
            if (var)
                var[(long)b * mstride + i] = gp[b].c - q;
        }
    }
    // the block of the y row: log-det and the scalars
    if ((int)blockIdx.x * 16 <= L.M && L.M < (int)blockIdx.x * 16 + 16) {
        double s = 0.0;
        for (int k = t; k < L.n; k += 1024)
            s += log(A[k + (long)k * lda]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            s += __shfl_down(s, off, 64);
        if ((t & 63) == 0)
            part[t >> 6] = s;
        __syncthreads();
        if (t == (L.M & 15)) { // the thread that holds the y row's sums (q = |z|^2)
            double ld = 0.0;
            for (int w = 0; w < 16; ++w)
                ld += part[w];
            ld *= 2.0;
            scal[b * 4 + 0] = -0.5 * q - 0.5 * ld - 0.5 * (double)L.n * 1.8378770664093453;
            scal[b * 4 + 1] = ld;
            scal[b * 4 + 2] = q;
        }
    }
}
