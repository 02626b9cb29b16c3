// moments.h -- closed-form gauss_c integrals, BQ moments, batched active-sampling systems
// Part of the libbqhip.so kernel set; compiled into moments.hip (host.h lists the units).
#pragma once
#include "common.h"

// ===========================================================================
// Closed-form Gaussian-kernel integrals (gauss_c.pyx) and the BQ moments that
// consume them (bq_c.pyx:157-213,264-355).  Each result is
//     scale * exp( log N(z | 0, C) + per-point terms ),
// with z a D-vector built from one or two points.  The host supplies the
// inverse Cholesky factor of the small D x D covariance (D <= 16), so the
// Mahalanobis term is || Linv z ||^2 with no division on the device.
// ===========================================================================
// (struct GaussForm<D>: types.h)

template <int D>
__device__ __forceinline__ double gauss_form_eval(const GaussForm<D> &f, const double (&z)[D])
{
    double maha = 0.0;
#pragma unroll
    for (int r = 0; r < D; ++r) {
        double y = 0.0;
#pragma unroll
        for (int c = 0; c <= r; ++c)
            y += f.linv[r * D + c] * z[c];
        maha += y * y;
    }
    return f.logc - 0.5 * maha;
}

// out_i = scale * exp(add + log N(x_i - mu | 0, C)); also, if alpha != null,
// accumulates sum_i out_i alpha_i into acc[0] (Z_mean) -- one block per 256 points
template <int D>
__global__ __launch_bounds__(256) void int_K_kernel(const double *__restrict__ x, int n,
                                                    GaussForm<D> f, double scale, double add,
                                                    double *__restrict__ out,
                                                    const double *__restrict__ alpha,
                                                    double *__restrict__ acc)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (i < n) {
        double z[D];
#pragma unroll
        for (int k = 0; k < D; ++k)
            z[k] = x[k + (long)i * D] - f.mu[k];
        v = scale * exp(add + gauss_form_eval<D>(f, z));
        if (out)
            out[i] = v;
        if (alpha)
            v *= alpha[i];
    }
    if (acc) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_down(v, off, 64);
        __shared__ double part[4];
        if ((threadIdx.x & 63) == 0)
            part[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0)
            acc[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
    }
}

// out_ij = scale * exp(log N([x1_i - mu; x2_j - mu] | 0, C2)), n1 x n2 column-major.
// With alpha (length n2) and beta (length n1): beta_i = sum_j out_ij alpha_j is
// accumulated instead of (or besides) storing the matrix; one block = 64 rows,
// its four waves split the columns.
template <int D>
__global__ __launch_bounds__(256) void int_K1_K2_kernel(const double *__restrict__ x1, int n1,
                                                        const double *__restrict__ x2, int n2,
                                                        GaussForm<2 * D> f, double scale,
                                                        double *__restrict__ out,
                                                        const double *__restrict__ alpha,
                                                        double *__restrict__ beta)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    double z[2 * D];
    const bool ok = i < n1;
#pragma unroll
    for (int k = 0; k < D; ++k)
        z[k] = ok ? x1[k + (long)i * D] - f.mu[k] : 0.0;
    double acc = 0.0;
    for (int j = wave; j < n2; j += 4) {
#pragma unroll
        for (int k = 0; k < D; ++k)
            z[D + k] = x2[k + (long)j * D] - f.mu[D + k];
        const double v = scale * exp(gauss_form_eval<2 * D>(f, z));
        if (out && ok)
            out[i + (long)j * n1] = v;
        if (alpha)
            acc += v * alpha[j];
    }
    if (beta) {
        __shared__ double part[4][64];
        part[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && ok)
            beta[i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    }
}

// out_ij = scale * exp(n1_i + n1_j + log N(b_i - b_j | 0, C)), n x n; with alpha the
// bilinear form sum_ij alpha_i alpha_j out_ij goes to acc[block] instead.
template <int D>
__global__ __launch_bounds__(256) void int_int_K1_K2_K1_kernel(const double *__restrict__ bpts,
                                                               const double *__restrict__ n1v,
                                                               int n, GaussForm<D> f, double scale,
                                                               double *__restrict__ out,
                                                               const double *__restrict__ alpha,
                                                               double *__restrict__ acc)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const bool ok = i < n;
    double bi[D];
#pragma unroll
    for (int k = 0; k < D; ++k)
        bi[k] = ok ? bpts[k + (long)i * D] : 0.0;
    const double ni = ok ? n1v[i] : 0.0;
    const double ai = (alpha && ok) ? alpha[i] : 0.0;
    double sum = 0.0;
    const int j0 = blockIdx.y * 256;
    const int j1 = (j0 + 256 < n) ? j0 + 256 : n;
    for (int j = j0 + wave; j < j1; j += 4) {
        double z[D];
#pragma unroll
        for (int k = 0; k < D; ++k)
            z[k] = bi[k] - bpts[k + (long)j * D];
        const double v = scale * exp(ni + n1v[j] + gauss_form_eval<D>(f, z));
        if (out && ok)
            out[i + (long)j * n] = v;
        if (alpha)
            sum += ai * v * alpha[j];
    }
    if (acc) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            sum += __shfl_down(sum, off, 64);
        __shared__ double part[4];
        if (lane == 0)
            part[wave] = sum;
        __syncthreads();
        if (threadIdx.x == 0)
            acc[blockIdx.x + (long)blockIdx.y * gridDim.x] = (part[0] + part[1]) + (part[2] + part[3]);
    }
}

// b_i = G x_i (D x D, row-major G), n1_i = log N(x_i - mu | 0, C1)
template <int D>
__global__ __launch_bounds__(256) void iikk_prepare_kernel(const double *__restrict__ x, int n,
                                                           GaussForm<D> f1, GaussForm<D> g,
                                                           double *__restrict__ bpts,
                                                           double *__restrict__ n1v)
{
    // g.linv carries the full D x D matrix G (row-major), g.mu / g.logc unused
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    double xi[D], z[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        xi[k] = x[k + (long)i * D];
        z[k] = xi[k] - f1.mu[k];
    }
    n1v[i] = gauss_form_eval<D>(f1, z);
#pragma unroll
    for (int r = 0; r < D; ++r) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c)
            s += g.linv[r * D + c] * xi[c];
        bpts[r + (long)i * D] = s;
    }
}

// out[0] = sum_k v[k]  (k < n), one block; out[1] = sum_k u[k] v[k] if u
__global__ __launch_bounds__(256) void reduce_sum_kernel(const double *__restrict__ v,
                                                         const double *__restrict__ u, int n,
                                                         double *__restrict__ out)
{
    const int t = threadIdx.x;
    double s = 0.0;
    for (int k = t; k < n; k += 256)
        s += u ? u[k] * v[k] : v[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off, 64);
    __shared__ double part[4];
    if ((t & 63) == 0)
        part[t >> 6] = s;
    __syncthreads();
    if (t == 0)
        out[0] = (part[0] + part[1]) + (part[2] + part[3]);
}

// ===========================================================================
// Batched expected-squared-mean systems (bq.py:447-527, bq_c.pyx:425-535).
// Batch element a is the Gram of the nsc points x_sc plus the candidate x_a[a]
// (no noise term: gp.Kxoxo), with the reference's jitter on the diagonal --
// jit1[a] on the candidates within `thresh` of x_a[a], jit2[a] on the new point
// (bq_c.pyx:127-140) -- bordered by two rows: int K(x_sca) p(x) dx and [l_sc, 0].
// After eliminating the npad columns, A_a and A_sc . l_sc are read off the panel
// and the Schur complement (esm_finalize_kernel); no back substitution.
// ===========================================================================
// (struct EsmLayout: types.h)

__global__ __launch_bounds__(256) void assemble_esm_kernel(const double *__restrict__ x_sc,
                                                           const double *__restrict__ x_a,
                                                           const double *__restrict__ intk_sc,
                                                           const double *__restrict__ intk_a,
                                                           const double *__restrict__ l_sc,
                                                           const double *__restrict__ jit1,
                                                           const double *__restrict__ jit2,
                                                           double thresh, GaussParams g,
                                                           double *__restrict__ A, long lda,
                                                           long astride, EsmLayout L)
{
    const int b = blockIdx.z;
    const int t = threadIdx.x;
    const int ib = blockIdx.x * 128, jb = blockIdx.y * 64;
    if (jb > ib + 127)
        return;
    A += (long)b * astride;
    const double xa = x_a[b];
    const int n1 = L.nsc + 1;
    const int i = ib + (t & 63) * 2;
    const int jbase = jb + (t >> 6) * 16;
    if (i >= L.ntot)
        return;
    double xi[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ii = i + r;
        xi[r] = ii < L.nsc ? x_sc[ii] : xa;
    }
    for (int jj = 0; jj < 16; ++jj) {
        const int j = jbase + jj;
        if (j >= L.ntot)
            break;
        const double xj = j < L.nsc ? x_sc[j] : xa;
        double v[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ii = i + r;
            double val;
            if (ii < n1 && j < n1) {
                const double tdiff = xi[r] - xj;
                val = g.c * exp_gauss((tdiff * tdiff) * g.nh[0]);
                if (ii == j) {
                    if (ii == L.nsc)
                        val += jit2[b];
                    else if (ii >= L.ns && fabs(xi[r] - xa) < thresh)
                        val += jit1[b];
                }
            } else if (ii == L.npad) {
                val = j < L.nsc ? intk_sc[j] : (j == L.nsc ? intk_a[b] : 0.0);
            } else if (ii == L.npad + 1) {
                val = j < L.nsc ? l_sc[j] : 0.0;
            } else {
                val = (ii == j) ? 1.0 : 0.0;
            }
            v[r] = val;
        }
        double2_t vv = {v[0], v[1]};
        *reinterpret_cast<double2_t *>(A + i + (long)j * lda) = vv;
    }
}

// out[2b] = A_a = (K^-1 intK)[last], out[2b+1] = A_sc . l_sc
__global__ void esm_finalize_kernel(const double *__restrict__ A, long lda, long astride,
                                    EsmLayout L, int batch, double *__restrict__ out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch)
        return;
    const double *Ab = A + (long)b * astride;
    const double z_last = Ab[L.npad + (long)L.nsc * lda];
    const double l_last = Ab[L.nsc + (long)L.nsc * lda];
    // A = L^-T z: the last component is z_last / L_nn
    out[2 * b] = z_last / l_last;
    // (L^-1 [l_sc, 0]) . z  sits, negated, in the Schur complement at (npad+1, npad)
    out[2 * b + 1] = -Ab[(L.npad + 1) + (long)L.npad * lda];
}
