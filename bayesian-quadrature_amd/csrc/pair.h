// pair.h -- kernels of the stacked pair of GPs at many hyper-parameter sets at once
// Part of the libbqhip.so kernel set; compiled into pair.hip (host.h lists the units).
#pragma once
#include "common.h"

// GP2's targets from GP1's posterior at the candidates, for every parameter set b
// (bq.py:933-957): y2[b] = [l_s, exp(mean_c[b])].  flag[b] = 1 where the reference raises
// "GP mean is too large" (mean + 2 sqrt(max(var, 0)) > log of the largest double / 16,
// bq.py:945-947).  grid (ceil(nsc / 256), S).
__global__ __launch_bounds__(256) void pair_targets_kernel(const double *__restrict__ l_s, int ns,
                                                           int nc, const double *__restrict__ mean,
                                                           const double *__restrict__ var,
                                                           long mstride, double max_log,
                                                           double *__restrict__ y2, long ystride,
                                                           int *__restrict__ flag)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ns + nc)
        return;
    double v;
    if (i < ns) {
        v = l_s[i];
    } else {
        const double m = mean[(long)b * mstride + (i - ns)];
        const double s2 = fmax(var[(long)b * mstride + (i - ns)], 0.0);
        if (m + 2.0 * sqrt(s2) > max_log)
            flag[b] = 1; // (every writer stores the same value)
        v = exp(m);
    }
    y2[(long)b * ystride + i] = v;
}

// One record per parameter set for a single read-back of bq_pair_llh:
// out[b * (5 + nc) + {0..4}] = log-ML of GP1, log-ML of GP2, GP1's failure flag, GP2's, the
// overflow flag; then the nc candidate values.  grid (S), block 64.
__global__ void pair_collect_kernel(const double *__restrict__ scal1,
                                    const double *__restrict__ scal2,
                                    const int *__restrict__ info1, const int *__restrict__ info2,
                                    const int *__restrict__ flag, const double *__restrict__ y2,
                                    long ystride, int ns, int nc, double *__restrict__ out)
{
    const int b = blockIdx.x;
    double *o = out + (long)b * (5 + nc);
    if (threadIdx.x == 0) {
        o[0] = scal1[4 * b];
        o[1] = scal2[4 * b];
        o[2] = (double)info1[b];
        o[3] = (double)info2[b];
        o[4] = (double)flag[b];
    }
    for (int i = threadIdx.x; i < nc; i += blockDim.x)
        o[5 + i] = y2[(long)b * ystride + ns + i];
}

// int K_b(x_i, x) N(x | mu, sigma^2) dx = h_b^2 N(x_i | mu, w_b^2 + sigma^2) for the 1-D kernels of
// S parameter sets (gauss_c.pyx:95-164 with d = 1): out[b * n + i].  par[b] = {h^2, 1 / sqrt(C),
// -(log 2 pi + log C) / 2} with C = w^2 + sigma^2, the host's GaussForm<1> of int_K_kernel.
// grid (ceil(n / 256), S).
__global__ __launch_bounds__(256) void pair_int_K_kernel(const double *__restrict__ x, int n,
                                                         const double *__restrict__ par, double mu,
                                                         double *__restrict__ out)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    // the same operations in the same order as int_K_kernel<1> (moments.h)
    const double y = par[3 * b + 1] * (x[i] - mu);
    out[(long)b * n + i] = par[3 * b] * exp(0.0 + (par[3 * b + 2] - 0.5 * (0.0 + y * y)));
}

// The batched expected-squared-mean systems of assemble_esm_kernel (moments.h) for S parameter
// sets x Ma candidates in one batch: element e = e0 + blockIdx.z belongs to parameter set
// s = e / Ma and candidate a = e % Ma; kernel parameters, l_sc and int K come from set s.
__global__ __launch_bounds__(256) void assemble_esm_multi_kernel(
    const double *__restrict__ x_sc, const double *__restrict__ x_a, int Ma, long e0,
    const double *__restrict__ intk_sc, const double *__restrict__ intk_a,
    const double *__restrict__ l_sc, long lstride, const double *__restrict__ jit1,
    const double *__restrict__ jit2, double thresh, const GaussParams *__restrict__ gp,
    double *__restrict__ A, long lda, long astride, EsmLayout L)
{
    const int b = blockIdx.z;
    const long e = e0 + b;
    const int s = (int)(e / Ma), ai = (int)(e % Ma);
    const int t = threadIdx.x;
    const int ib = blockIdx.x * 128, jb = blockIdx.y * 64;
    if (jb > ib + 127)
        return;
    A += (long)b * astride;
    const GaussParams g = gp[s];
    const double xa = x_a[ai];
    const double *ik = intk_sc + (long)s * L.nsc;
    const double *lv = l_sc + (long)s * lstride;
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
                        val += jit2[e];
                    else if (ii >= L.ns && fabs(xi[r] - xa) < thresh)
                        val += jit1[e];
                }
            } else if (ii == L.npad) {
                val = j < L.nsc ? ik[j] : (j == L.nsc ? intk_a[e] : 0.0);
            } else if (ii == L.npad + 1) {
                val = j < L.nsc ? lv[j] : 0.0;
            } else {
                val = (ii == j) ? 1.0 : 0.0;
            }
            v[r] = val;
        }
        double2_t vv = {v[0], v[1]};
        *reinterpret_cast<double2_t *>(A + i + (long)j * lda) = vv;
    }
}

// out[2b] = A_a = (K^-1 intK)[last], out[2b+1] = A_sc . l_sc (see esm_finalize_kernel)
__global__ void esm_multi_finalize_kernel(const double *__restrict__ A, long lda, long astride,
                                          EsmLayout L, int batch, double *__restrict__ out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch)
        return;
    const double *Ab = A + (long)b * astride;
    const double z_last = Ab[L.npad + (long)L.nsc * lda];
    const double l_last = Ab[L.nsc + (long)L.nsc * lda];
    out[2 * b] = z_last / l_last;
    out[2 * b + 1] = -Ab[(L.npad + 1) + (long)L.npad * lda];
}
