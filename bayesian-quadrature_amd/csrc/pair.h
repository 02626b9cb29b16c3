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


// ---------------------------------------------------------------------------
// The acquisition under S parameter sets as S factorisations + border rows (round 4).  The
// S x Ma bordered systems above differ, within a set, only in their LAST rows: the candidate
// x_a, the jitter on the candidate points near it and the two border rows.  With the cut at
// p = 64 floor(ns / 64) <= ns (no jitter above it) the first p columns of all Ma systems of a set
// are eliminated ONCE, with every candidate's row k_a carried along as a border row:
//   index   [0, nsc)          the points x_sc (the first p of them are eliminated)
//           [nsc, nsc + Ma)   the candidates x_a (rows only)
//           nsc + Ma          the row of int K p (over x_sc, then over the candidates)
//           nsc + Ma + 1      the row of l_sc
//           ...               identity padding up to ntot = p + 64 ceil((nsc - p + Ma + 2) / 64)
// (assemble_esmb_kernel; the sweep keeps the whole trailing block up to date).  The trailing block
// is then the Schur complement every candidate's small system starts from: esmb_gather_kernel
// cuts the (nsc - p + 1)-point system of element (set, candidate) out of it -- its points, ITS
// candidate row, the two border rows -- and adds the reference's jitter; a batch of S Ma systems
// of 128 or 192 rows finishes the elimination (the one-launch steps).  1/Ma of the flops.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assemble_esmb_kernel(
    const double *__restrict__ x_sc, const double *__restrict__ x_a, int Ma, long s0,
    const double *__restrict__ intk_sc, const double *__restrict__ intk_a,
    const double *__restrict__ l_sc, long lstride, const GaussParams *__restrict__ gp,
    double *__restrict__ A, long lda, long astride, int nsc, int ntot)
{
    const int b = blockIdx.z;
    const long s = s0 + b;
    const int t = threadIdx.x;
    const int ib = blockIdx.x * 128, jb = blockIdx.y * 64;
    if (jb > ib + 127)
        return;
    A += (long)b * astride;
    const GaussParams g = gp[s];
    const double *ik = intk_sc + s * nsc;
    const double *ika = intk_a + s * Ma;
    const double *lv = l_sc + s * lstride;
    const int npts = nsc + Ma, rb = npts, rl = npts + 1;
    const int i = ib + (t & 63) * 2;
    const int jbase = jb + (t >> 6) * 16;
    if (i >= ntot)
        return;
    double xi[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ii = i + r;
        xi[r] = ii < nsc ? x_sc[ii] : (ii < npts ? x_a[ii - nsc] : 0.0);
    }
    for (int jj = 0; jj < 16; ++jj) {
        const int j = jbase + jj;
        if (j >= ntot)
            break;
        const double xj = j < nsc ? x_sc[j] : (j < npts ? x_a[j - nsc] : 0.0);
        double v[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ii = i + r;
            double val;
            if (ii < npts && j < npts) {
                const double tdiff = xi[r] - xj;
                val = g.c * exp_gauss((tdiff * tdiff) * g.nh[0]);
            } else if (ii == rb) {
                val = j < nsc ? ik[j] : (j < npts ? ika[j - nsc] : 0.0);
            } else if (ii == rl) {
                val = j < nsc ? lv[j] : 0.0;
            } else {
                val = (ii == j) ? 1.0 : 0.0;
            }
            v[r] = val;
        }
        double2_t vv = {v[0], v[1]};
        *reinterpret_cast<double2_t *>(A + i + (long)j * lda) = vv;
    }
}

// element e = e0 + blockIdx.z = (set s - s0, candidate a): its small system from the trailing
// block T = A[p.., p..] of set s.  Small layout (EsmLayout Ls): points [0, nt0) = x_sc[p ..),
// nt0 = the candidate; rows Ls.npad / Ls.npad + 1 = the int K p / l_sc rows; identity padding.
// grid (ceil(ntot_s / 64), ntot_s / 64, elements), block 256: thread t -> row (t & 63), 16 columns.
__global__ __launch_bounds__(256) void esmb_gather_kernel(
    const double *__restrict__ A, long lda, long astride, int p, int nsc, int Ma, int ns,
    const double *__restrict__ x_sc, const double *__restrict__ x_a,
    const double *__restrict__ jit1, const double *__restrict__ jit2, double thresh, long s0,
    long e0, double *__restrict__ As, long ldas, long asstride, EsmLayout Ls)
{
    const long e = e0 + blockIdx.z;
    const int sl = (int)(e / Ma - s0), a = (int)(e % Ma);
    const double *T = A + (long)sl * astride;
    double *O = As + (long)blockIdx.z * asstride;
    const int nt0 = Ls.nsc; // trailing points of the big system = points of the small one - 1
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j0 = blockIdx.y * 64 + (threadIdx.x >> 6) * 16;
    if (i >= Ls.ntot)
        return;
    // the big system's index of small row i (-1: padding)
    auto big = [&](int q) {
        if (q < nt0)
            return p + q;
        if (q == nt0)
            return nsc + a;
        if (q == Ls.npad)
            return nsc + Ma;
        if (q == Ls.npad + 1)
            return nsc + Ma + 1;
        return -1;
    };
    const int bi = big(i);
    const double xa = x_a[a];
    for (int jj = 0; jj < 16; ++jj) {
        const int j = j0 + jj;
        if (j >= Ls.ntot)
            break;
        const int bj = big(j);
        double val;
        if (j > i) {
            val = 0.0; // (the upper triangle is never read)
        } else if (bi >= 0 && bj >= 0) {
            val = T[bi + (long)bj * lda]; // (big() is increasing: bi >= bj, the lower triangle)
            if (i == j && i == nt0)
                val += jit2[e];
            else if (i == j && i < nt0 && p + i >= ns && fabs(x_sc[p + i] - xa) < thresh)
                val += jit1[e];
        } else {
            val = (i == j) ? 1.0 : 0.0;
        }
        O[i + (long)j * ldas] = val;
    }
}
