// trsv.h -- sweeps over a resident factor for ONE right-hand side (cho_solve_vec, alpha, the
// quadratic form of Z_var): GEMV kernels that stream the factor once at HBM rate.
// Part of the libbqhip.so kernel set; compiled into k_reduce.hip (host.h lists the units).
//
// The row form of the MFMA sweeps pads one right-hand side to 64 rows and runs two GEMMs per
// B columns whose long-k product has too few tiles to fill the chip (0.95 ms at N = 4096 for
// 134 MB of factor).  Here a sweep is ONE launch per B columns.  Forward (L y = x), block J,
// with W_J = L_JJ^-1 explicit (the wide inverses of the row sweeps):
//     y_J = W_J (x_J - L_J,J-1 y_J-1) = W_J x_J - T_J y_J-1,      T_J = W_J L_J,J-1
// where x_J holds the updates of all blocks before J-1 only -- the coupling to the block just
// solved is folded into the B x B matrix T_J, built once per factor.  So the launch that
// computes y_J (a few workgroups: dots with the columns of -W_J^T and T_J^T, one wave per
// entry) ALSO applies y_J-1 to every row below block J (the other workgroups: axpy form,
// lane = row), and nothing in it waits for anything else in it.  Backward (L^T y = x):
//     y_J = W_J^T x_J - U_J^T y_J+1,   U_J = L_J+1,J W_J,   x[:J] -= L[J+1, :J]^T y_J+1
// (dot form, wave = column).  Every load is a wave reading 512 contiguous bytes.
#pragma once
#include "common.h"

// sum over the 64 lanes, every lane gets it: four DPP steps inside the rows of 16 (the
// shuffle form goes through the LDS crossbar, ~12 ds_bpermute per value), then the four row
// sums from lanes 0 / 16 / 32 / 48.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double s)
{
    s += dpp_f64<0xB1>(s);  // quad_perm [1, 0, 3, 2]
    s += dpp_f64<0x4E>(s);  // quad_perm [2, 3, 0, 1]
    s += dpp_f64<0x141>(s); // row_half_mirror
    s += dpp_f64<0x140>(s); // row_mirror
    return (readlane_f64(s, 0) + readlane_f64(s, 16)) + (readlane_f64(s, 32) + readlane_f64(s, 48));
}

// sum_q m[lane + 64 q + k ld] v[lane + 64 q] over lo <= i < hi (i = lane + 64 q < 512), all
// loads in flight
__device__ __forceinline__ double trsv_dot8(const double *__restrict__ col,
                                            const double *__restrict__ v, int lane, int lo,
                                            int hi)
{
    double m[8], xv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int i = lo + lane + 64 * q;
        m[q] = i < hi ? col[i] : 0.0;
        xv[q] = i < hi ? v[i] : 0.0;
    }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        s0 = fma(m[q], xv[q], s0);
        s1 = fma(m[q + 1], xv[q + 1], s1);
    }
    return s0 + s1;
}

// One forward step (block J of width bJ; B = the full block width = ld of the wide arrays,
// <= 512).  Workgroups 0 .. bJ/16-1: y[J + k] for 16 values of k each (a wave per k).  The
// others (J > 0 only): x[r] -= sum_k L[r, J-B+k] y[J-B+k] for 64 rows r >= J + bJ each; 16
// waves take the columns k = w, w+16, ..., partial sums meet in LDS.
// nr / tt: this block's -W^T and T^T (element (i, k) at [i + k B]).
template <int NB> // B / 64 (4 or 8), or 0: any B <= 512
__global__ __launch_bounds__(1024) void trsv_fwd_step_kernel(const double *__restrict__ L,
                                                             long ldl, int J, int bJ, int B,
                                                             const double *__restrict__ nr,
                                                             const double *__restrict__ tt,
                                                             double *__restrict__ x,
                                                             double *__restrict__ y)
{
    __shared__ double ys[512];
    __shared__ double part[16][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ndiag = bJ >> 4;
    if ((int)blockIdx.x < ndiag) {
        // both dots' loads in flight together (J = 0: the second one is masked off)
        const int k = blockIdx.x * 16 + wave;
        double s = trsv_dot8(nr + (long)k * B, x + J, lane, 0, (k | 63) + 1) +
                   trsv_dot8(tt + (long)k * B, y + J - B, lane, 0, J > 0 ? B : 0);
        s = wave_sum(s);
        if (lane == 0)
            y[J + k] = -s;
        return;
    }
    const long r = (long)J + bJ + (long)(blockIdx.x - ndiag) * 64 + lane;
    const double *p = L + r + (long)(J - B + wave) * ldl;
    const double x0 = wave == 0 ? x[r] : 0.0;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (NB > 0) {
        // every load of this wave's B / 16 columns in flight before the first use
        double v[NB > 0 ? NB : 1][4];
#pragma unroll
        for (int it = 0; it < NB; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[it][u] = p[(long)(64 * it + 16 * u) * ldl];
        if (t < B)
            ys[t] = y[J - B + t];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NB; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                s[u] = fma(v[it][u], ys[wave + 64 * it + 16 * u], s[u]);
    } else {
        if (t < B)
            ys[t] = y[J - B + t];
        __syncthreads();
        for (int k = wave; k < B; k += 64) {
            // B % 64 == 0: four columns per pass, every wave the same count
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = p[(long)(16 * u) * ldl];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                s[u] = fma(v[u], ys[k + 16 * u], s[u]);
            p += 64 * ldl;
        }
    }
    part[wave][lane] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (wave == 0) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w)
            a += part[w][lane];
        x[r] = x0 - a;
    }
}

// One backward step (block J, full width B unless it is the last; bn = width of block J + B,
// 0 for the last block).  Workgroups 0 .. bJ/16-1: y[J + k].  The others (bn > 0): x[i] -=
// sum_k L[J+B+k, i] y[J+B+k] for 64 columns i < J each (a wave per four columns, a lane
// holds its bn / 64 entries of y).  nt / uu: this block's -W and U (element (i, k) at [i + k B]).
__global__ __launch_bounds__(1024) void trsv_bwd_step_kernel(const double *__restrict__ L,
                                                             long ldl, int J, int bJ, int B,
                                                             int bn,
                                                             const double *__restrict__ nt,
                                                             const double *__restrict__ uu,
                                                             double *__restrict__ x,
                                                             double *__restrict__ y)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ndiag = bJ >> 4;
    if ((int)blockIdx.x < ndiag) {
        const int k = blockIdx.x * 16 + wave;
        double s = trsv_dot8(nt + (long)k * B, x + J, lane, k & ~63, bJ) +
                   trsv_dot8(uu + (long)k * B, y + J + B, lane, 0, bn);
        s = wave_sum(s);
        if (lane == 0)
            y[J + k] = -s;
        return;
    }
    const int nk = bn >> 6;
    const int i0 = (blockIdx.x - ndiag) * 64 + wave * 4;
    const double *p = L + J + B + lane + (long)i0 * ldl;
    const double *yn = y + J + B;
    const double x0 = lane < 4 ? x[i0 + lane] : 0.0;
    double v[4][8], yr[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        yr[q] = q < nk ? yn[lane + 64 * q] : 0.0;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
            v[cc][q] = q < nk ? p[64 * q + (long)cc * ldl] : 0.0;
    }
    double mine = 0.0;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            s0 = fma(v[cc][q], yr[q], s0);
            s1 = fma(v[cc][q + 1], yr[q + 1], s1);
        }
        const double s = wave_sum(s0 + s1);
        if (lane == cc)
            mine = s;
    }
    if (lane < 4)
        x[i0 + lane] = x0 - mine;
}

// dst block = transpose of the src block (ld ldm both), 64 x 64 tiles through LDS; grid
// (source rows / 64, source columns / 64, blocks), block stride bs.
__global__ __launch_bounds__(256) void transpose_blocks_kernel(const double *__restrict__ src,
                                                               double *__restrict__ dst, int ldm,
                                                               long bs)
{
    __shared__ double tile[64][65];
    const double *S = src + (long)blockIdx.z * bs;
    double *D = dst + (long)blockIdx.z * bs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bi = blockIdx.x * 64, bj = blockIdx.y * 64;
    for (int c = wave; c < 64; c += 4)
        tile[c][lane] = S[bi + lane + (long)(bj + c) * ldm];
    __syncthreads();
    for (int c = wave; c < 64; c += 4)
        D[bj + lane + (long)(bi + c) * ldm] = tile[lane][c];
}

// out[0] = -sum_i v[i]^2 (one workgroup)
__global__ __launch_bounds__(256) void neg_sumsq_kernel(const double *__restrict__ v, int n,
                                                        double *__restrict__ out)
{
    __shared__ double part[4];
    const int t = threadIdx.x;
    double s = 0.0;
    for (int i = t; i < n; i += 256)
        s = fma(v[i], v[i], s);
    s = wave_sum(s);
    if ((t & 63) == 0)
        part[t >> 6] = s;
    __syncthreads();
    if (t == 0)
        out[0] = -((part[0] + part[1]) + (part[2] + part[3]));
}
