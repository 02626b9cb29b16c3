// common.h -- types and device helpers shared by every kernel of libbqhip.so.
//
// GaussParams, the fp64 lane broadcast, the scheduling pin, the Gaussian-kernel exp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "types.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// Pins a value's computation at this point of the program: without it LLVM
// sinks the rank-1 updates of the right-looking factorisations down to their
// first use (a left-looking schedule), keeps every broadcast multiplier alive
// and spills thousands of registers.
#define PIN(v) asm volatile("" : "+v"(v))

__device__ __forceinline__ double readlane_f64(double v, int srclane)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, srclane);
    hi = __builtin_amdgcn_readlane(hi, srclane);
    return __hiloint2double(hi, lo);
}

// exp(x) for the Gaussian-kernel exponent (x <= 0): range reduction by ln 2 with a
// hi/lo split, degree-13 Taylor polynomial on |r| <= ln2/2 (truncation 6e-18),
// v_ldexp_f64 for the scale.  20 fp64 instructions against ~28 of the library
// routine, <= 1 ulp; x is clamped at -800 where the result is 0 anyway.
__device__ __forceinline__ double exp_gauss(double x)
{
    x = __builtin_fmax(x, -800.0);
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;            // 1/13!
    p = __builtin_fma(p, r, 2.0876756987868100e-09); // 1/12!
    p = __builtin_fma(p, r, 2.5052108385441720e-08); // 1/11!
    p = __builtin_fma(p, r, 2.7557319223985888e-07); // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06); // 1/9!
    p = __builtin_fma(p, r, 2.4801587301587302e-05); // 1/8!
    p = __builtin_fma(p, r, 1.9841269841269841e-04); // 1/7!
    p = __builtin_fma(p, r, 1.3888888888888889e-03); // 1/6!
    p = __builtin_fma(p, r, 8.3333333333333332e-03); // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02); // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01); // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)k);
}

// quad rotation of a 16-lane row (the four-block MFMA kernels, gemm.h)
template <int S>
__device__ __forceinline__ double row_ror_quads(double v)
{
    if (S == 0)
        return v;
    constexpr int ctrl = 0x120 + 4 * S; // row_ror:4S: lane l reads lane (l - 4S) mod 16 of its row
    int lo = __double2loint(v), hi = __double2hiint(v);
    // every lane is written (full row and bank masks), so no "old" value is needed:
    // mov_dpp avoids the zero-initialising v_mov that update_dpp(0, ...) costs
    lo = __builtin_amdgcn_mov_dpp(lo, ctrl, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

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

template <int D>
__device__ __forceinline__ double gauss_q(const double *p, const double *q, const GaussParams &g)
{
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const double t = p[k] - q[k];
        acc += (t * t) * g.nh[k];
    }
    return acc;
}
