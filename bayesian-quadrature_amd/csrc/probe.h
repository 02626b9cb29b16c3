// probe.h -- hardware probes (MFMA / FMA rate, HBM bandwidth, MFMA layout, rsq accuracy)
// Part of the libbqhip.so kernel set; compiled into probe.hip (host.h lists the units).
#pragma once
#include "common.h"
#include "potf2.h"

// ---------------------------------------------------------------------------
// hardware probes
// ---------------------------------------------------------------------------
// relative error of the raw v_rsq_f64 seed, of one and of two Newton steps, against
// the correctly rounded 1/sqrt; out[3*i + k]
__global__ void probe_rsq_kernel(const double *x, double *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double d = x[i];
    const double ref = 1.0 / sqrt(d);
    double y = __builtin_amdgcn_rsq(d);
    out[3 * i] = fabs(y - ref) / ref;
    const double hd = 0.5 * d;
    double t = __builtin_fma(-hd * y, y, 0.5);
    y = __builtin_fma(y, t, y);
    out[3 * i + 1] = fabs(y - ref) / ref;
    t = __builtin_fma(-hd * y, y, 0.5);
    y = __builtin_fma(y, t, y);
    out[3 * i + 2] = fabs(y - ref) / ref;
}

// exp_gauss (common.h) on an array of arguments, for the per-element accuracy test
__global__ void probe_exp_kernel(const double *x, double *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        out[i] = exp_gauss(x[i]);
}

// MFMA issue study: NACC independent accumulators per wave, 16x16x4 (KIND 0) or the
// four-block 4x4x4 form (KIND 1); waves per SIMD are set by the grid.
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void probe_mfma_var_kernel(double *out, int iters)
{
    const double a = 1.0 + threadIdx.x * 1e-9, bb = 1.0 - threadIdx.x * 1e-9;
    if (KIND == 0) {
        double4_t c[NACC];
#pragma unroll
        for (int k = 0; k < NACC; ++k)
            c[k] = (double4_t){0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < NACC; ++k)
                c[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, c[k], 0, 0, 0);
        }
        double4_t s = c[0];
#pragma unroll
        for (int k = 1; k < NACC; ++k)
            s += c[k];
        if (s[0] == 123.456)
            out[0] = s[1];
    } else {
        double c[NACC];
#pragma unroll
        for (int k = 0; k < NACC; ++k)
            c[k] = 0.0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < NACC; ++k)
                c[k] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, c[k], 0, 0, 0);
        }
        double s = c[0];
#pragma unroll
        for (int k = 1; k < NACC; ++k)
            s += c[k];
        if (s == 123.456)
            out[0] = s;
    }
}

// The GEMM inner step without memory: 4 P-fragments x 4 Q-fragments, 64 four-block MFMAs
// per step, with (ROT = 1) or without (ROT = 0) the three DPP quad rotations per Q fragment.
// RND: operands with random mantissas (a hash of the lane), as real data has -- the power an
// MFMA draws, and with it the clock the chip sustains, depends on the bits that toggle.
template <int ROT, int RND = 0>
__global__ __launch_bounds__(256, 2) void probe_mfma_step_kernel(double *out, int iters)
{
    double acc[4][4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[a][b][s] = 0.0;
    double pf[4], qf[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        pf[a] = 1.0 + threadIdx.x * 1e-9 + a;
        qf[a] = 1.0 - threadIdx.x * 1e-9 - a;
        if (RND) {
            unsigned long long h = (threadIdx.x * 8 + a + 1) * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
            h *= 0xBF58476D1CE4E5B9ull;
            h ^= h >> 32;
            pf[a] = (double)(long long)h * (1.0 / 9223372036854775808.0);
            h *= 0x94D049BB133111EBull;
            h ^= h >> 31;
            qf[a] = (double)(long long)h * (1.0 / 9223372036854775808.0);
        }
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
            qf[tn] += RND ? 0.000123456789 * pf[tn] : 1e-12; // keeps the rotations inside the loop
            const double q0 = qf[tn];
            const double q1 = ROT ? row_ror_quads<1>(q0) : q0;
            const double q2 = ROT ? row_ror_quads<2>(q0) : q0;
            const double q3 = ROT ? row_ror_quads<3>(q0) : q0;
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                acc[tm][tn][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(q0, pf[tm], acc[tm][tn][0], 0, 0, 0);
                acc[tm][tn][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(q1, pf[tm], acc[tm][tn][1], 0, 0, 0);
                acc[tm][tn][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(q2, pf[tm], acc[tm][tn][2], 0, 0, 0);
                acc[tm][tn][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(q3, pf[tm], acc[tm][tn][3], 0, 0, 0);
            }
        }
    }
    double sum = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                sum += acc[a][b][s];
    if (sum == 123.456)
        out[0] = sum;
}

// Operand map of v_mfma_f64_4x4x4_4b_f64: block (la, lb) of the grid sets A = 1 in lane la
// only and B = 1 in lane lb only; out[la*64 + lb] = the lane whose D becomes 1 (or -1).
template <int CBSZ, int ABID>
__global__ void probe_layout444_kernel(int *out)
{
    const int la = blockIdx.x, lb = blockIdx.y, l = threadIdx.x;
    const double a = (l == la) ? 1.0 : 0.0, bb = (l == lb) ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bb, 0.0, CBSZ, ABID, 0);
    // several D lanes may light up when the A block is broadcast: record a bit mask
    const unsigned long long m = __ballot(d != 0.0);
    if (l == 0) {
        out[2 * (la * 64 + lb)] = (int)(m & 0xffffffffu);
        out[2 * (la * 64 + lb) + 1] = (int)(m >> 32);
    }
}

// pseudo-random values in (-1, 1): operands of all zeros draw far less MFMA power than real
// data, and a product timed on them runs at clocks the real sweep never sees
__global__ __launch_bounds__(256) void probe_fill_kernel(double *dst, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = (i + seed) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29;
        h *= 0xBF58476D1CE4E5B9ull;
        h ^= h >> 32;
        dst[i] = (double)(long long)h * (1.0 / 9223372036854775808.0);
    }
}

__global__ void probe_empty_kernel(double *out)
{
    if (out == nullptr && threadIdx.x == 9999)
        out[0] = 0.0;
}

__global__ __launch_bounds__(256) void probe_mfma_kernel(double *out, int iters)
{
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = 1.0 + threadIdx.x * 1e-9, bb = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, c3, 0, 0, 0);
    }
    c0 += c1 + c2 + c3;
    if (c0[0] == 123.456)
        out[0] = c0[1];
}

__global__ __launch_bounds__(256) void probe_fma_kernel(double *out, int iters)
{
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
           a6 = a0 + 6, a7 = a0 + 7;
    const double m = 0.999999, c = 1e-9;
    for (int i = 0; i < iters; ++i) {
        a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c;
        a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
    }
    const double s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (s == 123.456)
        out[0] = s;
}

__global__ __launch_bounds__(256) void probe_write_kernel(double2_t *dst, size_t n2)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const double2_t v = {1.0, 2.0};
    for (; i < n2; i += stride)
        dst[i] = v;
}

__global__ __launch_bounds__(256) void probe_copy_kernel(double2_t *dst, const double2_t *src,
                                                         size_t n2)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n2; i += stride)
        dst[i] = src[i];
}

// Known-bytes calibration of the FETCH_SIZE counter for the access pattern of the single-vector
// sweeps (trsv.h): every lane loads 8 bytes, a wave 512 contiguous bytes, each byte of `src`
// exactly once per launch -- `n` doubles, nothing else is read.
__global__ __launch_bounds__(1024) void probe_read8_kernel(const double *__restrict__ src,
                                                           size_t n, double *out)
{
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 1024;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (; i + 3 * stride < n; i += 4 * stride) {
        a0 += src[i];
        a1 += src[i + stride];
        a2 += src[i + 2 * stride];
        a3 += src[i + 3 * stride];
    }
    for (; i < n; i += stride)
        a0 += src[i];
    const double s = (a0 + a1) + (a2 + a3);
    if (s == 123.456)
        out[0] = s;
}

// D = A B with A[i][k] = i (row tag), B[k][j] = [k==0] * 1 ... we want each D
// element to carry row*16+col: use A[i][k] = (k==0) ? i*16 : (k==1 ? 1 : 0),
// B[k][j] = (k==0) ? 1 : (k==1 ? j : 0)  ->  D[i][j] = 16 i + j.
__global__ void probe_layout_kernel(double *out)
{
    const int l = threadIdx.x;
    const int i = l & 15, k = l >> 4;
    const double a = (k == 0) ? 16.0 * i : (k == 1 ? 1.0 : 0.0);
    const double bb = (k == 0) ? 1.0 : (k == 1 ? (double)(l & 15) : 0.0);
    double4_t c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r)
        out[l * 4 + r] = c[r];
}

// timing probe: the factor alone on a block that is restored from Ain every launch;
// stamps[0..4] of the last launch = s_memtime at entry / loaded / chain done / blocks in LDS / end
template <int NW>
__global__ __launch_bounds__(64 * NW) void potf2_probe_kernel(const double *__restrict__ Ain,
                                                          double *__restrict__ A, long lda,
                                                          double *__restrict__ dinv,
                                                          int *__restrict__ info,
                                                          long long *stamps, int from_lds)
{
    __shared__ __attribute__((aligned(16))) double lds[BQ_POTF2_LDS_DOUBLES];
    __builtin_amdgcn_s_setprio(3);
    if (from_lds) {
        // as the slab step hands the block over: through LDS
        double *Ts = lds; // where the panel slots will be
        for (int e = threadIdx.x; e < 4096; e += 64 * NW)
            Ts[e] = Ain[(e & 63) + (long)(e >> 6) * lda];
        __syncthreads();
        potf2_body<NW>(A, lda, 0, dinv, info, lds, Ts, 64, stamps);
    } else {
        for (int e = threadIdx.x; e < 4096; e += 64 * NW)
            A[(e & 63) + (long)(e >> 6) * lda] = Ain[(e & 63) + (long)(e >> 6) * lda];
        __syncthreads();
        potf2_body<NW>(A, lda, 0, dinv, info, lds, nullptr, 0, stamps);
    }
}

// ---------------------------------------------------------------------------
// What a hand-off between two workgroups costs (could the C2 pass be ONE
// launch resident on ONE XCD, its 16 steps handing over through that XCD's L2 instead of through
// launch boundaries?).  Sixteen one-wave workgroups, eight ping-pong pairs; the producer writes
// n16 x 1 KiB of payload and a flag, the consumer polls the flag, reads the payload and answers.
//   mode 0: partners b and b + 8 (observed: the same XCD); plain payload stores, s_waitcnt,
//           flag stored sc1; flag polled and payload read with sc1 loads (L1 bypassed, served by
//           the XCD's L2) -- no buffer_wbl2, no buffer_inv
//   mode 1: partners b and b + 1 (different XCDs); agent-scope release before the flag, agent-
//           scope acquire after the poll -- the form every cross-XCD hand-off in this library uses
//   mode 2: as 1 between partners b and b + 8 (what the fences cost inside one XCD)
// Bounded spins: a partner that never answers raises *err and everyone leaves.
// out[b] = wall-clock ticks (100 MHz) of block b's loop, xcc[b] = its XCC id, bad[b] = payload
// words that did not carry the expected value.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void probe_hop_kernel(unsigned *flags, double *payload, int iters,
                                                       int mode, int n16, long long *out, int *xcc,
                                                       int *bad, int *err)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int pair = mode == 1 ? (b >> 1) : (b & 7), side = mode == 1 ? (b & 1) : (b >> 3);
    unsigned *mine = flags + (2 * pair + side) * 64, *peer = flags + (2 * pair + 1 - side) * 64;
    double *pmine = payload + (size_t)(2 * pair + side) * 128 * n16;
    const double *ppeer = payload + (size_t)(2 * pair + 1 - side) * 128 * n16;
    if (lane == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[b] = (int)(id & 0xf);
    }
    int nbad = 0;
    bool dead = false;
    const long long t0 = wall_clock64();
    for (int i = 1; i <= iters && !dead; ++i) {
        for (int ph = 0; ph < 2; ++ph) {
            if ((ph == 0) == (side == 0)) {
                // produce
                for (int q = 0; q < n16; ++q) {
                    double *d = pmine + q * 128 + 2 * lane;
                    d[0] = (double)i;
                    d[1] = (double)(i + q);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (mode != 0)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0)
                    __hip_atomic_store(mine, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // consume
                int spins = 0;
                unsigned v;
                do {
                    v = __hip_atomic_load(peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v = __builtin_amdgcn_readfirstlane(v);
                    if (++spins > (1 << 20) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        dead = true;
                        break;
                    }
                } while (v < (unsigned)i);
                if (dead)
                    break;
                if (mode != 0)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                for (int q = 0; q < n16; ++q) {
                    const double *s = ppeer + q * 128 + 2 * lane;
                    double a, c;
                    if (mode == 0) {
                        a = __hip_atomic_load(s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        c = __hip_atomic_load(s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        a = s[0];
                        c = s[1];
                    }
                    nbad += (a != (double)i) + (c != (double)(i + q));
                }
            }
        }
    }
    const long long t1 = wall_clock64();
    if (dead && lane == 0)
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int off = 32; off; off >>= 1)
        nbad += __shfl_down(nbad, off);
    if (lane == 0) {
        out[b] = t1 - t0;
        bad[b] = nbad;
    }
}
