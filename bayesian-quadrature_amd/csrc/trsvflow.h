// trsvflow.h -- the single-vector sweeps over a resident factor as ONE launch each
// Part of the libbqhip.so kernel set; compiled into k_reduce.hip (host.h lists the units).
#pragma once
#include "trsv.h"

// ---------------------------------------------------------------------------
// trsv.h runs a sweep as one launch per B columns; nothing in a launch waits on anything in it,
// and every step pays a dependent kernel boundary: N = 4096 is sixteen launches of 5-9 us for
// 21 us worth of bytes.  Here the workgroups of ALL steps are one grid and the boundaries become
// hand-offs through memory:
//   * a workgroup takes its logical id from a ticket counter as it starts, so ids follow the
//     actual start order whatever the dispatcher does; ids are laid out step after step -- the
//     diagonal pieces of a step first, then its update blocks by increasing row -- and a
//     workgroup only ever waits for work of EARLIER steps: lower tickets, already resident or
//     done.  No residency assumption, no deadlock, however few workgroups the chip admits;
//   * the payload is its own flag.  A first version signalled through counters (stores, every
//     wave's vmcnt drain, a workgroup barrier, an atomic add, the consumer's poll): 4-5 us per
//     hop, what a kernel boundary costs -- N = 4096 took 0.140 ms either way.  Now every value is
//     written ONCE into a slot that was filled with a sentinel (all bits set: a NaN no arithmetic
//     here produces -- results that are NaN are stored as the canonical one) before the launch,
//     and the consumer polls the very words it needs: y has one slot per entry, x one per entry
//     and VERSION (x after s updates is row s of an ns x npad array; version 0 is the caller's
//     vector).  Eight-byte agent-scope relaxed atomic loads / stores (sc1: L2-coherent, past the
//     CU's L1), no drain, no barrier, no counter on the producer's side;
//   * the factor's columns -- the only traffic that counts -- are requested BEFORE the wait.
// The arithmetic is trsv.h's, operation for operation: the same results bit for bit.
// Spins are bounded: after ~1 s without progress a wave raises the abort word (mapped host
// memory), every spinner sees it and leaves; the host reports BQ_ERR_HIP (flow_check).
// ---------------------------------------------------------------------------
#define BQ_FLOW_SPINS (1 << 21)
#define BQ_FLOW_SENT 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ unsigned long long flow_ldu(const double *p)
{
    return __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flow_st(double *p, double v)
{
    if (v != v) // (never the sentinel: a NaN result goes out as the canonical quiet NaN)
        v = __longlong_as_double(0x7FF8000000000000ll);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p),
                       (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// the pause between two polls of a wave, and the time-out: returns false to stop spinning
__device__ __forceinline__ bool flow_again(int &it, int *abort_w)
{
    if (++it >= BQ_FLOW_SPINS) {
        __hip_atomic_store(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
    }
    if ((it & 63) == 0 &&
        __hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)
        return false;
    __builtin_amdgcn_s_sleep(8);
    return true;
}

// Hand-off words are read with raw buffer loads (buffer_load_dwordx2 ... sc1): hipcc follows every
// relaxed atomic load with s_waitcnt vmcnt(0) -- sixteen dependent round trips per poll, which made
// the polled sweep ten times slower than the launches it replaces -- and schedules these like any
// load: sixteen in flight, one wait.  The descriptor carries the vector's range: a word beyond it
// reads as zero (not the sentinel, and the value the arithmetic wants there), so no lane needs a
// mask and no vector needs slack behind it.
typedef unsigned int flow_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t flow_rsrc(const double *p, int n)
{
    // raw buffer over n doubles from p (stride 0, range checked against num_records)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p), 0, n > 0 ? n * 8 : 0, 0x00020000);
}
__device__ __forceinline__ unsigned long long flow_bld(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    const flow_u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 16 /* sc1 */);
    return ((unsigned long long)v.y << 32) | v.x;
}

// The gate in front of a workgroup's polls: ONE lane of the workgroup watches one word of each
// vector it is going to read (nullptr: none) until neither is the sentinel, the others wait at
// the barrier.  Without it every wave polls all its words all the time -- 4096 resident waves
// times 8-16 loads per microsecond on a few dozen cache lines -- and the producers' stores queue
// behind the polls (N = 4096, both sweeps: 74 us with the gates, 97 with them on the update
// blocks only, 126 without).  The words behind the gate are written at about the same time as
// the watched one; the full polls that follow catch the stragglers.
__device__ __forceinline__ void flow_gate(const double *w0, const double *w1, int *abort_w)
{
    if (threadIdx.x == 0) {
        // (both words in one round trip: buffer loads; a missing word reads as zero)
        const __amdgpu_buffer_rsrc_t r0 = flow_rsrc(w0, w0 ? 1 : 0), r1 = flow_rsrc(w1, w1 ? 1 : 0);
        int it = 0;
        for (;;) {
            const unsigned long long a = flow_bld(r0, 0), b = flow_bld(r1, 0);
            if ((a != BQ_FLOW_SENT && b != BQ_FLOW_SENT) || !flow_again(it, abort_w))
                break;
        }
    }
    __syncthreads();
}

// v[q] = p[lo + lane + 64 q] for lo + lane + 64 q < hi (else 0) and the same for (p2, lo2, hi2),
// polled -- one round trip per poll for both -- until none of the wave's values is the sentinel.
__device__ __forceinline__ void flow_poll8x2(const double *p, int lo, int hi, double (&v)[8],
                                             const double *p2, int lo2, int hi2, double (&v2)[8],
                                             int lane, int *abort_w)
{
    const __amdgpu_buffer_rsrc_t r1 = flow_rsrc(p, hi), r2 = flow_rsrc(p2, hi2);
    const int o1 = (lo + lane) * 8, o2 = (lo2 + lane) * 8;
    int it = 0;
    for (;;) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v[q] = __longlong_as_double((long long)flow_bld(r1, o1 + 512 * q));
            v2[q] = __longlong_as_double((long long)flow_bld(r2, o2 + 512 * q));
        }
        // (the sentinel has every bit set: AND the words, compare once)
        unsigned long long all1 = 0ull, any = 0ull;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned long long a = (unsigned long long)__double_as_longlong(v[q]),
                                     b = (unsigned long long)__double_as_longlong(v2[q]);
            any |= (a == BQ_FLOW_SENT) | (b == BQ_FLOW_SENT);
        }
        (void)all1;
        if (__all(any == 0ull) || !flow_again(it, abort_w))
            return;
    }
}

__device__ __forceinline__ double flow_poll1(const double *p, bool mine, int *abort_w)
{
    int it = 0;
    for (;;) {
        const unsigned long long u = mine ? flow_ldu(p) : 0ull;
        if (__all(u != BQ_FLOW_SENT) || !flow_again(it, abort_w))
            return __longlong_as_double((long long)u);
    }
}

// Forward sweep L y = x, all steps in one grid.  Logical block (s, lb): J = s B;
// lb < bJ / 16: y[J + 16 lb + wave]; else rows J + bJ + 64 (lb - bJ / 16) .. + 63 take y of block
// s - 1.  x0: the right-hand side (version 0, not modified); xv: versions 1 .. (row s - 1 of an
// ns x npad array); y and xv filled with the sentinel.  nr / tt: the whole arrays (block s at
// + J B).  grid: the total over the steps, block 1024.
template <int NB> // B / 64 (4 or 8), or 0: any B <= 512
__global__ __launch_bounds__(1024) void trsv_fwd_flow_kernel(const double *__restrict__ L, long ldl,
                                                             int npad, int B,
                                                             const double *__restrict__ nr,
                                                             const double *__restrict__ tt,
                                                             const double *x0, double *xv, double *y,
                                                             int *ticket, int *abort_w, int fault)
{
    __shared__ double ys[512];
    __shared__ double part[16][64];
    __shared__ int sh[1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ns = (npad + B - 1) / B;
    if (t == 0)
        sh[0] = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    __syncthreads();
    int id = sh[0], s = 0, J = 0, bJ = 0, ndiag = 0;
    for (; s < ns; ++s) {
        J = s * B;
        bJ = min(B, npad - J);
        ndiag = bJ >> 4;
        const int cnt = ndiag + (s > 0 ? (npad - J - bJ) / 64 : 0);
        if (id < cnt)
            break;
        id -= cnt;
    }
    if (s >= ns)
        return;
    // x after v updates: version v
    auto xver = [&](int v) { return v == 0 ? x0 : xv + (long)(v - 1) * npad; };
    if (id < ndiag) {
        // ---- diagonal piece (a wave per entry, no workgroup barrier): the matrix columns
        // first, then x of block s (version s - 1: every update but y_{s-1}'s, which T carries)
        // and y of block s - 1 as they appear
        const int k = id * 16 + wave;
        const double *c1 = nr + (long)J * B + (long)k * B, *c2 = tt + (long)J * B + (long)k * B;
        const int hi1 = (k | 63) + 1, hi2 = J > 0 ? B : 0;
        double m1[8], m2[8], xb[8], yb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = lane + 64 * q;
            m1[q] = i < hi1 ? c1[i] : 0.0;
            m2[q] = i < hi2 ? c2[i] : 0.0;
        }
        // (the same piece of the block before, and the first of my rows' x)
        flow_gate(s > 0 ? y + J - B + id * 16 : nullptr, s > 1 ? xver(s - 1) + J + id * 16 : nullptr,
                  abort_w);
        flow_poll8x2(xver(s > 0 ? s - 1 : 0) + J, 0, hi1, xb, y + J - B, 0, hi2, yb, lane, abort_w);
        double s0 = 0.0, s1 = 0.0, u0 = 0.0, u1 = 0.0;
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            s0 = fma(m1[q], xb[q], s0);
            s1 = fma(m1[q + 1], xb[q + 1], s1);
        }
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            u0 = fma(m2[q], yb[q], u0);
            u1 = fma(m2[q + 1], yb[q + 1], u1);
        }
        const double sum = wave_sum((s0 + s1) + (u0 + u1));
        // (fault: a test's lost hand-off -- step 1 never publishes, its consumers time out)
        if (lane == 0 && !(fault && s == 1))
            flow_st(y + J + k, -sum);
        return;
    }
    // ---- update block: rows r take y of block s - 1 (columns J - B .. J - 1): version s - 1 -> s
    const int rb = (J + bJ) / 64 + (id - ndiag);
    const long r = 64L * rb + lane;
    const double *p = L + r + (long)(J - B + wave) * ldl;
    double v[NB > 0 ? NB : 1][4];
    if (NB > 0) {
#pragma unroll
        for (int it = 0; it < NB; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[it][u] = p[(long)(64 * it + 16 * u) * ldl];
    }
    flow_gate(y + J - B + (rb * 16) % B, s > 1 ? xver(s - 1) + r - lane : nullptr, abort_w);
    double xin = 0.0;
    {
        // every wave polls its 64 entries of y (B <= 512: waves 0 .. B / 64 - 1), wave 0 its
        // rows of x as well, in the same round trip
        int it = 0;
        const __amdgpu_buffer_rsrc_t ry = flow_rsrc(y + J - B, B), rx = flow_rsrc(xver(s - 1) + 64L * rb, 64);
        const int ox = wave == 0 ? lane * 8 : (1 << 20); // (other waves: out of range, zero)
        for (;;) {
            const unsigned long long uy = flow_bld(ry, t * 8), ux = flow_bld(rx, ox);
            if (__all(uy != BQ_FLOW_SENT && ux != BQ_FLOW_SENT) || !flow_again(it, abort_w)) {
                if (t < B)
                    ys[t] = __longlong_as_double((long long)uy);
                xin = __longlong_as_double((long long)ux);
                break;
            }
        }
    }
    __syncthreads();
    double sacc[4] = {0.0, 0.0, 0.0, 0.0};
    if (NB > 0) {
#pragma unroll
        for (int it = 0; it < NB; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                sacc[u] = fma(v[it][u], ys[wave + 64 * it + 16 * u], sacc[u]);
    } else {
        for (int k = wave; k < B; k += 64) {
            double vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                vv[u] = p[(long)(16 * u) * ldl];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                sacc[u] = fma(vv[u], ys[k + 16 * u], sacc[u]);
            p += 64 * ldl;
        }
    }
    part[wave][lane] = (sacc[0] + sacc[1]) + (sacc[2] + sacc[3]);
    __syncthreads();
    if (wave == 0) {
        double a = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w)
            a += part[w][lane];
        flow_st(xv + (long)(s - 1) * npad + r, xin - a);
    }
}

// Backward sweep L^T y = x, all steps in one grid.  Step t: J = last - t B (last = the last
// block's first column); lb < bJ / 16: y[J + 16 lb + wave]; else columns 64 (lb - bJ / 16) .. + 63
// (< J) take y of block J + B (step t - 1): version t - 1 -> t.  nt / uu: the whole arrays.
__global__ __launch_bounds__(1024) void trsv_bwd_flow_kernel(const double *__restrict__ L, long ldl,
                                                             int npad, int B,
                                                             const double *__restrict__ nt,
                                                             const double *__restrict__ uu,
                                                             const double *x0, double *xv, double *y,
                                                             int *ticket, int *abort_w, int fault)
{
    __shared__ int sh[1];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ns = (npad + B - 1) / B, last = (npad - 1) / B * B;
    if (t == 0)
        sh[0] = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    __syncthreads();
    int id = sh[0], st = 0, J = 0, bJ = 0, bn = 0, ndiag = 0;
    for (; st < ns; ++st) {
        J = last - st * B;
        bJ = min(B, npad - J);
        bn = st > 0 ? min(B, npad - J - B) : 0;
        ndiag = bJ >> 4;
        const int cnt = ndiag + (bn > 0 ? J / 64 : 0);
        if (id < cnt)
            break;
        id -= cnt;
    }
    if (st >= ns)
        return;
    auto xver = [&](int v) { return v == 0 ? x0 : xv + (long)(v - 1) * npad; };
    if (id < ndiag) {
        const int k = id * 16 + wave;
        const double *c1 = nt + (long)J * B + (long)k * B, *c2 = uu + (long)J * B + (long)k * B;
        const int lo1 = k & ~63;
        double m1[8], m2[8], xb[8], yb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i1 = lo1 + lane + 64 * q, i2 = lane + 64 * q;
            m1[q] = i1 < bJ ? c1[i1] : 0.0;
            m2[q] = i2 < bn ? c2[i2] : 0.0;
        }
        flow_gate(st > 0 ? y + J + B + (id * 16) % bn : nullptr,
                  st > 1 ? xver(st - 1) + J + id * 16 : nullptr, abort_w);
        flow_poll8x2(xver(st > 0 ? st - 1 : 0) + J, lo1, bJ, xb, y + J + B, 0, bn, yb, lane, abort_w);
        double s0 = 0.0, s1 = 0.0, u0 = 0.0, u1 = 0.0;
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            s0 = fma(m1[q], xb[q], s0);
            s1 = fma(m1[q + 1], xb[q + 1], s1);
        }
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            u0 = fma(m2[q], yb[q], u0);
            u1 = fma(m2[q + 1], yb[q + 1], u1);
        }
        const double sum = wave_sum((s0 + s1) + (u0 + u1));
        if (lane == 0)
            flow_st(y + J + k, -sum);
        return;
    }
    // ---- update block: 64 columns i < J take y of block J + B (a wave per four columns)
    const int cb = id - ndiag;
    const int nk = bn >> 6;
    const int i0 = cb * 64 + wave * 4;
    const double *p = L + J + B + lane + (long)i0 * ldl;
    double v[4][8], yr[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
            v[cc][q] = q < nk ? p[64 * q + (long)cc * ldl] : 0.0;
    flow_gate(y + J + B + (cb * 16) % bn, st > 1 ? xver(st - 1) + cb * 64 : nullptr, abort_w);
    double xin;
    {
        // y of block J + B (eight per lane) and my four columns' x in one round trip
        double xq[8];
        flow_poll8x2(y + J + B, 0, bn, yr, xver(st - 1) + i0, 0, 4, xq, lane, abort_w);
        xin = xq[0];
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
        const double sm = wave_sum(s0 + s1);
        if (lane == cc)
            mine = sm;
    }
    if (lane < 4)
        flow_st(xv + (long)(st - 1) * npad + i0 + lane, xin - mine);
}

// The host side of a one-vector solve through kernels instead of the copy engine and memsets
// (tools/stream_ops_bench.hip: a pinned copy costs a stream 8-9 us, a memset 2.7, a further kernel
// 2.9): flow_in_kernel brings the right-hand side in from the caller's mapped pinned vector
// (coalesced, zero padded to npad) and sets every hand-off slot of BOTH sweeps to the sentinel in
// the same launch; flow_out_kernel takes the solution out.  grid: enough 256-thread blocks for
// max(npad, nfill).
__global__ __launch_bounds__(256) void flow_in_kernel(const double *__restrict__ hsrc, int n,
                                                      double *__restrict__ x, int npad,
                                                      unsigned long long *__restrict__ fill,
                                                      long nfill)
{
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < npad)
        x[i] = i < n ? hsrc[i] : 0.0;
    for (long k = i; k < nfill; k += (long)gridDim.x * 256)
        fill[k] = BQ_FLOW_SENT;
}

__global__ __launch_bounds__(256) void flow_out_kernel(const double *__restrict__ x, int n,
                                                       double *__restrict__ hdst)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        hdst[i] = x[i];
}

// Small host <-> device transfers of a latency-bound call as ONE kernel on the mapped pinned staging
// buffer instead of copy-engine operations (+2.9 us on the stream instead of +8..9 each): up to two
// (dst, src, words) pairs of 8-byte words.  grid (1), block 256.
__global__ __launch_bounds__(256) void copy_words2_kernel(unsigned long long *__restrict__ d1,
                                                          const unsigned long long *__restrict__ s1,
                                                          int n1,
                                                          unsigned long long *__restrict__ d2,
                                                          const unsigned long long *__restrict__ s2,
                                                          int n2)
{
    for (int i = threadIdx.x; i < n1; i += 256)
        d1[i] = s1[i];
    for (int i = threadIdx.x; i < n2; i += 256)
        d2[i] = s2[i];
}

// dst[j] = src[j * stride], j < n: a row of a column-major matrix as a vector (the fit's z out of its
// factor) by a kernel rather than a strided copy-engine operation
__global__ __launch_bounds__(256) void gather_row_kernel(double *__restrict__ dst,
                                                         const double *__restrict__ src, long stride,
                                                         int n)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n)
        dst[j] = src[(long)j * stride];
}

// A small plan's inputs out of ONE mapped pinned staging buffer into their places (bq_plan_set_inputs
// below 256 KB: four copy operations from pageable memory and a synchronisation otherwise).  stage:
// [parameters gw nprob | x (d n) nprob | xo (d M) nprob | y n nprob] in 8-byte words; pts: a d x ntot
// block per problem, x at its columns [0, n), xo at [npad, npad + M); yd: npad per problem.
__global__ __launch_bounds__(256) void plan_scatter_kernel(const double *__restrict__ stage,
                                                           int nprob, int d, int n, int M, int ntot,
                                                           int npad, int gw,
                                                           double *__restrict__ gp,
                                                           double *__restrict__ pts,
                                                           double *__restrict__ yd)
{
    const long ng = (long)gw * nprob, nx = (long)d * n * nprob, nxo = (long)d * M * nprob,
               ny = (long)n * nprob;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < ng + nx + nxo + ny;
         i += (long)gridDim.x * 256) {
        const double v = stage[i];
        if (i < ng) {
            gp[i] = v;
        } else if (i < ng + nx) {
            const long k = i - ng, b = k / ((long)d * n), r = k % ((long)d * n);
            pts[b * (long)d * ntot + r] = v;
        } else if (i < ng + nx + nxo) {
            const long k = i - ng - nx, b = k / ((long)d * M), r = k % ((long)d * M);
            pts[b * (long)d * ntot + (long)d * npad + r] = v;
        } else {
            const long k = i - ng - nx - nxo, b = k / n, r = k % n;
            yd[b * (long)npad + r] = v;
        }
    }
}

// ... and its results into one: [scal 4 nb | info nb (ints) | mean M nb | var M nb]
__global__ __launch_bounds__(256) void plan_gather_kernel(double *__restrict__ out,
                                                          const double *__restrict__ scal,
                                                          const int *__restrict__ info,
                                                          const double *__restrict__ mean,
                                                          const double *__restrict__ var, int nb,
                                                          int M)
{
    const long o_info = 4L * nb, o_mean = o_info + nb, o_var = o_mean + (long)M * nb;
    int *oi = reinterpret_cast<int *>(out + o_info);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < 4L * nb; i += (long)gridDim.x * 256)
        out[i] = scal[i];
    for (long i = blockIdx.x * 256L + threadIdx.x; i < nb; i += (long)gridDim.x * 256)
        oi[i] = info[i];
    if (mean)
        for (long i = blockIdx.x * 256L + threadIdx.x; i < (long)M * nb; i += (long)gridDim.x * 256)
            out[o_mean + i] = mean[i];
    if (var)
        for (long i = blockIdx.x * 256L + threadIdx.x; i < (long)M * nb; i += (long)gridDim.x * 256)
            out[o_var + i] = var[i];
}

// A small host matrix (n x n, ld n, in the mapped pinned staging buffer) into its padded device
// matrix (ntot x ntot, identity in the padding) and the failure flag cleared, one launch; and the
// result back behind the flag: out[0] = *info, out[1 + i + j n] = A[i + j lda].  The linalg_c
// drop-ins at the reference's own sizes are a handful of launches: every copy-engine operation and
// every extra synchronisation costs them more than a kernel does (tools/stream_ops_bench.hip).
__global__ __launch_bounds__(256) void mat_in_kernel(const double *__restrict__ stage, int n,
                                                     double *__restrict__ A, long lda, int ntot,
                                                     int *__restrict__ info)
{
    if (info && blockIdx.x == 0 && threadIdx.x == 0)
        *info = 0;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < (long)ntot * ntot;
         e += (long)gridDim.x * 256) {
        const int i = (int)(e % ntot), j = (int)(e / ntot);
        A[i + (long)j * lda] = (i < n && j < n) ? stage[i + (long)j * n] : (i == j ? 1.0 : 0.0);
    }
}

__global__ __launch_bounds__(256) void mat_out_kernel(double *__restrict__ out,
                                                      const double *__restrict__ A, long lda, int n,
                                                      const int *__restrict__ info)
{
    if (blockIdx.x == 0 && threadIdx.x == 0)
        out[0] = (double)*info;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < (long)n * n; e += (long)gridDim.x * 256) {
        const int i = (int)(e % n), j = (int)(e / n);
        out[1 + e] = A[i + (long)j * lda];
    }
}

// (L L^T) X = B for a factor of at most 64 rows and at most 64 right-hand sides -- the linalg_c
// drop-ins at the reference's own sizes -- as ONE launch on the mapped staging buffer:
// stage = [X (n x nrhs, out) | L (n x n, ld n, lower) | B (n x nrhs)].  One workgroup: the factor goes
// to LDS (row stride 65), a wave takes a column at a time, lane = row, and runs the forward and
// the backward substitution with the column in one register (2 n steps of a v_readlane, an LDS read
// and an FMA).  block 1024.
__global__ __launch_bounds__(1024) void small_potrs_kernel(double *__restrict__ stage, int n,
                                                           int nrhs)
{
    __shared__ double Ls[64 * 65];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    double *X = stage;
    const double *L = stage + (long)n * nrhs, *B = L + (long)n * n;
    for (int e = t; e < n * n; e += 1024) {
        const int i = e % n, j = e / n;
        Ls[i * 65 + j] = L[e];
    }
    __syncthreads();
    const double rc = lane < n ? 1.0 / Ls[lane * 65 + lane] : 0.0;
    for (int r = wave; r < nrhs; r += 16) {
        double y = lane < n ? B[lane + (long)r * n] : 0.0;
        for (int k = 0; k < n; ++k) {
            const double yk = readlane_f64(y, k) * readlane_f64(rc, k);
            const double lik = lane < n ? Ls[lane * 65 + k] : 0.0;
            y = lane > k ? __builtin_fma(-lik, yk, y) : (lane == k ? yk : y);
        }
        for (int k = n - 1; k >= 0; --k) {
            const double xk = readlane_f64(y, k) * readlane_f64(rc, k);
            const double lki = lane < n ? Ls[k * 65 + lane] : 0.0;
            y = lane < k ? __builtin_fma(-lki, xk, y) : (lane == k ? xk : y);
        }
        if (lane < n)
            X[lane + (long)r * n] = y;
    }
}
