// potf2.h -- 64x64 diagonal Cholesky blocks (one-wave and four-wave variants)
// Part of the libbqhip.so kernel set; included through kernels.h.
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------
// refined reciprocal square root: v_rsq_f64 seed + two Newton steps, and the
// square root s = d r with one correction.  Relative error ~1 ulp; the pivot
// chain is the critical path of the whole factorisation, so it avoids the
// long div/sqrt library sequences.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void rsqrt_sqrt_f64(double d, double &r, double &s)
{
    double y = __builtin_amdgcn_rsq(d);
    const double hd = 0.5 * d;
    double t = __builtin_fma(-hd * y, y, 0.5);
    y = __builtin_fma(y, t, y);
    t = __builtin_fma(-hd * y, y, 0.5);
    y = __builtin_fma(y, t, y);
    double q = d * y;
    const double e = __builtin_fma(-q, q, d);
    q = __builtin_fma(0.5 * y, e, q);
    r = y;
    s = q;
}

// ---------------------------------------------------------------------------
// 64x64 diagonal block: unblocked right-looking Cholesky by ONE wave.  Lane i
// holds row i in 64 fp64 registers.  Per column: the pivot and the next
// column's multiplier travel by v_readlane (short dependency chain), the other
// multipliers l_k are broadcast through LDS (every lane reads the same
// address), two per ds_read_b128.  Writes the lower triangle back and 1/L_jj
// to dinv[64].  info[b] receives the 1-based global column of the first
// non-positive pivot (first failure wins; 0 = ok).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void potf2_64_kernel(double *__restrict__ A, long lda,
                                                      long astride, int j0,
                                                      double *__restrict__ dinv, long dstride,
                                                      int *__restrict__ info)
{
    __shared__ __attribute__((aligned(16))) double col[2][64];
    // the panel is the critical path; under look-ahead it shares SIMDs with the
    // trailing update's MFMA waves and should win instruction issue
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.z;
    double *Ab = A + (long)b * astride + j0 + (long)j0 * lda;
    const int lane = threadIdx.x;
    double a[64];
    {
        const double *pr = Ab + lane;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            a[j] = *pr;
            pr += lda;
        }
    }
    int bad = 0;
    double d = readlane_f64(a[0], 0);
    double r, s;
    double myr = 0.0; // lane j keeps 1 / L_jj
    rsqrt_sqrt_f64(d, r, s);
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        if (!(d > 0.0) && bad == 0)
            bad = j0 + j + 1;
        const double l = (lane == j) ? s : a[j] * r;
        a[j] = l;
        myr = (lane == j) ? r : myr;
        if (j < 63) {
            double2_t lk[32];
            if (j < 62) {
                // broadcast reads of the column are issued first; the next
                // pivot's readlane + rsqrt chain below runs under their latency
                col[j & 1][lane] = l;
                __syncthreads();
                const double2_t *c2 = reinterpret_cast<const double2_t *>(col[j & 1]);
#pragma unroll
                for (int kk = (j + 2) >> 1; kk < 32; ++kk)
                    lk[kk] = c2[kk];
            }
            a[j + 1] -= l * readlane_f64(l, j + 1);
            d = readlane_f64(a[j + 1], j + 1);
            rsqrt_sqrt_f64(d, r, s);
            if (j < 62) {
#pragma unroll
                for (int kk = (j + 2) >> 1; kk < 32; ++kk) {
                    if (2 * kk >= j + 2)
                        a[2 * kk] -= l * lk[kk][0];
                    a[2 * kk + 1] -= l * lk[kk][1];
                }
#pragma unroll
                for (int k = j + 2; k < 64; ++k)
                    PIN(a[k]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    dinv[(long)b * dstride + lane] = myr;
    // fresh per-lane pointer: without the opaque copy the compiler keeps the 64
    // load addresses alive across the whole factorisation and spills
    double *pw = Ab + lane;
    asm volatile("" : "+v"(pw));
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        if (lane >= j)
            *pw = a[j];
        pw += lda;
    }
    if (lane == 0 && bad != 0 && info[b] == 0)
        info[b] = bad;
}

// ---------------------------------------------------------------------------
// 64x64 diagonal block by FOUR waves (one per SIMD).  Every wave holds all 64
// rows (lane = row) and a quarter of the columns: wave w owns the columns
// 16q + 4w + s (q, s = 0..3), i.e. the matrix is cut into sixteen 4-column
// panels dealt round-robin to the waves.  Panel p is factored by its owner
// (pivot chain as in potf2_64_kernel, the in-panel updates by v_readlane) and
// published to a ring of three LDS slots; after ONE workgroup barrier per
// panel every wave applies the rank-4 update to its own later columns.  The
// owner of panel p+1 updates only that panel before starting its pivot chain
// and catches up on its remaining columns one barrier later (the slot of panel
// p stays valid that long), so the chain of rsqrt's -- the critical path of
// the whole factorisation -- waits for 16 FMAs per panel instead of 64.
// ---------------------------------------------------------------------------
struct Potf2W {
    double a[4][4]; // a[q][s] = column 16q + 4w + s of row `lane`
    double myr;     // 1 / L_cc for the lane that is the pivot row of an owned column
    int bad;
};

// rank-4 update of this wave's columns in group Q by the panel in `slot`
template <int Q>
__device__ __forceinline__ void potf2w_update_group(Potf2W &st, const double *slot,
                                                    const double (&li)[4], int w)
{
    double2_t lk[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double2_t *src = reinterpret_cast<const double2_t *>(slot + s * 64 + 16 * Q + 4 * w);
        lk[s][0] = src[0];
        lk[s][1] = src[1];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        st.a[Q][0] -= li[s] * lk[s][0][0];
        st.a[Q][1] -= li[s] * lk[s][0][1];
        st.a[Q][2] -= li[s] * lk[s][1][0];
        st.a[Q][3] -= li[s] * lk[s][1][1];
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
        PIN(st.a[Q][cc]);
}

// factor panel P (columns 4P .. 4P+3, group QP = P>>2) held by this wave; publish
template <int P>
__device__ __forceinline__ void potf2w_factor(Potf2W &st, double *slot, int lane, int j0)
{
    constexpr int QP = P >> 2;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        constexpr int dummy = 0;
        (void)dummy;
        const int c = 4 * P + s;
        const double d = readlane_f64(st.a[QP][s], c);
        if (!(d > 0.0) && st.bad == 0)
            st.bad = j0 + c + 1;
        double r, sq;
        rsqrt_sqrt_f64(d, r, sq);
        // the multipliers need only r; the refined square root (three more dependent
        // operations) goes to the diagonal entry alone, off the pivot chain -- row c of the
        // later columns is above the diagonal and never read
        const double l = st.a[QP][s] * r;
        st.myr = (lane == c) ? r : st.myr;
#pragma unroll
        for (int s2 = s + 1; s2 < 4; ++s2)
            st.a[QP][s2] -= l * readlane_f64(l, 4 * P + s2);
        const double lf = (lane == c) ? sq : l;
        st.a[QP][s] = lf;
        slot[s * 64 + lane] = lf;
    }
}

template <int P>
struct Potf2WSteps {
    static __device__ __forceinline__ void run(Potf2W &st, double *ring, int w, int lane, int j0)
    {
        constexpr int QP = P >> 2, WP = P & 3;
        constexpr int PN = P + 1, QN = PN >> 2, WN = PN & 3;
        __syncthreads(); // panel P is published
        const double *slot = ring + (P % 3) * 256;
        if (P >= 1 && w == WP) {
            // I factored panel P before touching my later groups with panel P-1
            const double *prev = ring + ((P + 2) % 3) * 256;
            double lp[4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                lp[s] = prev[s * 64 + lane];
            if (QP < 1) potf2w_update_group<1>(st, prev, lp, w);
            if (QP < 2) potf2w_update_group<2>(st, prev, lp, w);
            if (QP < 3) potf2w_update_group<3>(st, prev, lp, w);
        }
        double li[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
            li[s] = slot[s * 64 + lane];
        if (P < 15 && w == WN) {
            // next owner: bring its panel up to date, then run the pivot chain
            if (QN == 0) potf2w_update_group<0>(st, slot, li, w);
            if (QN == 1) potf2w_update_group<1>(st, slot, li, w);
            if (QN == 2) potf2w_update_group<2>(st, slot, li, w);
            if (QN == 3) potf2w_update_group<3>(st, slot, li, w);
            potf2w_factor<(P < 15 ? PN : 15)>(st, ring + (PN % 3) * 256, lane, j0);
        } else {
            // my columns of the panel's own group lie after it only if w > WP
            if (w > WP) {
                if (QP == 0) potf2w_update_group<0>(st, slot, li, w);
                if (QP == 1) potf2w_update_group<1>(st, slot, li, w);
                if (QP == 2) potf2w_update_group<2>(st, slot, li, w);
                if (QP == 3) potf2w_update_group<3>(st, slot, li, w);
            }
            if (QP < 1) potf2w_update_group<1>(st, slot, li, w);
            if (QP < 2) potf2w_update_group<2>(st, slot, li, w);
            if (QP < 3) potf2w_update_group<3>(st, slot, li, w);
        }
        Potf2WSteps<P + 1>::run(st, ring, w, lane, j0);
    }
};
template <>
struct Potf2WSteps<16> {
    static __device__ __forceinline__ void run(Potf2W &, double *, int, int, int) {}
};

// The factorisation proper, callable by any 256-thread workgroup: Ab points at
// the 64x64 block (leading dimension lda), j0 is its global column (for the
// failure report), dinv_b / info_b belong to this batch element.
// src / lsrc: where the block is read from when it is not in place (the slab step hands it
// over in LDS); the factor always goes to Ab.
__device__ __forceinline__ void potf2_64x4_body(double *__restrict__ Ab, long lda, int j0,
                                                double *__restrict__ dinv_b,
                                                int *__restrict__ info_b, double *ring, int *sbad,
                                                const double *src = nullptr, long lsrc = 0)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    Potf2W st;
    st.myr = 0.0;
    st.bad = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            st.a[q][s] = src ? src[lane + (long)(16 * q + 4 * w + s) * lsrc]
                             : Ab[lane + (long)(16 * q + 4 * w + s) * lda];
    if (w == 0)
        potf2w_factor<0>(st, ring, lane, j0);
    Potf2WSteps<0>::run(st, ring, w, lane, j0);
    // W_b = L_bb^-1 of the four 16 x 16 diagonal sub-blocks, for the MFMA panel solve
    // (trsm_blk_kernel).  Every wave drops its columns of the four blocks (and its
    // reciprocal pivots) into LDS straight from registers -- the ring is free now --, then
    // wave w inverts block w: lane j < 16 runs the forward substitution of unit column j,
    // every L entry a broadcast LDS read, four partial sums per row (about a microsecond,
    // instead of 64 dependent column steps in every workgroup of the panel solve).  All of
    // this comes BEFORE the write-back of the factor: the barrier would otherwise wait for
    // those 64 stores per lane to be acknowledged (measured: 3 us per launch).
    double *rd = ring + 1024; // 64 reciprocal pivots
    {
        // blk[b][i + 16 k] = L[16 b + i][16 b + k]; my columns: k = 4 w + s of every block
        const int bq = lane >> 4, i16 = lane & 15;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int sc = 0; sc < 4; ++sc)
                if (bq == q)
                    ring[256 * q + i16 + 16 * (4 * w + sc)] = st.a[q][sc];
        if (((lane >> 2) & 3) == w)
            rd[lane] = st.myr;
        if (lane == 0)
            sbad[w] = st.bad;
    }
    __syncthreads();
    // write back the lower triangle of my columns, and my reciprocal pivots
    {
        double *pw = Ab + lane + (long)(4 * w) * lda;
        asm volatile("" : "+v"(pw));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (lane >= 16 * q + 4 * w + s)
                    pw[(long)s * lda] = st.a[q][s];
            pw += 16 * lda;
        }
    }
    if (((lane >> 2) & 3) == w)
        dinv_b[lane] = st.myr;
    // (the stores above drain while the inverses are computed)
    if (threadIdx.x == 0) {
        int first = 0;
        for (int k = 0; k < 4; ++k)
            if (sbad[k] != 0 && (first == 0 || sbad[k] < first))
                first = sbad[k];
        if (first != 0 && info_b[0] == 0)
            info_b[0] = first;
    }
    if (lane < 16) {
        const double *blk = ring + 256 * w;
        double wc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double sacc[4] = {(i == lane) ? 1.0 : 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < i; ++k)
                sacc[k & 3] -= blk[i + 16 * k] * wc[k];
            // rows above the column's own are exactly zero (all partial sums are)
            wc[i] = ((sacc[0] + sacc[1]) + (sacc[2] + sacc[3])) * rd[16 * w + i];
        }
        double *Wb = dinv_b + 64 + 256 * w + 16 * lane;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            Wb[i] = wc[i];
    }
}

// ===========================================================================
// Second generation of the four-wave diagonal factor (potf2f): the same dealing of
// sixteen 4-column panels to four waves, with everything that is not the pivot
// chain taken off the owner wave:
//   * the reciprocal square root is ONE third-order (Halley) step from the
//     v_rsq_f64 seed -- y1 = y0 (1 + e/2 + 3 e^2/8), e = 1 - d y0^2: five
//     instructions, dependency depth four, error 5 e^3/16 ~ 1e-18 for a 2^-20
//     seed -- instead of two Newton steps (six instructions, depth six);
//   * the pivot row is scaled like every other row (L_cc = d r, about an ulp from
//     sqrt d): no per-pivot selects; the reciprocal pivots are one division per
//     lane at the very end, and a non-positive pivot is found afterwards as the
//     first NaN on the diagonal (d <= 0 makes r NaN or infinite and everything
//     after it NaN), so the chain carries no failure bookkeeping either;
//   * SYNC = 1: every panel has its own LDS slot and a published-column counter,
//     so no wave ever waits at a workgroup barrier: the owner of the next panel
//     applies the columns of the current one AS THEY APPEAR (one rank-1 update of
//     its four columns per published column), and starts its own pivot chain one
//     LDS round trip after the last of them.  LDS operations of one wave execute
//     in order, so a column store followed by the counter store needs no wait.
// ===========================================================================
__device__ __forceinline__ double rsqrt_halley_f64(double d)
{
    const double y0 = __builtin_amdgcn_rsq(d);
    const double t = d * y0;
    const double e = __builtin_fma(-t, y0, 1.0);
    const double p = __builtin_fma(0.375, e, 0.5);
    const double q = y0 * e;
    return __builtin_fma(q, p, y0);
}

struct Potf2F {
    double a[4][4]; // a[q][s] = column 16q + 4w + s of row `lane`
};

// LDS traffic of the column hand-over, as instructions: a volatile or atomic store would make
// the compiler wait for it (s_waitcnt) on the pivot chain, and a generic pointer would turn
// into a flat store.  The low half of a generic pointer into LDS is the LDS address.
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(unsigned long long)p;
}
template <int OFF>
__device__ __forceinline__ void lds_store_f64(unsigned addr, double v)
{
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_store_i32(unsigned addr, int v)
{
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ int lds_load_i32(unsigned addr)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// LDS of the factor: 16 panel slots of 4 x 64 doubles, then the four 16 x 16 diagonal
// sub-blocks (1024 doubles), then the published-column counter
#define BQ_POTF2F_SLOTS (16 * 256)
// (counter in the first word after the blocks; the next 64 words absorb the counter stores of
// the lanes that are not lane 0, so that the store needs no branch)
#define BQ_POTF2F_LDS_DOUBLES (BQ_POTF2F_SLOTS + 1024 + 40)

// slots_a: LDS address of this lane's entry of column 0 of slot 0; SLOT: the panel's slot
template <int P, int SLOT>
__device__ __forceinline__ void potf2f_factor(Potf2F &st, double *slot, unsigned slots_a,
                                              unsigned cnt_store, int lane, int sync)
{
    constexpr int QP = P >> 2;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 4 * P + s;
        const double d = readlane_f64(st.a[QP][s], c);
        const double r = rsqrt_halley_f64(d);
        const double l = st.a[QP][s] * r;
#pragma unroll
        for (int s2 = s + 1; s2 < 4; ++s2)
            st.a[QP][s2] = __builtin_fma(-l, readlane_f64(l, 4 * P + s2), st.a[QP][s2]);
        st.a[QP][s] = l;
        if (sync) {
            if (s == 0) lds_store_f64<2048 * SLOT>(slots_a, l);
            if (s == 1) lds_store_f64<2048 * SLOT + 512>(slots_a, l);
            if (s == 2) lds_store_f64<2048 * SLOT + 1024>(slots_a, l);
            if (s == 3) lds_store_f64<2048 * SLOT + 1536>(slots_a, l);
            lds_store_i32(cnt_store, c + 1); // in order behind the column store (same wave)
        } else {
            slot[s * 64 + lane] = l;
        }
    }
}

// rank-1 update of this wave's columns of group Q by published column `col` (li = its entry
// in this lane's row)
template <int Q>
__device__ __forceinline__ void potf2f_update1(Potf2F &st, const double *col, double li, int w)
{
    const double2_t *src = reinterpret_cast<const double2_t *>(col + 16 * Q + 4 * w);
    const double2_t k0 = src[0], k1 = src[1];
    st.a[Q][0] = __builtin_fma(-li, k0[0], st.a[Q][0]);
    st.a[Q][1] = __builtin_fma(-li, k0[1], st.a[Q][1]);
    st.a[Q][2] = __builtin_fma(-li, k1[0], st.a[Q][2]);
    st.a[Q][3] = __builtin_fma(-li, k1[1], st.a[Q][3]);
}

template <int Q>
__device__ __forceinline__ void potf2f_update_group(Potf2F &st, const double *slot,
                                                    const double (&li)[4], int w)
{
    double2_t lk[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double2_t *src = reinterpret_cast<const double2_t *>(slot + s * 64 + 16 * Q + 4 * w);
        lk[s][0] = src[0];
        lk[s][1] = src[1];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        st.a[Q][0] = __builtin_fma(-li[s], lk[s][0][0], st.a[Q][0]);
        st.a[Q][1] = __builtin_fma(-li[s], lk[s][0][1], st.a[Q][1]);
        st.a[Q][2] = __builtin_fma(-li[s], lk[s][1][0], st.a[Q][2]);
        st.a[Q][3] = __builtin_fma(-li[s], lk[s][1][1], st.a[Q][3]);
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
        PIN(st.a[Q][cc]);
}

// (the asm's memory clobber keeps the data reads that follow behind the poll).  The spin is
// bounded: a counter that never arrives -- a bug, not a data condition -- must not hang the
// GPU; the word behind the counter records it and the factor reports info = j0 + 65.
__device__ __forceinline__ void potf2f_wait(unsigned cnt_a, int need)
{
    for (int spin = 0; lds_load_i32(cnt_a) < need; ++spin) {
        if (spin > (1 << 22)) {
            lds_store_i32(cnt_a + 4u, 1);
            break;
        }
        __builtin_amdgcn_s_sleep(0);
    }
}

template <int SYNC, int P>
struct Potf2FSteps {
    static __device__ __forceinline__ void run(Potf2F &st, double *slots, unsigned cnt,
                                               unsigned cnt_store, int w, int lane)
    {
        constexpr int QP = P >> 2, WP = P & 3;
        constexpr int PN = P + 1, QN = PN >> 2, WN = PN & 3;
        // SYNC 0: ring of three slots + one barrier per panel; SYNC 1: a slot per panel
        const double *slot = slots + (SYNC ? P : P % 3) * 256;
        double *nslot = slots + (SYNC ? PN : PN % 3) * 256;
        if (!SYNC)
            __syncthreads(); // panel P is published
        if (P < 15 && w == WN) {
            // next owner.  First what it still owes panel P-1 ... nothing: its own later
            // groups are caught up after its chain (see below); now its panel meets panel P.
            double li[4];
            if (SYNC) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    potf2f_wait(cnt, 4 * P + s + 1);
                    li[s] = slot[s * 64 + lane];
                    if (QN == 0) potf2f_update1<0>(st, slot + s * 64, li[s], w);
                    if (QN == 1) potf2f_update1<1>(st, slot + s * 64, li[s], w);
                    if (QN == 2) potf2f_update1<2>(st, slot + s * 64, li[s], w);
                    if (QN == 3) potf2f_update1<3>(st, slot + s * 64, li[s], w);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    li[s] = slot[s * 64 + lane];
                if (QN == 0) potf2f_update_group<0>(st, slot, li, w);
                if (QN == 1) potf2f_update_group<1>(st, slot, li, w);
                if (QN == 2) potf2f_update_group<2>(st, slot, li, w);
                if (QN == 3) potf2f_update_group<3>(st, slot, li, w);
            }
            potf2f_factor<(P < 15 ? PN : 15), (SYNC ? (P < 15 ? PN : 15) : (P < 15 ? PN : 15) % 3)>(
                st, nslot, lds_addr(slots) + 8u * (unsigned)lane, cnt_store, lane, SYNC);
            // catch up: my later groups with panel P (panel PN is mine and needs no update
            // of my own columns of its group beyond the in-panel ones -- but the columns of
            // group QN that come AFTER panel PN do not exist in this wave: a wave owns one
            // panel per group)
            if (QN < 1) potf2f_update_group<1>(st, slot, li, w);
            if (QN < 2) potf2f_update_group<2>(st, slot, li, w);
            if (QN < 3) potf2f_update_group<3>(st, slot, li, w);
        } else {
            if (SYNC)
                potf2f_wait(cnt, 4 * P + 4);
            if (w != WP || P == 15) {
                double li[4];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    li[s] = slot[s * 64 + lane];
                // my panel of the panel's own group lies after it only if w > WP
                if (w > WP) {
                    if (QP == 0) potf2f_update_group<0>(st, slot, li, w);
                    if (QP == 1) potf2f_update_group<1>(st, slot, li, w);
                    if (QP == 2) potf2f_update_group<2>(st, slot, li, w);
                    if (QP == 3) potf2f_update_group<3>(st, slot, li, w);
                }
                if (QP < 1) potf2f_update_group<1>(st, slot, li, w);
                if (QP < 2) potf2f_update_group<2>(st, slot, li, w);
                if (QP < 3) potf2f_update_group<3>(st, slot, li, w);
            } else {
                // the owner of panel P: its later groups with its own panel (from registers'
                // copy in LDS -- the slot it has just written)
                double li[4];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    li[s] = st.a[QP][s];
                if (QP < 1) potf2f_update_group<1>(st, slot, li, w);
                if (QP < 2) potf2f_update_group<2>(st, slot, li, w);
                if (QP < 3) potf2f_update_group<3>(st, slot, li, w);
            }
        }
        Potf2FSteps<SYNC, P + 1>::run(st, slots, cnt, cnt_store, w, lane);
    }
};
template <int SYNC>
struct Potf2FSteps<SYNC, 16> {
    static __device__ __forceinline__ void run(Potf2F &, double *, unsigned, unsigned, int, int) {}
};

// lds: BQ_POTF2F_LDS_DOUBLES doubles.  src / lsrc as for potf2_64x4_body; when src lies in
// the slots' LDS (the slab step's Ts), pass src_in_slots so that nobody publishes before
// every wave has its columns.
template <int SYNC>
__device__ __forceinline__ void potf2f_body(double *__restrict__ Ab, long lda, int j0,
                                            double *__restrict__ dinv_b,
                                            int *__restrict__ info_b, double *lds,
                                            const double *src = nullptr, long lsrc = 0,
                                            bool src_in_slots = false,
                                            long long *stamps = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#define BQ_STAMP(k)                                                                                \
    if (stamps && threadIdx.x == 0)                                                                \
    stamps[k] = (long long)__builtin_amdgcn_s_memtime()
    BQ_STAMP(0);
    double *slots = lds;
    double *blk = lds + BQ_POTF2F_SLOTS;
    int *cntp = reinterpret_cast<int *>(lds + BQ_POTF2F_SLOTS + 1024);
    const unsigned cnt = lds_addr(cntp);
    // lane 0 stores the counter, the other lanes hit a word of their own behind it
    // (word 1 = "a wait gave up")
    const unsigned cnt_store = cnt + (lane == 0 ? 0u : 8u + 4u * (unsigned)lane);
    Potf2F st;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            st.a[q][s] = src ? src[lane + (long)(16 * q + 4 * w + s) * lsrc]
                             : Ab[lane + (long)(16 * q + 4 * w + s) * lda];
    if (threadIdx.x == 0) {
        cntp[0] = 0;
        cntp[1] = 0;
    }
    if (SYNC || src_in_slots)
        __syncthreads();
    BQ_STAMP(1);
    if (w == 0)
        potf2f_factor<0, 0>(st, slots, lds_addr(slots) + 8u * (unsigned)lane, cnt_store, lane, SYNC);
    Potf2FSteps<SYNC, 0>::run(st, slots, cnt, cnt_store, w, lane);
    BQ_STAMP(2);
    // the four 16 x 16 diagonal sub-blocks into LDS from registers:
    // blk[b][i + 16 k] = L[16 b + i][16 b + k]; my columns: k = 4 w + s of every block
    {
        const int bq = lane >> 4, i16 = lane & 15;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int sc = 0; sc < 4; ++sc)
                if (bq == q)
                    blk[256 * q + i16 + 16 * (4 * w + sc)] = st.a[q][sc];
    }
    __syncthreads();
    BQ_STAMP(3);
    // write back the lower triangle of my columns (the stores drain under the inverses)
    {
        double *pw = Ab + lane + (long)(4 * w) * lda;
        asm volatile("" : "+v"(pw));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (lane >= 16 * q + 4 * w + s)
                    pw[(long)s * lda] = st.a[q][s];
            pw += 16 * lda;
        }
    }
    // reciprocal pivots (lane = column) and the failure report: a non-positive pivot left a
    // NaN on the diagonal at its own column and at every later one
    {
        const double dg = blk[256 * (lane >> 4) + 17 * (lane & 15)];
        const double rc = 1.0 / dg;
        if (w == 0) {
            dinv_b[lane] = rc;
            const unsigned long long badm = __ballot(!(dg > 0.0) || !(dg < 1.7e308));
            if (lane == 0 && badm != 0ull && info_b[0] == 0)
                info_b[0] = j0 + __builtin_ctzll(badm) + 1;
            if (SYNC && lane == 0 && cntp[1] != 0)
                info_b[0] = j0 + 65;
        }
        // wave w inverts block w: lane j < 16 runs the forward substitution of unit column j
        const double *bw = blk + 256 * w;
        double wc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double sacc[4] = {(i == (lane & 15)) ? 1.0 : 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < i; ++k)
                sacc[k & 3] -= bw[i + 16 * k] * wc[k];
            wc[i] = ((sacc[0] + sacc[1]) + (sacc[2] + sacc[3])) * readlane_f64(rc, 16 * w + i);
        }
        if (lane < 16) {
            double *Wb = dinv_b + 64 + 256 * w + 16 * lane;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                Wb[i] = wc[i];
        }
    }
    BQ_STAMP(4);
#undef BQ_STAMP
}

template <int SYNC>
__global__ __launch_bounds__(256) void potf2f_kernel(double *__restrict__ A, long lda,
                                                     long astride, int j0,
                                                     double *__restrict__ dinv, long dstride,
                                                     int *__restrict__ info)
{
    __shared__ __attribute__((aligned(16))) double lds[BQ_POTF2F_LDS_DOUBLES];
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.z;
    potf2f_body<SYNC>(A + (long)b * astride + j0 + (long)j0 * lda, lda, j0,
                      dinv + (long)b * dstride, info + b, lds);
}

__global__ __launch_bounds__(256) void potf2_64x4_kernel(double *__restrict__ A, long lda,
                                                         long astride, int j0,
                                                         double *__restrict__ dinv, long dstride,
                                                         int *__restrict__ info)
{
    __shared__ __attribute__((aligned(16))) double ring[4 * 4 * 64 + 64]; // three panel slots; then the four 16x16 blocks + 64 pivots
    __shared__ int sbad[4];
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.z;
    potf2_64x4_body(A + (long)b * astride + j0 + (long)j0 * lda, lda, j0, dinv + (long)b * dstride,
                    info + b, ring, sbad);
}

// ---------------------------------------------------------------------------
// The diagonal factor every fused kernel calls.  BQ_POTF2_VAR picks the body at build time
// (0: the first-generation ring + per-pivot bookkeeping, 1: potf2f with a barrier per panel,
// 2: potf2f with published-column counters); the shipped library is built with the default.
// ---------------------------------------------------------------------------
#ifndef BQ_POTF2_VAR
#define BQ_POTF2_VAR 2
#endif
#if BQ_POTF2_VAR == 0
#define BQ_POTF2_LDS_DOUBLES (4 * 4 * 64 + 64 + 8)
#else
#define BQ_POTF2_LDS_DOUBLES BQ_POTF2F_LDS_DOUBLES
#endif
// whether the factor's first 4096 LDS doubles may hold its own input block
#define BQ_POTF2_SRC_IN_LDS (BQ_POTF2_VAR != 0)

__device__ __forceinline__ void potf2_body(double *__restrict__ Ab, long lda, int j0,
                                           double *__restrict__ dinv_b, int *__restrict__ info_b,
                                           double *lds, const double *src = nullptr,
                                           long lsrc = 0, long long *stamps = nullptr)
{
#if BQ_POTF2_VAR == 0
    potf2_64x4_body(Ab, lda, j0, dinv_b, info_b, lds, reinterpret_cast<int *>(lds + 1088), src,
                    lsrc);
#elif BQ_POTF2_VAR == 1
    potf2f_body<0>(Ab, lda, j0, dinv_b, info_b, lds, src, lsrc, src == lds, stamps);
#else
    potf2f_body<1>(Ab, lda, j0, dinv_b, info_b, lds, src, lsrc, src == lds, stamps);
#endif
}

__global__ __launch_bounds__(256) void potf2_kernel(double *__restrict__ A, long lda, long astride,
                                                    int j0, double *__restrict__ dinv,
                                                    long dstride, int *__restrict__ info)
{
    __shared__ __attribute__((aligned(16))) double lds[BQ_POTF2_LDS_DOUBLES];
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.z;
    potf2_body(A + (long)b * astride + j0 + (long)j0 * lda, lda, j0, dinv + (long)b * dstride,
               info + b, lds);
}

// timing probe: the factor alone on a block that is restored from Ain every launch;
// stamps[0..4] of the last launch = s_memtime at entry / loaded / chain done / blocks in LDS / end
__global__ __launch_bounds__(256) void potf2_probe_kernel(const double *__restrict__ Ain,
                                                          double *__restrict__ A, long lda,
                                                          double *__restrict__ dinv,
                                                          int *__restrict__ info,
                                                          long long *stamps, int from_lds)
{
    __shared__ __attribute__((aligned(16))) double lds[BQ_POTF2_LDS_DOUBLES < 4096 + 1100
                                                           ? 4096 + 1100
                                                           : BQ_POTF2_LDS_DOUBLES];
    __builtin_amdgcn_s_setprio(3);
    if (from_lds) {
        // as the slab step hands the block over: through LDS
        double *Ts = BQ_POTF2_SRC_IN_LDS ? lds : lds + 1100;
        for (int e = threadIdx.x; e < 4096; e += 256)
            Ts[e] = Ain[(e & 63) + (long)(e >> 6) * lda];
        __syncthreads();
        potf2_body(A, lda, 0, dinv, info, lds, Ts, 64, stamps);
    } else {
        for (int e = threadIdx.x; e < 4096; e += 256)
            A[(e & 63) + (long)(e >> 6) * lda] = Ain[(e & 63) + (long)(e >> 6) * lda];
        __syncthreads();
        potf2_body(A, lda, 0, dinv, info, lds, nullptr, 0, stamps);
    }
}
