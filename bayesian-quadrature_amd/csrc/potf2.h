// potf2.h -- the 64x64 diagonal Cholesky block (four waves) and its block inverses
// Part of the libbqhip.so kernel set; included through kernels.h.
#pragma once
#include "common.h"

// ===========================================================================
// 64 x 64 diagonal Cholesky block by FOUR waves (one per SIMD).  Every wave holds all 64
// rows (lane = row) and a quarter of the columns: wave w owns the columns 16q + 4w + s
// (q, s = 0..3), i.e. the matrix is cut into sixteen 4-column panels dealt round-robin to the
// waves.  Panel p is factored by its owner (in-panel updates by v_readlane) and published to
// LDS; every wave then applies the rank-4 update to its own later columns.  The owner of
// panel p+1 updates only that panel before starting its pivot chain and catches up on its
// remaining columns afterwards.  Everything that is not the pivot chain is off the owner:
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

// slots_a: LDS address of this lane's entry of column 0 of slot 0; SLOT: the panel's slot.
// The dependent chain of a pivot, in instruction hops (an fp64 hop costs 12-16 cycles):
//   readlane d -> rsq y0 -> {t = d y0, l0 = a y0} -> e = 1 - t y0 -> {p = 1/2 + 3/8 e, m = l0 e}
//   -> l = l0 + m p -> dn = a' - l l (the next pivot's diagonal, in its own lane) -> readlane.
// The scaled column is formed directly (l = a y1 = l0 + l0 e p), never the refined
// reciprocal, and the next diagonal entry needs no broadcast: every lane squares its own l.
template <int P, int SLOT>
__device__ __forceinline__ void potf2f_factor(Potf2F &st, double *slot, unsigned slots_a,
                                              unsigned cnt_store, int lane, int sync)
{
    constexpr int QP = P >> 2;
    double dn = st.a[QP][0];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 4 * P + s;
        const double d = readlane_f64(dn, c);
        const double y0 = __builtin_amdgcn_rsq(d);
        const double t = d * y0;
        const double l0 = st.a[QP][s] * y0;
        const double e = __builtin_fma(-t, y0, 1.0);
        const double p = __builtin_fma(0.375, e, 0.5);
        const double m = l0 * e;
        const double l = __builtin_fma(m, p, l0);
        if (s < 3)
            dn = __builtin_fma(-l, l, st.a[QP][s + 1]);
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

#ifndef BQ_POLL_SLEEP
#define BQ_POLL_SLEEP 0
#endif
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
#if BQ_POLL_SLEEP >= 0
        __builtin_amdgcn_s_sleep(BQ_POLL_SLEEP);
#endif
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

// ---------------------------------------------------------------------------
// The diagonal factor every fused kernel calls.  BQ_POTF2_VAR picks the hand-over at build
// time (1: a ring of three slots and one workgroup barrier per panel -- shipped; 2: a slot
// per panel and published-column counters, measured slower: 16.6 vs 12.7 us per launch).
// ---------------------------------------------------------------------------
#ifndef BQ_POTF2_VAR
#define BQ_POTF2_VAR 1
#endif
#define BQ_POTF2_LDS_DOUBLES BQ_POTF2F_LDS_DOUBLES

__device__ __forceinline__ void potf2_body(double *__restrict__ Ab, long lda, int j0,
                                           double *__restrict__ dinv_b, int *__restrict__ info_b,
                                           double *lds, const double *src = nullptr,
                                           long lsrc = 0, long long *stamps = nullptr)
{
#if BQ_POTF2_VAR == 1
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
    __shared__ __attribute__((aligned(16))) double lds[BQ_POTF2_LDS_DOUBLES];
    __builtin_amdgcn_s_setprio(3);
    if (from_lds) {
        // as the slab step hands the block over: through LDS
        double *Ts = lds; // where the panel slots will be
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
