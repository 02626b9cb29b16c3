// potf2.h -- the 64x64 diagonal Cholesky block (four waves) and its block inverses
// Part of the libbqhip.so kernel set; compiled into k_gemm.hip / k_panel.hip / probe.hip (host.h lists the units).
#pragma once
#include "common.h"

// ===========================================================================
// 64 x 64 diagonal Cholesky block by FOUR waves (one per SIMD).  Every wave holds all 64
// rows (lane = row) and a quarter of the columns: wave w owns the columns 16q + 4w + s
// (q, s = 0..3), i.e. the matrix is cut into sixteen 4-column panels dealt round-robin to the
// waves.  Panel p is factored by its owner (in-panel updates by v_readlane) and published to
// its LDS slot (a full block keeps all sixteen -- the epilogue reads the factor out of them --, a
// padded one a ring of three); after ONE workgroup barrier per panel every wave applies the
// rank-4 update to its own later columns.  What was measured on gfx950 and shaped this file
// (tools/potf2_probe.py, per-wave barrier stamps):
//   * a wave's time per panel is set by the waves that only update in the early panels
//     (64 FMAs + 36 LDS reads, ~1100 cycles when every group waited for its own reads) and by
//     the owner's path (own-panel update + four pivots, ~740 cycles) in the late ones -- not
//     by the instruction count of the pivot: Newton vs Halley, per-pivot selects or not,
//     moved the 15,000-cycle chain by 2 %;
//   * published-column counters polled in LDS instead of barriers (every panel its own slot,
//     the next owner applying columns as they appear) were SLOWER: 24,500 cycles;
//   * the block inverses and the write-back after the chain were 5,800 cycles, 41 LDS waits.
// ===========================================================================
// NW waves (4, or 8: two per SIMD -- potf2f_body): panel P = columns 4P .. 4P + 3 belongs to wave
// P % NW, which holds it as its group P / NW
template <int NW> struct Potf2FT {
    double a[16 / NW][4]; // a[g][s] = column 4 (NW g + w) + s of row `lane`
};
using Potf2F = Potf2FT<4>;

// LDS of the factor: the panels' slots (4 x 64 doubles each: sixteen, or a ring of three) in the first 4096
// doubles -- the region may hold the factor's own input block, which every wave has in
// registers before the first panel is published --, then the four 16 x 16 diagonal
// sub-blocks (1024 doubles)
#define BQ_POTF2F_SLOTS (16 * 256)
#ifndef BQ_POTF2F_EARLY
#define BQ_POTF2F_EARLY 1
#endif
#define BQ_POTF2F_LDS_DOUBLES (BQ_POTF2F_SLOTS + 1024)

// Factor panel P (columns 4P .. 4P+3, group QP = P >> 2) held by this wave and publish it.
// The dependent chain of a pivot, in instruction hops (an fp64 hop costs 12-16 cycles):
//   readlane d -> rsq y0 -> {t = d y0, l0 = a y0} -> e = 1 - t y0 -> {p = 1/2 + 3/8 e, m = l0 e}
//   -> l = l0 + m p -> dn = a' - l l (the next pivot's diagonal, in its own lane) -> readlane.
// The scaled column is formed directly (l = a y1 = l0 + l0 e p, a third-order step from the
// v_rsq_f64 seed: error 5 e^3 / 16 ~ 1e-21 for the 2^-24 seed measured on gfx950), never the
// refined reciprocal, and the next diagonal entry needs no broadcast: every lane squares its
// own l.  The pivot row is scaled like every other row (L_cc = d r, about an ulp from sqrt d):
// no per-pivot selects, no failure bookkeeping -- a non-positive pivot makes r NaN or infinite
// and everything after it NaN, and potf2f_body finds it afterwards as the first NaN on the
// diagonal.
template <int P, int NW = 4>
__device__ __forceinline__ void potf2f_factor(Potf2FT<NW> &st, double *slot, int lane)
{
    constexpr int QP = P / NW;
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
        slot[s * 64 + lane] = l;
    }
}

// Rank-4 update of this wave's columns of the groups in QMASK by the panel in `slot` (li: the
// panel's entries in this lane's row).  All multipliers of all groups are requested before
// the first FMA (uniform ds_read_b128, two columns each): one LDS latency per call instead
// of one per group -- the waves that only update were the slow ones of the early panels.
template <int QMASK, int NW = 4>
__device__ __forceinline__ void potf2f_update(Potf2FT<NW> &st, const double *slot,
                                              const double (&li)[4], int w)
{
    constexpr int NG = 16 / NW;
    double2_t lk[NG][4][2];
#pragma unroll
    for (int q = 0; q < NG; ++q)
        if (QMASK & (1 << q)) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const double2_t *src =
                    reinterpret_cast<const double2_t *>(slot + s * 64 + 4 * (NW * q + w));
                lk[q][s][0] = src[0];
                lk[q][s][1] = src[1];
            }
        }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < NG; ++q)
            if (QMASK & (1 << q)) {
                st.a[q][0] = __builtin_fma(-li[s], lk[q][s][0][0], st.a[q][0]);
                st.a[q][1] = __builtin_fma(-li[s], lk[q][s][0][1], st.a[q][1]);
                st.a[q][2] = __builtin_fma(-li[s], lk[q][s][1][0], st.a[q][2]);
                st.a[q][3] = __builtin_fma(-li[s], lk[q][s][1][1], st.a[q][3]);
            }
#pragma unroll
    for (int q = 0; q < NG; ++q)
        if (QMASK & (1 << q)) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                PIN(st.a[q][cc]);
        }
}

// groups after Q of NG: bits Q+1 .. NG-1
#define BQ_LATER(Q, NG) (((1 << (NG)) - 1) & ~((1 << ((Q) + 1)) - 1))

// ---------------------------------------------------------------------------
// The epilogue of a full block's factor (BQ_POTF2F_EARLY; tools/potf2_probe.py, potf2_waves.py).
// After the last pivot the factor used to copy the four 16 x 16 diagonal sub-blocks to LDS, pass a
// barrier, and run the reciprocal pivots and the four block inverses beside the write-back: 6,000
// cycles behind a chain of 13,000, most of them the inverses -- sixteen dependent steps, each lane
// of a 16-lane group running the same 120 FMAs as the other three groups' lanes.  Now:
//   * every panel keeps a slot of its own in LDS (16 x 256 doubles -- the region is there; the ring
//     of three was all the chain needs), so nothing is copied and no barrier follows the chain;
//   * wave b < 4 inverts sub-block b in four slices of four steps: group g = lane / 16 keeps the rows
//     4 g .. 4 g + 3 of unit column lane % 16, the group that holds a slice's pivot rows solves
//     their 4 x 4 triangle, the four multipliers cross to the other groups by ds_bpermute in one
//     go, and the groups behind apply them -- 6 + 16 FMAs per lane and slice, the same operations
//     on the same operands in the same order as the sixteen-step form;
//   * waves 4-7 write the factor home out of the slots; wave 4 also takes the reciprocal pivots and
//     the failure report, wave 5 log|K|.
// Measured and dropped: the same slices INSIDE the chain, on the waves that have run out of columns
// (wave w from step 8 + w on; they reach every later barrier ~500 cycles before the panel's owner).
// Whatever those waves did -- slices of 2, 3 or 4 steps, SIMD-aware placement, lower priority, no
// global stores before the last barrier -- stretched the owners' steps by 150-300 cycles each: the
// chain grew from 13,700 to 15,500-16,500 cycles and the slab step gained 0.14 us where the factor
// alone gained 1.0.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double potf2f_rcp(double dg)
{
    // 1 / L_cc: v_rcp_f64 + two Newton steps (the IEEE division sequence is three times as long)
    double rc = __builtin_amdgcn_rcp(dg);
    rc = __builtin_fma(__builtin_fma(-dg, rc, 1.0), rc, rc);
    rc = __builtin_fma(__builtin_fma(-dg, rc, 1.0), rc, rc);
    return rc;
}

// steps 4 S .. 4 S + 3 of sub-block b's inversion.  Column 4 S + s of the sub-block is column s of
// panel 4 b + S, rows 16 b .. 16 b + 15, in that panel's slot.  rcv[r]: 1 / L_ii of this lane's row
// i = 4 g + r.  Every lane runs the pivot rows' 4 x 4 substitution on a copy of ITS rows with ITS
// rows' entries (mb) -- in group S those are the pivot rows and the pivot triangle, and it is group
// S's result that crosses to the others.
template <int S>
__device__ __forceinline__ void potf2f_inv_slice(double (&sv4)[4], double (&wc)[12], double (&wl)[4],
                                                 const double *slots, int b,
                                                 const double (&rcv)[4], int lane)
{
    const int g = lane >> 4, j = lane & 15;
    const double *base = slots + (256 * (4 * b + S) + 16 * b);
    double mb[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (S < 3 || r > s)
                mb[s][r] = base[64 * s + 4 * g + r];
    double t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        t[r] = sv4[r];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        wl[s] = t[s] * rcv[s];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r > s)
                t[r] = __builtin_fma(-mb[s][r], wl[s], t[r]);
    }
    if (S == 3)
        return; // the last four rows of W stay in group 3: nobody is left to apply them
    double wk[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        wk[s] = __shfl(wl[s], 16 * S + j, 64);
        wc[4 * S + s] = wk[s];
    }
    // (every group applies them: the rows of the groups up to S are never read again, and
    // without the branch the compiler is free to issue the next slice's LDS reads early)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            sv4[r] = __builtin_fma(-mb[s][r], wk[s], sv4[r]);
}

// W_b = inverse of sub-block b by one wave, out of the panels' slots
__device__ __forceinline__ void potf2f_inverse(const double *slots, int b, double *__restrict__ Wb,
                                               int lane)
{
    const int g = lane >> 4, j = lane & 15;
    // 1 / L_tt of the sub-block's column t in lane t, then this lane's rows' four
    const double rc = potf2f_rcp(slots[256 * (4 * b + (j >> 2)) + 64 * (j & 3) + 16 * b + j]);
    double rcv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        rcv[r] = __shfl(rc, 4 * g + r, 64);
    double sv4[4], wc[12], wl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        sv4[r] = (4 * g + r == j) ? 1.0 : 0.0;
    potf2f_inv_slice<0>(sv4, wc, wl, slots, b, rcv, lane);
    potf2f_inv_slice<1>(sv4, wc, wl, slots, b, rcv, lane);
    potf2f_inv_slice<2>(sv4, wc, wl, slots, b, rcv, lane);
    potf2f_inv_slice<3>(sv4, wc, wl, slots, b, rcv, lane);
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 12; ++i)
            Wb[16 * lane + i] = wc[i];
    }
    if (lane >= 48) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            Wb[16 * j + 12 + s] = wl[s];
    }
}

// sum over the wave, the same value in every lane: DPP row_shr within the rows of 16, the four row
// sums by v_readlane (a __shfl_down tree is six LDS round trips for a double)
__device__ __forceinline__ double potf2f_wave_sum(double v)
{
#define BQ_DPP_ADD(CTRL)                                                                           \
    {                                                                                              \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);    \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);    \
        v += __hiloint2double(hi, lo);                                                             \
    }
    BQ_DPP_ADD(0x111) // row_shr:1
    BQ_DPP_ADD(0x112)
    BQ_DPP_ADD(0x114)
    BQ_DPP_ADD(0x118)
#undef BQ_DPP_ADD
    return (readlane_f64(v, 15) + readlane_f64(v, 31)) + (readlane_f64(v, 47) + readlane_f64(v, 63));
}

// Step P, entered with panel P factored by its owner (wave P & 3) and on its way to slot
// P % 3 of the ring (EARLY: slot P); ONE workgroup barrier per panel:
//   * the owner of panel P + 1 brings only that panel up to date, runs its pivot chain and
//     publishes; what its later groups owe panel P waits until step P + 1;
//   * the owner of panel P pays that debt for panel P - 1 (slot (P - 1) % 3 is not reused
//     before step P + 1) and applies its own panel P to its later groups;
//   * the other two waves apply panel P to everything of theirs that lies behind it.
// wst (profiling probe only): every wave's arrival at and release from barrier P.
template <int P, int NW = 4, bool PAD = false, bool EARLY = false>
struct Potf2FSteps {
    // PAD, nreal: the block's rows / columns from nreal on are identity padding (a system whose
    // size is not a multiple of 64): panels that lie wholly in it are neither factored nor applied
    // -- they are what they will be -- and the sweep over the panels ends with the last real one.
    // (An instantiation of its own: the tests cost the full block's straight-line code 0.9 us.)
    static __device__ __forceinline__ void run(Potf2FT<NW> &st, double *slots, int w, int lane,
                                               long long *wst, int nreal)
    {
        constexpr int NG = 16 / NW;
        constexpr int QP = P / NW, WP = P % NW;
        constexpr int PN = P < 15 ? P + 1 : 15, QN = PN / NW, WN = PN % NW;
        constexpr int LATER = BQ_LATER(QP, NG);
        // (EARLY: every panel keeps a slot of its own)
        const double *slot = slots + (EARLY ? P : P % 3) * 256;
        // (eight waves: arrivals only, wst[8 P + w])
        if (wst && lane == 0)
            wst[NW == 8 ? 8 * P + w : (NW * P + w) * 2] = (long long)__builtin_amdgcn_s_memtime();
        __syncthreads(); // panel P is published
        if (NW == 4 && wst && lane == 0)
            wst[(NW * P + w) * 2 + 1] = (long long)__builtin_amdgcn_s_memtime();
        double li[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
            li[s] = slot[s * 64 + lane];
        if (P < 15 && w == WN) {
            if (!PAD || 4 * PN < nreal) {
                potf2f_update<(1 << QN), NW>(st, slot, li, w);
                potf2f_factor<PN, NW>(st, slots + (EARLY ? PN : PN % 3) * 256, lane);
            }
        } else if (w == WP) {
            if (P >= 1 && LATER != 0) {
                const double *prev = slots + (EARLY ? P - 1 : (P + 2) % 3) * 256;
                double lp[4];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    lp[s] = prev[s * 64 + lane];
                potf2f_update<LATER, NW>(st, prev, lp, w);
            }
            potf2f_update<LATER, NW>(st, slot, li, w);
        } else if (w > WP) {
            potf2f_update<(1 << QP) | LATER, NW>(st, slot, li, w);
        } else {
            potf2f_update<LATER, NW>(st, slot, li, w);
        }
        if (!PAD || 4 * (P + 1) < nreal)
            Potf2FSteps<P + 1, NW, PAD, EARLY>::run(st, slots, w, lane, wst, nreal);
    }
};
template <int NW, bool PAD, bool EARLY>
struct Potf2FSteps<16, NW, PAD, EARLY> {
    static __device__ __forceinline__ void run(Potf2FT<NW> &, double *, int, int, long long *, int)
    {
    }
};

// lds: BQ_POTF2F_LDS_DOUBLES doubles.  src (leading dimension lsrc): where the block is read
// from when it is not in place; when src lies in the slots' LDS (the slab step's Ts), pass
// src_in_slots so that nobody publishes before every wave has its columns.
// NW = 8: the workgroup has EIGHT waves (512 threads), two per SIMD, two panels each: the waves
// that only update have half the columns to bring up to date per panel (32 FMAs + 20 LDS reads
// instead of 64 + 36), which is what paces the early panels.
template <int NW = 4>
__device__ __forceinline__ void potf2f_body(double *__restrict__ Ab, long lda, int j0,
                                            double *__restrict__ dinv_b,
                                            int *__restrict__ info_b, double *lds,
                                            const double *src = nullptr, long lsrc = 0,
                                            bool src_in_slots = false,
                                            long long *stamps = nullptr,
                                            double *logdet = nullptr, int nreal = 64)
{
    constexpr int NG = 16 / NW;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#define BQ_STAMP(k)                                                                                \
    if (stamps && threadIdx.x == 0)                                                                \
    stamps[k] = (long long)__builtin_amdgcn_s_memtime()
    BQ_STAMP(0);
    double *slots = lds;
    double *blk = lds + BQ_POTF2F_SLOTS;
    Potf2FT<NW> st;
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            st.a[q][s] = src ? src[lane + (long)(4 * (NW * q + w) + s) * lsrc]
                             : Ab[lane + (long)(4 * (NW * q + w) + s) * lda];
    if (src_in_slots)
        __syncthreads(); // the ring overwrites the block: every wave has its columns first
    BQ_STAMP(1);
    if (w == 0)
        potf2f_factor<0, NW>(st, slots, lane);
    // (per-wave barrier stamps: the four-wave probe only, and only when it asks for them with
    // stamps[5] != 0 -- they cost the chain 2,500 cycles)
    // (only the assembly's first factor passes nreal -- a system of fewer than 64 points has no
    // other --: inside slab_step_kernel the second instantiation cost C2 0.4 us per step)
    if (BQ_POTF2F_EARLY && nreal >= 64) {
        // (a full block: the epilogue straight out of the panels' slots)
        const double ld0 = logdet ? *logdet : 0.0; // one writer per launch, launches in order
        Potf2FSteps<0, NW, false, true>::run(
            st, slots, w, lane,
            (stamps && stamps[5] != 0) ? stamps + 8 : nullptr, 64);
        BQ_STAMP(2);
        BQ_STAMP(3);
        // The factor goes home (the stores drain under the inverses).  Eight waves: waves 0-3 go
        // straight to the inverses and waves 4-7 write sixteen columns each out of the panels'
        // slots; four waves: every wave its own columns out of its registers.
        if (NW == 8 && w >= 4) {
            const int c0 = 16 * (w - 4);
            double *pw = Ab + lane + (long)c0 * lda;
            const double *ps = slots + 64 * c0 + lane; // column c: panel c / 4, its column c % 4
            asm volatile("" : "+v"(pw));
            double col[16];
#pragma unroll
            for (int c = 0; c < 16; ++c)
                col[c] = ps[64 * c];
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (lane >= c0 + c)
                    pw[(long)c * lda] = col[c];
        }
        if (NW == 4) {
            double *pw = Ab + lane + (long)(4 * w) * lda;
            asm volatile("" : "+v"(pw));
#pragma unroll
            for (int q = 0; q < NG; ++q) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (lane >= 4 * (NW * q + w) + s)
                        pw[(long)s * lda] = st.a[q][s];
                pw += 4 * NW * lda;
            }
        }
        // reciprocal pivots + failure report and log|K|: waves 4 and 5 of eight, else 1 and 2
        constexpr int WR = NW == 8 ? 4 : 1, WL = NW == 8 ? 5 : 2;
        if (w < 4)
            potf2f_inverse(slots, w, dinv_b + 64 + 256 * w, lane);
        if (w == WR) {
            const double dg = slots[256 * (lane >> 2) + 64 * (lane & 3) + lane];
            dinv_b[lane] = potf2f_rcp(dg);
            const unsigned long long badm = __ballot(!(dg > 0.0) || !(dg < 1.7e308));
            if (lane == 0 && badm != 0ull && info_b[0] == 0)
                info_b[0] = j0 + __builtin_ctzll(badm) + 1;
        }
        if (w == WL && logdet) {
            const double lg = potf2f_wave_sum(log(slots[256 * (lane >> 2) + 64 * (lane & 3) + lane]));
            if (lane == 0)
                *logdet = ld0 + 2.0 * lg;
        }
        BQ_STAMP(4);
        return;
    }
    if (nreal < 64)
        Potf2FSteps<0, NW, true>::run(st, slots, w, lane, nullptr, nreal > 0 ? nreal : 1);
    else
        Potf2FSteps<0, NW, false>::run(st, slots, w, lane,
                                       (stamps && NW == 4 && stamps[5] != 0) ? stamps + 8
                                                                              : nullptr,
                                       64);
    BQ_STAMP(2);
    // the four 16 x 16 diagonal sub-blocks into LDS from registers:
    // blk[b][i + 16 k] = L[16 b + i][16 b + k]; my columns c = 4 (NW q + w) + sc: block c >> 4
    {
        const int bq = lane >> 4, i16 = lane & 15;
#pragma unroll
        for (int q = 0; q < NG; ++q)
#pragma unroll
            for (int sc = 0; sc < 4; ++sc) {
                const int c = 4 * (NW * q + w) + sc;
                if (bq == (c >> 4))
                    blk[256 * (c >> 4) + i16 + 16 * (c & 15)] = st.a[q][sc];
            }
    }
    __syncthreads();
    BQ_STAMP(3);
    // write back the lower triangle of my columns (the stores drain under the inverses)
    {
        double *pw = Ab + lane + (long)(4 * w) * lda;
        asm volatile("" : "+v"(pw));
#pragma unroll
        for (int q = 0; q < NG; ++q) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (lane >= 4 * (NW * q + w) + s)
                    pw[(long)s * lda] = st.a[q][s];
            pw += 4 * NW * lda;
        }
    }
    // This block's share of log|K| = 2 sum log L_cc (SlabOut: the read-out rides in the sweep;
    // one writer per launch, launches in order) -- an fp64 log is ~400 cycles of one wave: with
    // eight waves wave 4 takes it while waves 0-3 run the inverses, with four it waits until
    // wave 3 is through with its inverse (BQ_POTF2F_LOGDET below)
#define BQ_POTF2F_LOGDET                                                                           \
    {                                                                                              \
        double lg = log(blk[256 * (lane >> 4) + 17 * (lane & 15)]);                                \
        _Pragma("unroll") for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off, 64);   \
        if (lane == 0)                                                                             \
            *logdet += 2.0 * lg;                                                                   \
    }
    if (NW > 4 && w >= 4) {
        if (logdet && w == 4)
            BQ_POTF2F_LOGDET
        return; // (no barrier follows: the reciprocals and the four inverses are waves 0-3's)
    }
    // reciprocal pivots (lane = column) and the failure report: a non-positive pivot left a
    // NaN on the diagonal at its own column and at every later one
    {
        const double dg = blk[256 * (lane >> 4) + 17 * (lane & 15)];
        // 1 / L_cc: v_rcp_f64 + two Newton steps (the IEEE division sequence is three times
        // as long, and this sits between the pivot chain and the end of the launch)
        double rc = __builtin_amdgcn_rcp(dg);
        rc = __builtin_fma(__builtin_fma(-dg, rc, 1.0), rc, rc);
        rc = __builtin_fma(__builtin_fma(-dg, rc, 1.0), rc, rc);
        if (w == 0) {
            dinv_b[lane] = rc;
            const unsigned long long badm = __ballot(!(dg > 0.0) || !(dg < 1.7e308));
            if (lane == 0 && badm != 0ull && info_b[0] == 0)
                info_b[0] = j0 + __builtin_ctzll(badm) + 1;
        }
        // Wave w inverts block w: lane j (mod 16) runs the forward substitution of unit column
        // j in its right-looking form -- w_k = s_k / L_kk, then s_i -= L_ik w_k for i > k --
        // so that the dependent chain is two operations per row and column k + 1 of the block
        // (uniform LDS reads) is in flight while column k is applied.  (The row-by-row dot
        // products it replaces waited for LDS 41 times: 5,800 cycles for this epilogue.)
        const double *bw = blk + 256 * w;
        double sv[16], wc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            sv[i] = (i == (lane & 15)) ? 1.0 : 0.0;
        double cur[16], nxt[16];
#pragma unroll
        for (int i = 1; i < 16; ++i)
            cur[i] = bw[i];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < 14) {
#pragma unroll
                for (int i = k + 2; i < 16; ++i)
                    nxt[i] = bw[i + 16 * (k + 1)];
            }
            wc[k] = sv[k] * readlane_f64(rc, 16 * w + k);
#pragma unroll
            for (int i = k + 1; i < 16; ++i)
                sv[i] = __builtin_fma(-cur[i], wc[k], sv[i]);
#pragma unroll
            for (int i = k + 2; i < 16; ++i)
                cur[i] = nxt[i];
        }
        if (lane < 16) {
            double *Wb = dinv_b + 64 + 256 * w + 16 * lane;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                Wb[i] = wc[i];
        }
    }
    if (NW == 4 && logdet && w == 3)
        BQ_POTF2F_LOGDET
#undef BQ_POTF2F_LOGDET
    BQ_STAMP(4);
#undef BQ_STAMP
}

// ---------------------------------------------------------------------------
// The diagonal factor as every fused kernel calls it.  lds: BQ_POTF2_LDS_DOUBLES doubles; src
// (leading dimension lsrc) is where the block is read from when it is not in place -- the
// one-launch steps hand it over in the first 4096 doubles of `lds` itself.
// ---------------------------------------------------------------------------
#define BQ_POTF2_LDS_DOUBLES BQ_POTF2F_LDS_DOUBLES

template <int NW = 4>
__device__ __forceinline__ void potf2_body(double *__restrict__ Ab, long lda, int j0,
                                           double *__restrict__ dinv_b, int *__restrict__ info_b,
                                           double *lds, const double *src = nullptr,
                                           long lsrc = 0, long long *stamps = nullptr,
                                           double *logdet = nullptr, int nreal = 64)
{
    potf2f_body<NW>(Ab, lda, j0, dinv_b, info_b, lds, src, lsrc, src == lds, stamps, logdet,
                    nreal);
}
