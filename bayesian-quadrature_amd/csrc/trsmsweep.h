// trsmsweep.h -- the batched panel solve X <- X L11^-T of a whole outer block in ONE launch
// Part of the libbqhip.so kernel set; compiled into k_gemm.hip (host.h lists the units).
#pragma once
#include "gemm.h"

// ---------------------------------------------------------------------------
// potrf.hip, enqueue_potrf_dfirst: once the KB x KB diagonal block L11 of an outer block is
// factored (with the block inverses of its 64 x 64 diagonal blocks left behind in records), the
// rows below it take no part in one another's solve:
//     X_s = (A_s - sum_{t<s} X_t L_st^T) L_ss^-T,   s = 0 .. KB / 64 - 1 (64-column slabs).
// A workgroup owns 64 rows and walks the slabs itself, left-looking: the product over the slabs
// it has already solved runs on gemm_lds64_kernel's 64 x 64 tile (LDS-DMA staging, the 4x4x4
// MFMA, pre-rotated views; P = its own solved rows, read back from global memory -- the same
// CU wrote them --, Q = rows 64 s .. of L11), the solve against L_ss on trsm_blk_kernel's
// scheme with the tile handed through LDS into that kernel's 16-rows-per-wave operand form.
// One launch per outer block instead of a chain of ~13 products and solves, one read and one
// write of the panel from HBM instead of ~30 slab passes.  36 KiB of LDS, <= 128 VGPRs: four
// workgroups per CU hide each other's slab boundaries.
// Placement: every workgroup of a matrix re-reads the same rows of L11 (Q), slab by slab and at
// about the same time; dealt over the chip as they come (consecutive workgroup ids go round-robin
// over the eight XCDs) each XCD fetches every matrix's L11 for itself and the launch is bound by
// the fabric (4 TB/s measured into LDS).  The 1-D grid is therefore cut by XCD: workgroup id w runs
// on XCD w % 8 (observed dispatch, speed only) and takes row block (w / 8) % nrb of matrix
// w % 8 + 8 ((w / 8) / nrb) -- a matrix's row blocks share one L2.
// grid (8 nrb ceil(batch / 8)), block 256; m a multiple of 64, kb of 64.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void trsm_sweep_kernel(double *__restrict__ X, long ldx,
                                                            long xstride,
                                                            const double *__restrict__ L11, long ldl,
                                                            long lstride,
                                                            const double *__restrict__ rec,
                                                            long rstride, int kb, int nrb,
                                                            int batch
#ifdef BQ_TS_DBG
                                                            , int dbg // ablations (measurements only)
#endif
)
{
#ifndef BQ_TS_DBG
    constexpr int dbg = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int slot = blockIdx.x >> 3;
    const int b = (int)(blockIdx.x & 7) + 8 * (slot / nrb);
    if (b >= batch)
        return; // (the whole workgroup, before any barrier)
    const int rb = slot % nrb;
    X += (long)b * xstride;
    L11 += (long)b * lstride;
    rec += (long)b * rstride;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int R0 = rb * 64;
    const int wr = (wave & 1) * 32, wc = (wave >> 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;

    // staging of a chunk (16 k columns): 8 DMA rows of P and 8 of Q, wave w moves DMA rows
    // 4 (w & 1) .. + 3 of P (w < 2) or Q; lane i carries operand rows 2 (i & 31), + 1 of k row
    // (dma row) + 8 (i >> 5) -- gemm_lds64_body's layout
    const bool stq = wave >= 2;
    const int dma0 = 4 * (wave & 1);
    const double *gp = X + ((dbg & 4) ? 0 : R0) + 2 * (lane & 31) + (long)(dma0 + 8 * (lane >> 5)) * ldx;
    const double *gq0 = L11 + 2 * (lane & 31) + (long)(dma0 + 8 * (lane >> 5)) * ldl;
    const int srow = ((stq ? 8 : 0) + dma0) * BQ_LDS_ROW;
    const unsigned char *pview = smem + l4 * BQ_LDS_ROW + (wr + l15) * 8;
    const unsigned char *qview[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        qview[s] = smem + (8 + l4) * BQ_LDS_ROW + (wc + ((l15 - 4 * s) & 15)) * 8;

#define BQ_TS_FILL(BUF_, CH_)                                                                      \
    if (!(dbg & 8)) {                                                                              \
        const double *g_ = gsrc + (long)(CH_) * 16 * sld;                                          \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) __builtin_amdgcn_global_load_lds(            \
            (global_cvoid_t *)(g_ + (long)r * sld),                                                \
            (lds_void_t *)(smem + (BUF_) * BQ_L64_STAGE + srow + r * BQ_LDS_ROW), 16, 0, 0);       \
    }
#define BQ_TS_OFF(BUF_, ST_) ((BUF_) * BQ_L64_STAGE + 4 * ((ST_) & 1) * BQ_LDS_ROW + ((ST_) >> 1) * 512)
#define BQ_TS_READ_P(BUF_, ST_, PF)                                                                \
    _Pragma("unroll") for (int tm = 0; tm < 2; ++tm) PF[tm] = *reinterpret_cast<const double *>(   \
        pview + BQ_TS_OFF(BUF_, ST_) + tm * 128);
#define BQ_TS_READ_Q(BUF_, ST_, TN_, QF)                                                           \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) QF[s] = *reinterpret_cast<const double *>(       \
        qview[s] + BQ_TS_OFF(BUF_, ST_) + (TN_) * 128);
    // (the wait is explicit: hipcc orders an LDS-DMA only against the ISSUING wave's own LDS reads,
    // and whether a vmcnt(0) lands in front of this barrier -- for the other waves' reads -- is
    // luck of its placement: at the loop header of this kernel it did not)
#define BQ_TS_CHUNK(BUF_, CH_)                                                                     \
    {                                                                                              \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
        __syncthreads();                                                                           \
        if ((CH_) + 1 < nchunk)                                                                    \
            BQ_TS_FILL(1 - (BUF_), (CH_) + 1)                                                      \
        BQ_TS_READ_P(BUF_, 0, pf[0])                                                               \
        BQ_TS_READ_Q(BUF_, 0, 0, qf[0])                                                            \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                              \
        {                                                                                          \
            const int st = j >> 1, tn = j & 1;                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            asm volatile("" ::"v"(qf[j & 1][0]), "v"(qf[j & 1][1]), "v"(qf[j & 1][2]),             \
                         "v"(qf[j & 1][3]), "v"(pf[st & 1][0]), "v"(pf[st & 1][1]));               \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            if (j < 7)                                                                             \
                BQ_TS_READ_Q(BUF_, (j + 1) >> 1, (j + 1) & 1, qf[(j + 1) & 1])                     \
            if (tn == 1 && j < 7)                                                                  \
                BQ_TS_READ_P(BUF_, st + 1, pf[(st + 1) & 1])                                       \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            if (!(dbg & 16))                                                                       \
            _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                                       \
                _Pragma("unroll") for (int s = 0; s < 4; ++s) acc[tm][tn][s] =                     \
                    __builtin_amdgcn_mfma_f64_4x4x4f64(qf[j & 1][s], pf[st & 1][tm],               \
                                                       acc[tm][tn][s], 0, 0, 0);                   \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }

    const int blk = (lane >> 2) & 3;
    double *Ts = reinterpret_cast<double *>(smem); // the updated tile, column-major 64 x 64
    const int nslab = kb >> 6;
    for (int sl = 0; sl < nslab; ++sl) {
        double acc[2][2][4];
        const Tile444<2, 2> ct(X + (long)(64 * sl) * ldx, ldx, R0 + wr, wc, lane);
        const int nchunk = 4 * sl; // k = 64 sl
        if (sl > 0) {
            // this wave's staging source: my solved rows (P) or rows 64 sl .. of L11 (Q)
            const double *gsrc = stq ? gq0 + 64 * sl : gp;
            const long sld = stq ? ldl : ldx;
            BQ_TS_FILL(0, 0)
            ct.load_neg(acc);
            double pf[2][2], qf[2][4];
            for (int ch = 0; ch < nchunk; ch += 2) {
                BQ_TS_CHUNK(0, ch)
                BQ_TS_CHUNK(1, ch + 1)
            }
            __syncthreads(); // every wave is through with the staging buffers
        } else {
            ct.load_neg(acc);
        }
        // the tile -> LDS -> rows 16 wave .. + 15 in trsm_blk_kernel's operand form
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    Ts[wr + 16 * tm + l15 + 64 * (wc + 16 * tn + l4 + 4 * ((blk - s) & 3))] =
                        -acc[tm][tn][s];
        __syncthreads();
        const double *Lss = L11 + 64 * sl + (long)(64 * sl) * ldl + l15 + (long)l4 * ldl;
        const double *W = rec + (long)sl * BQ_DINV_HALF + 64 + l15 + 16 * l4;
        const double *Tw = Ts + 16 * wave + l15 + 64 * l4;
        double *Xr = X + R0 + 16 * wave + l15 + (long)(64 * sl + l4) * ldx;
        double4_t x[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double4_t a4 = {-Tw[64 * (16 * c)], -Tw[64 * (16 * c + 4)], -Tw[64 * (16 * c + 8)],
                            -Tw[64 * (16 * c + 12)]};
            double4_t xc = {0.0, 0.0, 0.0, 0.0};
            if (!(dbg & 1)) {
#pragma unroll
                for (int bb = 0; bb < c; ++bb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(
                            Lss[16 * c + (long)(16 * bb + 4 * r) * ldl], x[bb][r], a4, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xc = __builtin_amdgcn_mfma_f64_16x16x4f64(-W[256 * c + 64 * r], a4[r], xc, 0, 0, 0);
            } else {
                xc = -a4 * 1e-3;
            }
            x[c] = xc;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Xr[(long)(16 * c + 4 * r) * ldx] = xc[r];
        }
        // The solved slab is the next slabs' P operand.  Slab sl + 1 stages columns 16 ch .. of my
        // rows in chunk ch: columns of slab 0 from its first fill on, columns of THIS slab not
        // before chunk 4 sl >= 4 -- and every chunk's barrier drains vmcnt first.  So only slab
        // 0's stores have to be in memory here; the later slabs' land under the next fills.
        if (sl == 0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // Ts is free again
    }
#undef BQ_TS_FILL
#undef BQ_TS_OFF
#undef BQ_TS_READ_P
#undef BQ_TS_READ_Q
#undef BQ_TS_CHUNK
}

// ---------------------------------------------------------------------------
// Round 6: the same sweep with a TALL workgroup tile.  trsm_sweep_kernel's 64 x 64 tile stages
// 16 bytes of operand per 128 flops (8 flop / B into LDS) and runs at the rate of the 64-tile
// product (51.6 TFLOP/s with L2-hot operands and no solve, 38-43 as shipped); its 64-row
// workgroups live ~200 us, four to a CU, so a launch is 0.56-1.44 ROUNDS of the chip and the
// partial round is lost.  Here a workgroup owns R = 16 RT rows (64 ... 144), the slab tile is
// R x 64 and wave w holds ALL R rows of slab columns 16 w .. + 15: per k-step of 4 a wave reads
// RT fragments of P and the four rotated views of ONE Q fragment for 4 RT MFMAs (0.375 LDS reads
// per MFMA at RT = 8 against 0.625), the workgroup stages (R + 64) x 16 doubles per 2 R 64 16
// flops (10.7 flop / B at RT = 8), and the host picks RT per launch so that the grid is close to
// a whole number of rounds (launch_trsm_sweep).  The solve is trsm_sweep_kernel's: the tile goes
// through LDS into the 16-rows-per-wave operand form, wave w solves row groups w, w + 4, ... --
// interleaved, so two or three dependent MFMA chains run side by side --, and the NEXT slab's
// tile of A is requested before the solve starts and lands under it.  Same operations in the
// same order per element as trsm_sweep_kernel: the same bits.
// Staging image (per buffer): P as 16 k rows of R doubles (+ 128 B pad when RT is even, so that
// consecutive k rows start 32 banks apart), then Q as 16 k rows of 64 doubles + 128 B; the image
// is filled in 1 KiB pieces, one LDS-DMA wave instruction each, piece i by wave i % 4; the lane's
// source offset of every piece is computed once (a piece may straddle k rows; lanes that land
// in a pad fetch a valid dummy).  The tile buffer of the solve overlays the staging buffers.
// grid (8 nrb ceil(batch / 8)) cut by XCD as above, nrb = ceil(m / R); block 256.
// ---------------------------------------------------------------------------
template <int RT> struct SweepGeom {
    static constexpr int R = 16 * RT;
    static constexpr int PPAD = (RT % 2 == 0) ? 1 : 0;
    static constexpr int PGR = RT + PPAD;        // 128-byte groups per staged k row of P
    static constexpr int PPITCH = PGR * 128;     // bytes
    static constexpr int NP = 2 * PGR;           // 1 KiB pieces per P chunk (16 k rows)
    static constexpr int QPITCH = 5 * 128;
    static constexpr int NQ = 10;
    static constexpr int NI = NP + NQ;           // a multiple of 4 (PGR is odd)
    static constexpr int NJ = NI / 4;            // pieces per wave and chunk
    static constexpr int STAGE = NI * 1024;
    static constexpr int TP = R + 16 * PPAD;     // leading dimension of the solve's tile buffer
    static constexpr int TS = TP * 64 * 8;
    static constexpr int LDS = (2 * STAGE > TS) ? 2 * STAGE : TS;
    static constexpr int NG = (RT + 3) / 4;      // row groups per wave in the solve
};

template <int RT, int WPE>
__global__ __launch_bounds__(256, WPE) void trsm_sweep_tall_kernel(
    double *__restrict__ X, long ldx, long xstride, const double *__restrict__ L11, long ldl,
    long lstride, const double *__restrict__ rec, long rstride, int kb, int m, int nrb, int batch)
{
    using G = SweepGeom<RT>;
    static_assert(G::NI % 4 == 0, "pieces per chunk must split evenly over four waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int slot = blockIdx.x >> 3;
    const int b = (int)(blockIdx.x & 7) + 8 * (slot / nrb);
    if (b >= batch)
        return; // (the whole workgroup, before any barrier)
    const int rb = slot % nrb;
    X += (long)b * xstride;
    L11 += (long)b * lstride;
    rec += (long)b * rstride;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int R0 = rb * G::R;
    const int rv = min(G::R, m - R0); // rows of this block that exist (a multiple of 16)
    const int ngr = rv >> 4;
    const int l15 = lane & 15, l4 = lane >> 4, blk = (lane >> 2) & 3;

    // source offsets (bytes) of this wave's pieces: piece i = wave + 4 j covers granules
    // (16 B = two rows of one k row) 64 i .. 64 i + 63 of the image
    unsigned voff[G::NJ];
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
        const int i = wave + 4 * j;
        if (i < G::NP) {
            const int g = 64 * i + lane, k = g / (8 * G::PGR), pos = g % (8 * G::PGR);
            const int row = min(pos < 8 * RT ? 2 * pos : 0, rv - 2);
            voff[j] = (unsigned)(((long)k * ldx + row) * 8);
        } else {
            const int g = 64 * (i - G::NP) + lane, k = g / 40, pos = g % 40;
            const int row = pos < 32 ? 2 * pos : 0;
            voff[j] = (unsigned)(((long)k * ldl + row) * 8);
        }
    }
    const unsigned char *pview = smem + l4 * G::PPITCH + l15 * 8;
    const unsigned char *qview[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        qview[s] = smem + G::NP * 1024 + l4 * G::QPITCH + (16 * wave + ((l15 - 4 * s) & 15)) * 8;

#define BQ_TT_FILL(BUF_, CH_)                                                                      \
    {                                                                                              \
        const char *bp_ = reinterpret_cast<const char *>(Xp + (long)(CH_) * 16 * ldx);             \
        const char *bq_ = reinterpret_cast<const char *>(Lq + (long)(CH_) * 16 * ldl);             \
        _Pragma("unroll") for (int j = 0; j < G::NJ; ++j)                                          \
        {                                                                                          \
            const int i_ = wave + 4 * j;                                                           \
            const char *g_ = (i_ < G::NP ? bp_ : bq_) + voff[j];                                   \
            __builtin_amdgcn_global_load_lds((global_cvoid_t *)g_,                                 \
                                             (lds_void_t *)(smem + (BUF_) * G::STAGE + i_ * 1024), \
                                             16, 0, 0);                                            \
        }                                                                                          \
    }
#define BQ_TT_READ(BUF_, ST_, PF, QF)                                                              \
    {                                                                                              \
        _Pragma("unroll") for (int tm = 0; tm < RT; ++tm) PF[tm] =                                 \
            *reinterpret_cast<const double *>(pview + (BUF_) * G::STAGE + (ST_) * 4 * G::PPITCH +  \
                                              tm * 128);                                           \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) QF[s] = *reinterpret_cast<const double *>(   \
            qview[s] + (BUF_) * G::STAGE + (ST_) * 4 * G::QPITCH);                                 \
    }
    // (the wait is explicit: see trsm_sweep_kernel)
#define BQ_TT_CHUNK(BUF_, CH_)                                                                     \
    {                                                                                              \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
        __syncthreads();                                                                           \
        if ((CH_) + 1 < nchunk)                                                                    \
            BQ_TT_FILL(1 - (BUF_), (CH_) + 1)                                                      \
        BQ_TT_READ(BUF_, 0, pf[0], qf[0])                                                          \
        _Pragma("unroll") for (int st = 0; st < 4; ++st)                                           \
        {                                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            asm volatile("" ::"v"(qf[st & 1][0]), "v"(qf[st & 1][1]), "v"(qf[st & 1][2]),          \
                         "v"(qf[st & 1][3]));                                                      \
            _Pragma("unroll") for (int tm = 0; tm < RT; ++tm) asm volatile("" ::"v"(pf[st & 1][tm])); \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            if (st < 3)                                                                            \
                BQ_TT_READ(BUF_, st + 1, pf[(st + 1) & 1], qf[(st + 1) & 1])                       \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            _Pragma("unroll") for (int tm = 0; tm < RT; ++tm)                                      \
                _Pragma("unroll") for (int s = 0; s < 4; ++s) acc[tm][s] =                         \
                    __builtin_amdgcn_mfma_f64_4x4x4f64(qf[st & 1][s], pf[st & 1][tm], acc[tm][s],  \
                                                       0, 0, 0);                                   \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }

    // the slab's tile of A: rows R0 + 16 tm + l15 (clamped to existing rows), columns
    // 16 wave + 4 ((blk - s) & 3) + l4 of the slab -- the rotated-quad layout of Tile444
    unsigned coff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        coff[s] = (unsigned)((l15 + (long)(16 * wave + l4 + 4 * ((blk - s) & 3)) * ldx) * 8);
#define BQ_TT_LOADC(SL_)                                                                           \
    {                                                                                              \
        const char *c_ = reinterpret_cast<const char *>(X + R0 + (long)(64 * (SL_)) * ldx);        \
        _Pragma("unroll") for (int tm = 0; tm < RT; ++tm)                                          \
        {                                                                                          \
            const int tmc = min(tm, ngr - 1);                                                      \
            _Pragma("unroll") for (int s = 0; s < 4; ++s) acc[tm][s] =                             \
                *reinterpret_cast<const double *>(c_ + tmc * 128 + coff[s]);                       \
        }                                                                                          \
    }

    double acc[RT][4];
    double pf[2][RT], qf[2][4];
    double *Ts = reinterpret_cast<double *>(smem); // the updated tile, column-major TP x 64
    const double *Xp = X + R0;
    const int nslab = kb >> 6;
    BQ_TT_LOADC(0)
    for (int sl = 0; sl < nslab; ++sl) {
        const int nchunk = 4 * sl; // k = 64 sl
        const double *Lq = L11 + 64 * sl;
        if (sl > 0)
            BQ_TT_FILL(0, 0)
        // the diagonal block's fragments for the solve, requested NOW: read where they are used
        // (round 4's form) every stage of the solve waited a trip to L2 -- and, vmcnt counting
        // stores too, the previous stage's stores of X
        const double *Lss = L11 + 64 * sl + (long)(64 * sl) * ldl + l15 + (long)l4 * ldl;
        const double *W = rec + (long)sl * BQ_DINV_HALF + 64 + l15 + 16 * l4;
        double lfr[6][4], wfr[4][4];
#pragma unroll
        for (int c = 1; c < 4; ++c)
#pragma unroll
            for (int bb = 0; bb < c; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lfr[c * (c - 1) / 2 + bb][r] = Lss[16 * c + (long)(16 * bb + 4 * r) * ldl];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                wfr[c][r] = -W[256 * c + 64 * r];
        __builtin_amdgcn_sched_barrier(0);
        // (the tile was requested before the previous slab's solve; the first chunk's wait
        // covers it)
#pragma unroll
        for (int tm = 0; tm < RT; ++tm)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[tm][s] = -acc[tm][s];
        for (int ch = 0; ch < nchunk; ch += 2) {
            BQ_TT_CHUNK(0, ch)
            BQ_TT_CHUNK(1, ch + 1)
        }
        __syncthreads(); // every wave is through with the staging buffers
        // the tile -> LDS -> row groups in trsm_blk_kernel's operand form
#pragma unroll
        for (int tm = 0; tm < RT; ++tm)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                Ts[16 * tm + l15 + G::TP * (16 * wave + l4 + 4 * ((blk - s) & 3))] = -acc[tm][s];
        __builtin_amdgcn_sched_barrier(0);
        if (sl + 1 < nslab)
            BQ_TT_LOADC(sl + 1)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        double4_t x[G::NG][4], tn[G::NG];
        const double *Tw[G::NG];
#pragma unroll
        for (int gi = 0; gi < G::NG; ++gi) {
            const int g = min(wave + 4 * gi, RT - 1);
            Tw[gi] = Ts + 16 * g + l15 + G::TP * l4;
            tn[gi] = (double4_t){Tw[gi][0], Tw[gi][G::TP * 4], Tw[gi][G::TP * 8], Tw[gi][G::TP * 12]};
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double4_t a4[G::NG];
#pragma unroll
            for (int gi = 0; gi < G::NG; ++gi)
                a4[gi] = -tn[gi];
            if (c < 3) {
                // the next stage's tile columns, on their way while this stage's chain runs
#pragma unroll
                for (int gi = 0; gi < G::NG; ++gi)
                    tn[gi] = (double4_t){Tw[gi][G::TP * (16 * c + 16)], Tw[gi][G::TP * (16 * c + 20)],
                                         Tw[gi][G::TP * (16 * c + 24)], Tw[gi][G::TP * (16 * c + 28)]};
            }
#pragma unroll
            for (int bb = 0; bb < c; ++bb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int gi = 0; gi < G::NG; ++gi)
                        a4[gi] = __builtin_amdgcn_mfma_f64_16x16x4f64(lfr[c * (c - 1) / 2 + bb][r],
                                                                      x[gi][bb][r], a4[gi], 0, 0, 0);
                }
            double4_t xc[G::NG];
#pragma unroll
            for (int gi = 0; gi < G::NG; ++gi)
                xc[gi] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int gi = 0; gi < G::NG; ++gi)
                    xc[gi] = __builtin_amdgcn_mfma_f64_16x16x4f64(wfr[c][r], a4[gi][r], xc[gi], 0, 0, 0);
            }
#pragma unroll
            for (int gi = 0; gi < G::NG; ++gi) {
                x[gi][c] = xc[gi];
                const int g = wave + 4 * gi;
                if (g < ngr) {
                    double *Xr = X + R0 + 16 * g + l15 + (long)(64 * sl + l4) * ldx;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Xr[(long)(16 * c + 4 * r) * ldx] = xc[gi][r];
                }
            }
        }
        // The solved slab is the next slabs' P operand: see trsm_sweep_kernel (only slab 0's stores
        // have to be in memory before the next fill; the later slabs' land under four chunks'
        // waits)
        if (sl == 0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // Ts is free again
    }
#undef BQ_TT_FILL
#undef BQ_TT_READ
#undef BQ_TT_CHUNK
#undef BQ_TT_LOADC
}

// ---------------------------------------------------------------------------
// Round 6, second form: RIGHT-looking, the panel rows' tiles resident in registers.  The two
// left-looking kernels above re-read a workgroup's own solved rows (its P operand) from memory
// for every later slab: 10.5 KiB per row per launch, 1.4 GB for 64 panels of 2048 rows, which
// no other workgroup shares and which therefore comes from the Infinity Cache or HBM at
// 2.6 TB/s.  The launch is bound by the socket's power cap (1.38 kW at 2.1 GHz; the update kernel
// holds 2.26 GHz at the same MFMA occupancy), and that traffic is what the sweep pays beyond the
// update.  Here a workgroup owns 64 rows and holds the tiles of ALL slabs in accumulators (wave w:
// the 64 rows x columns 16 w .. + 15 of every slab, 16 doubles per slab, <= 6 live slabs = 192
// VGPRs).  Step s: the slab's tile goes through LDS into the 16-rows-per-wave form and is solved
// (trsm_blk_kernel's scheme), the solved slab stays IN LDS as the P operand and every later
// slab t takes  acc_t += X_s L_ts^T  with L_ts (64 x 64) streamed through a double-buffered
// stage of half blocks (32 k columns, one barrier per 128 MFMAs of a wave); the first half block
// of a step is requested before the previous step ends.  A is read once, X written once, nothing
// is read back; 16 flop per staged byte, the 128 x 128 update's figure.
// The tile buffer is swizzled (row ^ 16 for odd k) so that the P views and the solve's reads are
// free of bank conflicts without a pad.  NS = kb / 64 <= 7.  grid cut by XCD as above.
// ---------------------------------------------------------------------------
#define BQ_RL_QPITCH 640
#define BQ_RL_QSTAGE (20 * 1024)
#define BQ_RL_TS (64 * 64 * 8)
#define BQ_RL_LDS (BQ_RL_TS + 2 * BQ_RL_QSTAGE)

template <int NS>
__global__ __launch_bounds__(256, 2) void trsm_sweep_rl_kernel(
    double *__restrict__ X, long ldx, long xstride, const double *__restrict__ L11, long ldl,
    long lstride, const double *__restrict__ rec, long rstride, int nrb, int batch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int slot = blockIdx.x >> 3;
    const int b = (int)(blockIdx.x & 7) + 8 * (slot / nrb);
    if (b >= batch)
        return; // (the whole workgroup, before any barrier)
    const int rb = slot % nrb;
    X += (long)b * xstride + 64 * rb;
    L11 += (long)b * lstride;
    rec += (long)b * rstride;
    const int t_ = threadIdx.x, lane = t_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t_ >> 6);
    const int l15 = lane & 15, l4 = lane >> 4, blk = (lane >> 2) & 3;
    const int sw = (l4 & 1) << 4;
    double *Ts = reinterpret_cast<double *>(smem);
    unsigned char *qst = smem + BQ_RL_TS;

    // a half block of L_ts: image [32 k][64 rows + 128 B], 20 pieces of 1 KiB, piece i by wave
    // i % 4; source offsets per lane, once
    unsigned voff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int g = 64 * (wave + 4 * j) + lane, k = g / 40, pos = g % 40;
        voff[j] = (unsigned)(((long)k * ldl + (pos < 32 ? 2 * pos : 0)) * 8);
    }
#define BQ_RL_FILL(H_, T_, S_)                                                                     \
    {                                                                                              \
        const char *b_ = reinterpret_cast<const char *>(L11 + 64 * (T_) +                          \
                                                        (long)(64 * (S_) + 32 * (H_)) * ldl);      \
        _Pragma("unroll") for (int j = 0; j < 5; ++j) __builtin_amdgcn_global_load_lds(            \
            (global_cvoid_t *)(b_ + voff[j]),                                                      \
            (lds_void_t *)(qst + (H_) * BQ_RL_QSTAGE + (wave + 4 * j) * 1024), 16, 0, 0);          \
    }
    // P views of the tile buffer: element (row 16 tm + l15, k) at Ts[((16 tm + l15) ^ sw) + 64 k],
    // k = 32 h + 4 kk + l4 -- even tm from pe, odd tm from po
    const unsigned char *pe = smem + ((l15 + sw) + 64 * l4) * 8;
    const unsigned char *po = smem + ((l15 - sw) + 64 * l4) * 8;
    const unsigned char *qview[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        qview[s] = qst + l4 * BQ_RL_QPITCH + (16 * wave + ((l15 - 4 * s) & 15)) * 8;
#define BQ_RL_READ(H_, KK_, PF, QF)                                                                \
    {                                                                                              \
        _Pragma("unroll") for (int tm = 0; tm < 4; ++tm) PF[tm] = *reinterpret_cast<const double *>( \
            ((tm & 1) ? po : pe) + (16 * tm + 64 * (32 * (H_) + 4 * (KK_))) * 8);                  \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) QF[s] = *reinterpret_cast<const double *>(   \
            qview[s] + (H_) * BQ_RL_QSTAGE + 4 * (KK_) * BQ_RL_QPITCH);                            \
    }
    // one half block: (1) its fill has landed, every wave is through with the other buffer;
    // (2) the next half block's fill; (3) 8 k-steps of 16 MFMAs, fragments one k-step ahead
#define BQ_RL_UNIT(H_, T_, NEXT_)                                                                  \
    {                                                                                              \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
        __syncthreads();                                                                           \
        NEXT_                                                                                      \
        BQ_RL_READ(H_, 0, pf[0], qf[0])                                                            \
        _Pragma("unroll") for (int kk = 0; kk < 8; ++kk)                                           \
        {                                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            asm volatile("" ::"v"(qf[kk & 1][0]), "v"(qf[kk & 1][1]), "v"(qf[kk & 1][2]),          \
                         "v"(qf[kk & 1][3]), "v"(pf[kk & 1][0]), "v"(pf[kk & 1][1]),               \
                         "v"(pf[kk & 1][2]), "v"(pf[kk & 1][3]));                                  \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            if (kk < 7)                                                                            \
                BQ_RL_READ(H_, kk + 1, pf[(kk + 1) & 1], qf[(kk + 1) & 1])                         \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            _Pragma("unroll") for (int tm = 0; tm < 4; ++tm)                                       \
                _Pragma("unroll") for (int s = 0; s < 4; ++s) acc[T_][tm][s] =                     \
                    __builtin_amdgcn_mfma_f64_4x4x4f64(qf[kk & 1][s], pf[kk & 1][tm],              \
                                                       acc[T_][tm][s], 0, 0, 0);                   \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }

    // every slab's tile of A, negated: rows 16 tm + l15, columns 64 t + 16 wave +
    // 4 ((blk - s) & 3) + l4 -- Tile444's rotated-quad layout.  Slab 0's tile goes straight into
    // the tile buffer: acc[t] is slab t + 1 (a slab that has been solved still occupies its
    // registers -- the step loop is not unrolled --, so the first one should not)
    double acc[NS > 1 ? NS - 1 : 1][4][4];
    {
        unsigned coff[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
            coff[s] = (unsigned)((l15 + (long)(16 * wave + l4 + 4 * ((blk - s) & 3)) * ldx) * 8);
        double a0[4][4];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                a0[tm][s] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(X) +
                                                              tm * 128 + coff[s]);
#pragma unroll
        for (int t = 1; t < NS; ++t) {
            const char *c_ = reinterpret_cast<const char *>(X + (long)(64 * t) * ldx);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[t - 1][tm][s] = *reinterpret_cast<const double *>(c_ + tm * 128 + coff[s]);
        }
        if (NS > 1)
            BQ_RL_FILL(0, 1, 0)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                Ts[((16 * tm + l15) ^ sw) + 64 * (16 * wave + l4 + 4 * ((blk - s) & 3))] = a0[tm][s];
    }
#pragma unroll
    for (int t = 1; t < NS; ++t)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[t - 1][tm][s] = -acc[t - 1][tm][s];

    double pf[2][4], qf[2][4];
    for (int sl = 0; sl < NS; ++sl) {
        // slab sl's tile -> LDS (the previous step's P views are read out: the barrier that ends
        // every step)
#pragma unroll
        for (int t = 1; t < NS; ++t)
            if (t == sl) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        Ts[((16 * tm + l15) ^ sw) + 64 * (16 * wave + l4 + 4 * ((blk - s) & 3))] =
                            -acc[t - 1][tm][s];
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // the solve: wave w, rows 16 w .. + 15; X to memory and, in place, back into the tile
        // buffer -- the later slabs' P operand
        {
            const double *Lss = L11 + 64 * sl + (long)(64 * sl) * ldl + l15 + (long)l4 * ldl;
            const double *W = rec + (long)sl * BQ_DINV_HALF + 64 + l15 + 16 * l4;
            double *Tw = Ts + ((16 * wave + l15) ^ sw) + 64 * l4;
            double *Xr = X + 16 * wave + l15 + (long)(64 * sl + l4) * ldx;
            double4_t x[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double4_t a4 = {-Tw[64 * (16 * c)], -Tw[64 * (16 * c + 4)], -Tw[64 * (16 * c + 8)],
                                -Tw[64 * (16 * c + 12)]};
#pragma unroll
                for (int bb = 0; bb < c; ++bb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(
                            Lss[16 * c + (long)(16 * bb + 4 * r) * ldl], x[bb][r], a4, 0, 0, 0);
                double4_t xc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xc = __builtin_amdgcn_mfma_f64_16x16x4f64(-W[256 * c + 64 * r], a4[r], xc, 0, 0, 0);
                x[c] = xc;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Xr[(long)(16 * c + 4 * r) * ldx] = xc[r];
                    Tw[64 * (16 * c + 4 * r)] = xc[r];
                }
            }
        }
        // the later slabs' updates (unit (t, h) reads buffer h; its barrier also orders the solve's
        // LDS stores before the P views)
#pragma unroll
        for (int t = 1; t < NS; ++t)
            if (t > sl) {
                BQ_RL_UNIT(0, t - 1, BQ_RL_FILL(1, t, sl))
                BQ_RL_UNIT(1, t - 1, if (t + 1 < NS) { BQ_RL_FILL(0, t + 1, sl) } else if (sl + 2 < NS) {
                    BQ_RL_FILL(0, sl + 2, sl + 1)
                })
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // every wave is through with the P views
    }
#undef BQ_RL_FILL
#undef BQ_RL_READ
#undef BQ_RL_UNIT
}
