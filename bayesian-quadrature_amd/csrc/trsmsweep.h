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

