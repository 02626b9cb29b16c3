// potrf.hip -- the launch sequences of the blocked right-looking Cholesky: block sizes,
// recursive panels, the one-launch sweeps of small systems, the two-stream look-ahead and the
// two half-batches of mid-sized batches.  No kernels of its own (k_panel.hip, k_gemm.hip).
#include "host.h"

namespace bqh {

// A batch's diagonal factors (enqueue_potrf_dfirst): one workgroup per matrix from 96 matrices on
// -- the workgroups then cover the chip, and unlike the one-launch steps they solve nothing
// twice --, the one-launch steps below (64 matrices would leave three quarters of the CUs without
// a factor to work on: 366 us alone against 236).  BQ_DF_WG = 0 / 1 forces either.
static bool dfirst_wg(const bq_ctx *c, int batch) { return c->df_wg < 0 ? batch >= 96 : c->df_wg != 0; }

int auto_nb(const bq_ctx *c, int ntot, int batch)
{
    if (c->nb_override > 0)
        return c->nb_override;
    // the trailing update re-reads and re-writes the whole remaining matrix once per
    // outer block: when the batch's matrices do not fit the caches the outer block
    // must be wide (256: 46 GB instead of 183 GB of traffic at N=16384), when they do
    // a narrow block means fewer, shorter launches
    const double mb = 8.0 * (double)ntot * ntot * batch / 1e6;
    if (batch <= 2) {
        // One or two matrices cannot fill the chip with a 64-column panel: the sweep is a
        // chain of dependent launches and the one-launch step of outer block 64 (slab.h) is
        // the shortest chain until the k = 64 updates cost more than it saves.  Re-measured in
        // round 3 with the 64-tile trailing updates (tools/potrf_sizes.py on one matrix, ms with
        // blocks 64 / 256: N=2560 0.761 / 0.774, 3072 1.013 / 0.995, 3584 1.346 / 1.318, 4096
        // 1.776 / 1.617, 4608 2.311 / 1.967): from 3072 rows the first panels are blocked (with
        // the look-ahead once there is room for it) and the LAST rows go to the one-launch steps
        // (slab_max / la_min, enqueue_potrf_group)
        if (ntot < c->slab_nb_max)
            return 64;
        // a wider block halves the trailing update's C traffic per flop (60 instead of 56
        // TFLOP/s at k = 512); it pays once the panel it lengthens hides behind the bulk
        // update (round 3, ms with 256 / 320 / 384 / 512: N = 5120 2.354 / 2.330 / 2.289 / 2.378,
        // 8192 5.432 / 5.357 / 5.306 / 5.416, 10240 8.687 / 8.552 / 8.441 / 8.422, 12288 13.29 /
        // 12.92 / 12.79 / 12.71, 16384 27.36 / 26.72 / 26.19 / 26.00)
        return ntot < 9216 ? 384 : 512;
    }
    // A few small matrices (the stacked parameter sets of the hyper-parameter loops: 5-6 systems
    // of the sample count) are a chain of dependent launches like one matrix is, and the
    // one-launch steps stay the shortest chain while a step's workgroups -- batch x T (T + 1) / 2
    // tiles, each with its redundant panel solves -- fit the chip about five times over
    // (tools/nb_sweep.py, ms with blocks 64 / 128 / 256: 5 x N=1024 0.377 / 0.475 / 0.512,
    // 12 x 1024 0.564 / 0.611 / 0.664, 16 x 1024 0.667 / 0.648 / 0.697, 8 x 1536 0.940 / 0.953 /
    // 0.977, 5 x 2048 1.237 / 1.138 / 1.185; the hyper-parameter objective + gradient at 1024
    // samples 0.99 -> 0.78 ms)
    const long T = ntot / 64;
    if ((long)batch * T * (T + 1) / 2 <= 2500)
        return 64;
    // (batches of mid-sized matrices, recursive panels: C5 shard 6.55 / 6.42 / 6.70 / 6.57 ms
    // with 256 / 320 / 384 / 512; 256 x C2 5.72 / 5.47 / 5.64 ms with 256 / 320 / 448; below
    // half a gigabyte 128 is still ahead: 12 x N=2048 1.694 / 1.716, 16 x 2048 2.024 / 1.954,
    // 5 x 3072 2.250 / 2.272, 8 x 3072 3.069 / 2.856 with 128 / 256)
    // (re-measured at the end of round 3, 320 against 448 in one box: C3 190.4 / 187.0 ms,
    // 256 x C2 5.23 / 5.16, C5 shard 5.82 / 5.80)
    // (round 4, diagonal block first, ms with 320 / 384 / 448 / 512 in one box: C5 shard 5.85 / 5.78 /
    // 5.71 / 5.90 on the one-launch steps; with a workgroup per matrix 256 x C2 4.45 / 4.32 / 4.50 /
    // 4.52, the C3 grid 188 / 178 / 190 / 180)
    if (ntot >= 1024 && mb >= 500.0)
        return (c->diag_first && dfirst_wg(c, batch)) ? 384 : 448;
    if (ntot >= 512)
        return 128;
    return 64;
}

// Columns [j0, j0 + w) of a panel, recursively: the left half, ONE update of the right half's
// columns with the whole left half, the right half.  Same flops as the left-looking slab
// order (each 64-column slab updated with everything before it, n = 64 per launch), but two
// thirds of a 256-wide panel's update flops are then ONE n = 128, k = 128 product -- wide
// enough for the LDS-staged kernel -- instead of two n = 64 launches that re-stream the panel
// (a batch's panel does not fit in L2: the n = 64 updates ran at 12 TFLOP/s; C5 shard
// 6.8 -> see DESIGN).  A 64-column slab: its diagonal factor (unless the launch that last
// updated it carried it: diag_done) and the solve of the rows below.
static int enqueue_panel_rec(bq_ctx *c, double *A, long lda, long astride, int batch, int ntot, int j0,
                      int w, double *dinv, int *info, bool diag_done)
{
    if (w <= 64) {
        double *Ajj = A + j0 + (long)j0 * lda;
        if (!diag_done)
            BQCHK(launch_potf2(c, A, lda, astride, j0, dinv, BQ_DINV_STRIDE, info, batch));
        return launch_trsm_blk(c, Ajj + 64, lda, astride, ntot - j0 - 64, Ajj, lda, astride, dinv,
                               BQ_DINV_STRIDE, batch);
    }
    const int wl = ((w / 64 + 1) / 2) * 64, wr = w - wl;
    BQCHK(enqueue_panel_rec(c, A, lda, astride, batch, ntot, j0, wl, dinv, info, diag_done));
    const int r0 = j0 + wl;
    const double *P = A + r0 + (long)j0 * lda;
    const int fj = gemm_uses_lds(c, ntot - r0, wr, wl, 1, batch) ? -1 : r0;
    BQCHK(launch_gemm(c, BQ_K_GEMM, A + r0 + (long)r0 * lda, lda, astride, P, lda, astride, P, 1,
                      lda, astride, ntot - r0, wr, wl, 1, batch, fj, dinv, BQ_DINV_STRIDE, info));
    return enqueue_panel_rec(c, A, lda, astride, batch, ntot, r0, wr, dinv, info, fj >= 0);
}

// the 64-column slabs of one outer block [K0, K0+KB): left-looking update, diagonal
// factor, panel solve -- enqueued on c->cur.  With fusion on, the diagonal factor of a
// slab rides in the launch that last updates it (the slab update here, or the
// trailing update of the previous block: `diag_done`).
// With a scratch column pair `ws` (panel_ws_doubles) every step is ONE launch
// (panel_step_kernel, slab.h): the workgroups solve the rows they need themselves.
static int enqueue_panel(bq_ctx *c, double *A, long lda, long astride, int batch, int ntot, int K0,
                  int KB, double *dinv, int *info, bool diag_done, double *ws = nullptr)
{
    // (a batch fills the chip without this: the redundant solves then only cost throughput)
    if (ws && batch <= 2) {
        if (!diag_done)
            BQCHK(launch_potf2(c, A, lda, astride, K0, dinv, BQ_DINV_STRIDE, info, batch));
        if (ntot - K0 - 64 <= 0)
            return BQ_OK;
        const long sstride = 64L * ntot;
        double *S[2] = {ws, ws + sstride * batch};
        double *SL = ws + 2 * sstride * batch; // 64 x 64 per problem
        const int ns = KB / 64;
        for (int sidx = 0; sidx < ns; ++sidx) {
            const int j0 = K0 + 64 * sidx;
            const int nrb = (ntot - j0 - 64) / 64;
            if (nrb <= 0)
                break;
            const int has_next = sidx + 1 < ns;
            const double m = 64.0 * nrb;
            const int par = sidx & 1;
            BQCHK(launch_panel_step(
                c, A, lda, astride, batch, nrb, S[par], S[par ^ 1], (long)ntot, sstride, K0, j0,
                dinv + par * BQ_DINV_HALF, dinv + (par ^ 1) * BQ_DINV_HALF, has_next, sidx == 0, SL,
                info,
                (m * 64.0 * 64.0 + (has_next ? 2.0 * m * 64.0 * 64.0 * (sidx + 1) : 0.0)) * batch));
        }
        return BQ_OK;
    }
    return enqueue_panel_rec(c, A, lda, astride, batch, ntot, K0, KB, dinv, info, diag_done);
}

// Eliminate the first ncols columns (multiple of 64) of the ntot x ntot lower
// matrix (ntot multiple of 64), batched.  dinv: BQ_DINV_STRIDE doubles per problem.
//
// With more than one outer block and a wide block the factorisation runs with a
// look-ahead of one panel on two streams.  The main stream carries only the bulk
// trailing updates (everything right of the next panel), back to back; the
// high-priority aux stream updates the next panel's columns and factors that
// panel meanwhile.  The two meet through events: update k+1 waits for panel k+1,
// the panel-column update k+1 waits for trailing update k (which last wrote
// those columns).
// doubles of scratch the one-launch slab sweep needs (two panel columns per problem)
size_t panel_ws_doubles(int ntot, int batch) { return ((size_t)2 * 64 * ntot + 4096) * batch; }

// whether a sweep over `batch` matrices of size ntot can use that scratch at all (with the
// block size in force now): callers that own long-lived workspaces skip the allocation else
bool panel_ws_useful(const bq_ctx *c, int ntot, int batch)
{
    return batch <= 2 || auto_nb(c, ntot, batch) == 64;
}

// Outer block 64 (small systems): one launch per 64-column step (slab.h) after the first
// diagonal factor and the staging of panel 0.
// col0: global column of A's first column (a sweep over the trailing block of a larger
// factorisation reports failures in the larger matrix's numbering)
// rstride > 0: `dinv` is a row of RECORDS per problem (rstride doubles apart), one BQ_DINV_HALF
// per 64-column step, and every step's reciprocal pivots / block inverses stay behind in the
// record of its own diagonal block (the batched panel solve reads them: enqueue_potrf_dfirst);
// else the two halves of a BQ_DINV_STRIDE ping-pong.
static int enqueue_slab_sweep(bq_ctx *c, double *A, long lda, long astride, int batch, int ntot,
                       int ncols, double *dinv, int *info, double *ws, int col0 = 0,
                       bool first_done = false, long rstride = 0)
{
    const long dstride = rstride > 0 ? rstride : (long)BQ_DINV_STRIDE;
    if (ntot <= 64)
        return launch_potf2(c, A - col0 - (long)col0 * lda, lda, astride, col0, dinv, dstride,
                            info, batch);
    const long sstride = 64L * ntot;
    double *S[2] = {ws, ws + sstride * batch};
    if (!first_done)
        // the first diagonal factor and the staging of panel 0 share a launch (or ride in the
        // assembly: assemble_first_kernel)
        BQCHK(launch_slab_first(c, A, lda, astride, batch, S[0], (long)ntot, sstride, ntot, dinv,
                                info, col0, dstride));
    for (int j0 = 0, par = 0; j0 < ncols; j0 += 64, par ^= 1) {
        const int r0 = j0 + 64;
        if (r0 >= ntot)
            break;
        const int fnext = r0 < ncols;
        const double m = (double)(ntot - r0);
        // only batch element 0 is stamped, and only while the probe's buffer has room
        long long *stamps = (c->stamp_buf && j0 / 64 < c->stamp_steps)
                                ? c->stamp_buf + 160 * (j0 / 64)
                                : nullptr;
        // the tile updates (lower half of 2 m^2 64) and the solve of the panel (m 64^2)
        const long din = rstride > 0 ? (long)(j0 / 64) * BQ_DINV_HALF : (long)par * BQ_DINV_HALF;
        const long dout =
            rstride > 0 ? (long)(j0 / 64 + 1) * BQ_DINV_HALF : (long)(par ^ 1) * BQ_DINV_HALF;
        BQCHK(launch_slab_step(c, A, lda, astride, batch, S[par], S[par ^ 1], (long)ntot, sstride,
                               ntot, j0, dinv + din, dinv + dout, fnext, !fnext, info, col0, stamps,
                               (m * m * 64.0 + m * 64.0 * 64.0) * batch, dstride));
    }
    return BQ_OK;
}

// the look-ahead pays up to this many matrices per batch (enqueue_potrf_partial)
#define BQ_LA_MAX_BATCH 48

// (tile choice while two streams share the chip: gemm_lds_tile)
struct Sharing {
    bq_ctx *c;
    int prev;
    Sharing(bq_ctx *c_, int how) : c(c_), prev(c_->sharing) { c->sharing = how; }
    ~Sharing() { c->sharing = prev; }
};

// nb_forced: the outer block of the whole batch when this call factors one half of it
// whether a factorisation of these sizes goes to the one-launch slab sweep from its first column
bool sweep_is_slab(const bq_ctx *c, int ntot, int ncols, int batch, size_t panel_ws_len)
{
    return auto_nb(c, ntot, batch) == 64 && panel_ws_len >= panel_ws_doubles(ntot, batch) &&
           ncols >= 64 && ntot > 64;
}

// skip_border: the caller reads its results off the border ROWS (plan_readout_kernel), so the
// trailing updates leave the border x border block alone
static int enqueue_potrf_group(bq_ctx *c, double *A, long lda, long astride, int batch, int ntot,
                        int ncols, double *dinv, int *info, double *panel_ws, size_t panel_ws_len,
                        int nb_forced, bool first_done = false, bool skip_border = false)
{
    const int NB = nb_forced > 0 ? nb_forced : auto_nb(c, ntot, batch);
    double *ws = (panel_ws && panel_ws_len >= panel_ws_doubles(ntot, batch)) ? panel_ws : nullptr;
    if (NB == 64 && ws && ncols >= 64)
        return enqueue_slab_sweep(c, A, lda, astride, batch, ntot, ncols, dinv, info, ws, 0,
                                  first_done);
    const bool la = c->lookahead && c->aux && NB >= 128 && ncols > NB && batch <= BQ_LA_MAX_BATCH;
    int K0 = 0;
    bool panel_done = false; // panel K0 was already factored by the look-ahead phase
    int st = BQ_OK;
    if (la && ntot - std::min(NB, ncols) >= c->la_min) {
        // fork: the aux stream starts after everything already queued on the main stream
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_fork, 0));
        // aux stream: panel 0
        c->cur = c->aux;
        st = enqueue_panel(c, A, lda, astride, batch, ntot, 0, std::min(NB, ncols), dinv, info,
                           false, ws);
        c->cur = c->stream;
        if (st != BQ_OK)
            return st;
        HIPCHK(c, hipEventRecord(c->ev_panel, c->aux));
        bool have_b = false; // a trailing update is in flight on the main stream
        Sharing la_scope(c, 1);
        for (; K0 < ncols && st == BQ_OK; K0 += NB) {
            const int KB = std::min(NB, ncols - K0);
            const int r0 = K0 + KB;
            // main stream: wait for panel K0
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_panel, 0));
            panel_done = true;
            if (r0 >= ntot) {
                K0 = ncols;
                break;
            }
            // Once the bulk update is shorter than the panel chain it has to hide, the two
            // streams only slow each other down (N = 4096: 2.50 ms with, 2.26 ms without):
            // the rest of the sweep runs sequentially on the main stream.
            if (ntot - r0 < c->la_min)
                break;
            const double *P = A + r0 + (long)K0 * lda;
            const int nw = (r0 < ncols) ? std::min(NB, ncols - r0) : 0; // width of the next panel
            if (nw > 0) {
                // aux stream: bring the next panel's columns up to date (they were last
                // written by the previous trailing update on the main stream), factor it
                if (have_b)
                    HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_next, 0));
                c->cur = c->aux;
                // (a wide panel's update is worth the LDS-staged kernel, which carries no
                // fused diagonal factor: enqueue_panel then factors the block itself)
                const int fj =
                    !gemm_uses_lds(c, ntot - r0, nw, KB, 1, batch) ? r0 : -1;
                st = launch_gemm(c, BQ_K_SYRK, A + r0 + (long)r0 * lda, lda, astride, P, lda,
                                 astride, P, 1, lda, astride, ntot - r0, nw, KB, 1, batch, fj, dinv,
                                 BQ_DINV_STRIDE, info);
                if (st == BQ_OK)
                    st = enqueue_panel(c, A, lda, astride, batch, ntot, r0, nw, dinv, info,
                                       fj >= 0, ws);
                c->cur = c->stream;
                if (st != BQ_OK)
                    break;
                HIPCHK(c, hipEventRecord(c->ev_panel, c->aux));
                // main stream: everything right of the next panel, concurrently
                const int r1 = r0 + nw;
                if (r1 < ntot) {
                    const double *P1 = A + r1 + (long)K0 * lda;
                    st = launch_gemm(c, BQ_K_SYRK, A + r1 + (long)r1 * lda, lda, astride, P1, lda,
                                     astride, P1, 1, lda, astride, ntot - r1, ntot - r1, KB, 1,
                                     batch, -1, nullptr, 0, nullptr,
                                     skip_border ? ncols - r1 : 0);
                    HIPCHK(c, hipEventRecord(c->ev_next, c->stream));
                    have_b = true;
                }
            } else {
                // no further panel: the remaining trailing block is pure Schur complement
                st = launch_gemm(c, BQ_K_SYRK, A + r0 + (long)r0 * lda, lda, astride, P, lda,
                                 astride, P, 1, lda, astride, ntot - r0, ntot - r0, KB, 1, batch);
            }
            panel_done = false;
        }
        c->cur = c->stream;
        if (st != BQ_OK)
            return st;
    }
    // sequential sweep: everything without look-ahead, otherwise the rest
    bool diag_done = false;
    for (; K0 < ncols; K0 += NB) {
        const int KB = std::min(NB, ncols - K0);
        if (!panel_done)
            BQCHK(enqueue_panel(c, A, lda, astride, batch, ntot, K0, KB, dinv, info, diag_done,
                                ws));
        panel_done = false;
        const int r0 = K0 + KB;
        diag_done = false;
        // The last rows of a large matrix are a small factorisation of their own -- the
        // Schur complement once this block's update is in --, and for one or two matrices
        // the one-launch steps are its shortest chain: hand the rest to the slab sweep.
        const bool to_slab = batch <= 2 && ws && NB > 64 && r0 < ncols && ntot - r0 < c->slab_max;
        if (r0 < ntot) {
            const double *P = A + r0 + (long)K0 * lda;
            // the trailing update also factors the next diagonal block if there is one
            const int fj = (r0 < ncols && !to_slab &&
                            !gemm_uses_lds(c, ntot - r0, ntot - r0, KB, 1, batch))
                               ? r0
                               : -1;
            BQCHK(launch_gemm(c, BQ_K_SYRK, A + r0 + (long)r0 * lda, lda, astride, P, lda, astride,
                              P, 1, lda, astride, ntot - r0, ntot - r0, KB, 1, batch, fj, dinv,
                              BQ_DINV_STRIDE, info, skip_border ? ncols - r0 : 0));
            diag_done = fj >= 0;
        }
        if (to_slab)
            return enqueue_slab_sweep(c, A + r0 + (long)r0 * lda, lda, astride, batch, ntot - r0,
                                      ncols - r0, dinv, info, ws, r0);
    }
    return BQ_OK;
}

// ---------------------------------------------------------------------------------------------
// Batches, diagonal block first (round 4).  The recursive panels above factor an outer block's
// m x NB panel as a chain of 64-column steps over ALL its rows: per step a one-workgroup diagonal
// factor, a panel solve and a product, each a pass over rows that do not fit a cache (a C5
// shard's panels: 440 MB) -- a third of a batch's sweep at a fifth of the chip's rate, and
// a chain that two half-batches on two streams only hide by slowing each other down.  But
// the rows below an outer block's NB x NB diagonal block take no part in factoring it:
//     D  the diagonal block alone: a small Cholesky (one-launch steps, slab.h), a few MB per
//        matrix, whose 64-column steps leave their block inverses behind in records;
//     S  L21 = A21 L11^-T for all rows below in ONE sweep of throughput launches, recursively
//        (left half, product, right half) with the solve of a product's first 64 columns fused
//        into the product (gemm_trsm64_kernel): NB / 64 launches, none latency-bound;
//     U  the trailing update, k = NB -- first, on the second stream, the NEXT diagonal block
//        and its factorisation D, hidden beside the rest of the update.
// The whole batch moves in lock-step through launches that each fill the chip.
// ---------------------------------------------------------------------------------------------
static size_t dfirst_rec_doubles(int nb, int batch) { return (size_t)(nb / 64) * BQ_DINV_HALF * batch; }
// (two sets of records: with the early fork the next block's diagonal factor writes its own while
// the rest of this block's panel solve still reads this block's)
static size_t dfirst_ws_doubles(int nb, int batch)
{
    return panel_ws_doubles(nb, batch) + 2 * dfirst_rec_doubles(nb, batch);
}

static bool dfirst_applies(const bq_ctx *c, int ntot, int ncols, int batch)
{
    return c->diag_first && batch >= 3 && ncols >= 128 && auto_nb(c, ntot, batch) >= 128;
}

size_t sweep_ws_doubles(const bq_ctx *c, int ntot, int batch)
{
    if (dfirst_applies(c, ntot, ntot, batch))
        return dfirst_ws_doubles(std::min(auto_nb(c, ntot, batch), ntot), batch);
    return panel_ws_useful(c, ntot, batch) ? panel_ws_doubles(ntot, batch) : 0;
}

// X = rows [rx, rx + m) of the panel that starts at column K0: columns [j0, j0 + w) (relative
// to K0) solved against L11 = A[K0.., K0..]; rec: the records of L11's 64 x 64 diagonal blocks.
// solved: the launch that last updated slab j0 solved it as well.
static int enqueue_trsm_rec(bq_ctx *c, double *A, long lda, long astride, int batch, int rx, int m,
                            int K0, int j0, int w, const double *rec, long rstride, bool solved)
{
    if (w <= 64) {
        if (solved)
            return BQ_OK;
        const long cj = K0 + j0;
        return launch_trsm_blk(c, A + rx + cj * lda, lda, astride, m, A + cj + cj * lda, lda,
                               astride, rec + (long)(j0 / 64) * BQ_DINV_HALF, rstride, batch);
    }
    const int wl = ((w / 64 + 1) / 2) * 64, wr = w - wl;
    BQCHK(enqueue_trsm_rec(c, A, lda, astride, batch, rx, m, K0, j0, wl, rec, rstride, solved));
    const long cl = K0 + j0, cr = cl + wl;
    double *C = A + rx + cr * lda;
    const double *P = A + rx + cl * lda;
    const double *Q = A + cr + cl * lda; // L11[j0 + wl .., j0 ..): wr x wl
    const bool fuse = gemm_trsm_ok(c, m, wr, wl);
    if (fuse)
        BQCHK(launch_gemm_trsm(c, C, lda, astride, P, lda, astride, Q, lda, astride, m, wr, wl,
                               A + cr + cr * lda, lda, astride,
                               rec + (long)((j0 + wl) / 64) * BQ_DINV_HALF, rstride, batch));
    else
        BQCHK(launch_gemm(c, BQ_K_GEMM, C, lda, astride, P, lda, astride, Q, 1, lda, astride, m, wr,
                          wl, 0, batch));
    return enqueue_trsm_rec(c, A, lda, astride, batch, rx, m, K0, j0 + wl, wr, rec, rstride, fuse);
}

// the panel solve of one outer block as enqueue_potrf_dfirst issues it (also bq_probe_panel_solve)
int enqueue_panel_solve(bq_ctx *c, double *A, long lda, long astride, int batch, int r0, int m2,
                        int K0, int KB, const double *rec, long rstride)
{
    if (c->df_sweep && c->gemm_lds64)
        return launch_trsm_sweep(c, A + r0 + (long)K0 * lda, lda, astride, m2,
                                 A + K0 + (long)K0 * lda, lda, astride, rec, rstride, KB, batch);
    return enqueue_trsm_rec(c, A, lda, astride, batch, r0, m2, K0, 0, KB, rec, rstride, false);
}

// (the condition under which enqueue_potrf_partial takes enqueue_potrf_dfirst, and the sweep has at
// least two outer blocks: with one there is no product that could seed what lies right of it)
int dfirst_seed_cols(const bq_ctx *c, int ntot, int ncols, int batch, size_t panel_ws_len)
{
    if (!c->asm_fuse || (ntot & 63) || (ncols & 63) || ncols > ntot ||
        !dfirst_applies(c, ntot, ncols, batch))
        return 0;
    const int NB = std::min(auto_nb(c, ntot, batch), ncols);
    if (panel_ws_len < dfirst_ws_doubles(NB, batch) || NB >= ncols)
        return 0;
    return NB;
}

static int enqueue_potrf_dfirst(bq_ctx *c, double *A, long lda, long astride, int batch, int ntot,
                                int ncols, int *info, double *ws, bool skip_border)
{
    const int NB = std::min(auto_nb(c, ntot, batch), ncols);
    double *recs[2] = {ws + panel_ws_doubles(NB, batch),
                       ws + panel_ws_doubles(NB, batch) + dfirst_rec_doubles(NB, batch)};
    const long rstride = (long)(NB / 64) * BQ_DINV_HALF;
    const bool la = c->lookahead && c->aux && c->cur == c->stream;
    // df_early: the streams fork BEFORE the panel solve.  The next diagonal block's chain -- the
    // panel solve of ITS rows (the top nw of the panel), its update, its factor -- starts on the
    // second stream at once, beside the solve of the other rows; round 4 forked after the whole
    // panel solve, and in the late blocks the update was shorter than the chain it should hide
    // (C5 shard: 0.11-0.14 ms exposed after three of four blocks, tools/plan_timeline.py).
    const bool early = la && c->df_early;
    // A factor that has a long update to hide behind (df_wg_rows rows or more below its block) goes
    // to the workgroup-per-matrix kernel whatever the batch: slower alone (366 against 236 us for
    // 64 blocks of 448), but its `batch` workgroups take far fewer slots from the update beside it
    // than the one-launch steps' redundant tiles (C5 shard 5.65 -> 5.57 ms; the first block's
    // factor and the late ones, which nothing hides, stay on the steps)
    auto diag = [&](int K0, int KB, double *rec) {
        if (dfirst_wg(c, batch) ||
            (c->df_wg < 0 && c->df_wg_rows > 0 && K0 > 0 && ntot - K0 - KB >= c->df_wg_rows))
            return launch_potrf_wg(c, A + K0 + (long)K0 * lda, lda, astride, KB, rec, rstride, info,
                                   K0, batch);
        return enqueue_slab_sweep(c, A + K0 + (long)K0 * lda, lda, astride, batch, KB, KB, rec, info,
                                  ws, K0, false, rstride);
    };
    BQCHK(diag(0, NB, recs[0]));
    Sharing scope(c, la ? c->df_sharing : c->sharing);
    // The caller assembled only the first NB columns (dfirst_seed_cols) and left the rest as a
    // GramSeed: block 0's three products are the first to touch everything right of them, and they
    // compute their tiles of the system instead of loading them -- the assembly of two thirds of a
    // C5 system (an HBM-write-bound launch) and its read-back disappear.
    const bool seeded = c->gram_seed.pts != nullptr;
    auto seed_at = [&](int r, int cc) {
        GramSeed sd = c->gram_seed;
        sd.r = r;
        sd.c = cc;
        return sd;
    };
    int par = 0;
    for (int K0 = 0; K0 < ncols; K0 += NB, par ^= 1) {
        const int KB = std::min(NB, ncols - K0), r0 = K0 + KB, m2 = ntot - r0;
        if (m2 <= 0)
            break;
        const double *rec = recs[par];
        const double *P = A + r0 + (long)K0 * lda;
        const int nw = r0 < ncols ? std::min(NB, ncols - r0) : 0;
        if (nw == 0) {
            BQCHK(enqueue_panel_solve(c, A, lda, astride, batch, r0, m2, K0, KB, rec, rstride));
            // what is left is the Schur complement of the border: nobody reads it when the
            // results come off the border rows
            if (!skip_border)
                BQCHK(launch_gemm(c, BQ_K_SYRK, A + r0 + (long)r0 * lda, lda, astride, P, lda,
                                  astride, P, 1, lda, astride, m2, m2, KB, 1, batch));
            break;
        }
        const int r1 = r0 + nw;
        if (!early)
            BQCHK(enqueue_panel_solve(c, A, lda, astride, batch, r0, m2, K0, KB, rec, rstride));
        // the next diagonal block: (its rows of the panel solved,) updated and factored -- on the
        // second stream, beside the rest
        if (la) {
            HIPCHK(c, hipEventRecord(c->ev_next, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_next, 0));
            c->cur = c->aux;
        }
        int st = BQ_OK;
        if (early) {
            st = enqueue_panel_solve(c, A, lda, astride, batch, r0, nw, K0, KB, rec, rstride);
            // (no early return while c->cur is the second stream: a failed record falls through
            // to the block below that restores it -- ADVICE r05)
            if (st == BQ_OK && hipEventRecord(c->ev_top, c->aux) != hipSuccess)
                st = fail(c, BQ_ERR_HIP, "hipEventRecord(ev_top)");
        }
        const bool sd0 = seeded && K0 == 0;
        if (st == BQ_OK) {
            const GramSeed sd = seed_at(r0, r0);
            st = launch_gemm(c, BQ_K_SYRK, A + r0 + (long)r0 * lda, lda, astride, P, lda, astride, P,
                             1, lda, astride, nw, nw, KB, 1, batch, -1, nullptr, 0, nullptr, 0,
                             sd0 ? &sd : nullptr);
        }
        if (st == BQ_OK)
            st = diag(r0, nw, recs[par ^ 1]);
        if (la) {
            c->cur = c->stream;
            if (st == BQ_OK)
                HIPCHK(c, hipEventRecord(c->ev_panel, c->aux));
        }
        BQCHK(st);
        if (r1 < ntot) {
            const double *P1 = A + r1 + (long)K0 * lda;
            if (early) {
                BQCHK(enqueue_panel_solve(c, A, lda, astride, batch, r1, ntot - r1, K0, KB, rec,
                                          rstride));
                // the column update reads the top rows as its Q operand
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_top, 0));
            }
            // the rows below it: the next panel's columns, then the square behind them.  (ONE
            // launch over everything right of the panel with the next diagonal block's tiles
            // skipped -- one launch tail instead of two -- was measured in round 5 and gained
            // nothing: C5 5.67 / 256 x C2 4.21 / C3 177-183 either way; removed.  So was the square
            // deferred to every second panel and run once with both panels as its operand, k = 2 NB
            // -- bit-identical results, C5 5.59 / 256 x C2 4.25 / C3 174-176 against 5.60 / 4.21 /
            // 176: what the deeper product gains, the factor that now hides behind the column
            // update alone gives back.)
            const bool sq = !(skip_border && r1 >= ncols);
            const GramSeed sdc = seed_at(r1, r0), sds = seed_at(r1, r1);
            BQCHK(launch_gemm(c, BQ_K_SYRK, A + r1 + (long)r0 * lda, lda, astride, P1, lda, astride,
                              P, 1, lda, astride, ntot - r1, nw, KB, 0, batch, -1, nullptr, 0,
                              nullptr, 0, sd0 ? &sdc : nullptr));
            if (sq)
                BQCHK(launch_gemm(c, BQ_K_SYRK, A + r1 + (long)r1 * lda, lda, astride, P1, lda,
                                  astride, P1, 1, lda, astride, ntot - r1, ntot - r1, KB, 1, batch,
                                  -1, nullptr, 0, nullptr, skip_border ? ncols - r1 : 0,
                                  sd0 ? &sds : nullptr));
        }
        if (la)
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_panel, 0));
    }
    return BQ_OK;
}

// Eliminate the first ncols columns of `batch` matrices.  A batch of mid-sized matrices
// (config C5: 64 x N = 2048) sweeps in lock-step: every 64-column panel step is two short
// dependent launches that leave most of the chip idle, and a third of the sweep's time is
// such panel work.  The batch is therefore cut in two halves on the two streams: one half's
// panel chain runs beside the other half's MFMA trailing update (C5 shard: 7.21 -> 6.78 ms;
// three or four groups on more streams were slower, 7.8 / 7.5 ms).  (Large single matrices
// use the second stream for the look-ahead instead, small ones the one-launch steps.)
int enqueue_potrf_partial(bq_ctx *c, double *A, long lda, long astride, int batch, int ntot,
                          int ncols, double *dinv, int *info, double *panel_ws,
                          size_t panel_ws_len, bool first_done, bool skip_border)
{
    if ((ntot & 63) || (ncols & 63) || ncols > ntot)
        return fail(c, BQ_ERR_BAD_ARG, "potrf: sizes must be multiples of 64");
    const int NB = auto_nb(c, ntot, batch);
    if (dfirst_applies(c, ntot, ncols, batch) && panel_ws &&
        panel_ws_len >= dfirst_ws_doubles(std::min(NB, ncols), batch)) {
        // (Two half-batches on the two streams, each in lock-step with its diagonal factors on its
        // own stream, were measured in rounds 4 and 5 -- in phase: C5 shard 5.77 ms / 256 x C2 4.33
        // against 5.67 / 4.21 for the whole batch in lock-step; out of phase, the second half
        // starting when the first half's first panel solve is through, so that one half's
        // panel-solve workgroups and diagonal chains run beside the other half's update tiles:
        // 5.87 / 4.52-4.81, C3 186-188 against 177 -- and removed: every launch of a half fills
        // the chip half as well, and two MFMA-bound kernels gain nothing from sharing CUs.)
        return enqueue_potrf_dfirst(c, A, lda, astride, batch, ntot, ncols, info, panel_ws,
                                    skip_border);
    }
    if (c->gram_seed.pts)
        return fail(c, BQ_ERR_BAD_ARG, "potrf: a partly assembled system on a sweep that cannot seed it");
    // large systems (a look-ahead's size) in a batch that fills the chip many times over run
    // as ONE sequential group: the product is power-bound (docs/LABBOOK.md section 4), beside it the
    // panel chain only takes clock away (C3, 100 x N = 4096: 189.8 ms with the look-ahead,
    // 198 in two halves, 186.3 sequential; up to 32 matrices the look-ahead still gains 1-2 %)
    const bool big = ncols > NB && ntot - std::min(NB, ncols) >= c->la_min;
    if (c->split_batch && c->lookahead && c->aux && c->cur == c->stream && batch >= 8 &&
        NB >= 128 && !big) {
        const int b0 = batch / 2, b1 = batch - b0;
        Sharing halves(c, 2);
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->aux, c->ev_fork, 0));
        c->cur = c->aux;
        int st = enqueue_potrf_group(c, A + (long)b0 * astride, lda, astride, b1, ntot, ncols,
                                     dinv + (long)b0 * BQ_DINV_STRIDE, info + b0, nullptr, 0, NB,
                                     false, skip_border);
        c->cur = c->stream;
        if (st != BQ_OK)
            return st;
        HIPCHK(c, hipEventRecord(c->ev_panel, c->aux));
        BQCHK(enqueue_potrf_group(c, A, lda, astride, b0, ntot, ncols, dinv, info, nullptr, 0, NB,
                                  false, skip_border));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_panel, 0));
        return BQ_OK;
    }
    return enqueue_potrf_group(c, A, lda, astride, batch, ntot, ncols, dinv, info, panel_ws,
                               panel_ws_len, 0, first_done, skip_border);
}

} // namespace bqh
