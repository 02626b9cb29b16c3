// host.h -- internal header of libbqhip.so's host side: the context, the handle types and
// the launch / enqueue helpers the translation units share.  Nothing here is part of the C
// ABI (include/bqhip.h is); the library is built with hidden visibility and exports only
// the extern "C" entry points.
//
// Translation units (Makefile):
//   k_gram.hip   gram.h kernels                 launch_gram_sym / _cross
//   k_gemm.hip   gemm.h kernels                 launch_gemm, launch_gemm_rows, launch_rows_step
//   k_panel.hip  potf2.h trsm.h slab.h kernels  launch_assemble, launch_potf2, launch_trsm_blk,
//                                               the one-launch steps
//   k_reduce.hip reduce.h trsv.h kernels        read-outs, single-vector sweeps, utilities
//   potrf.hip    the blocked factorisation's launch sequences (no kernels of its own)
//   sweeps.hip   sweeps over a resident factor (no kernels of its own)
//   ctx.hip      contexts, device memory, timers, the launch profiler
//   linalg.hip   linalg_c drop-ins, Gram entry points, bq_potrf_dev
//   plan.hip     resident batched plans, batched / grid entry points
//   fit.hip      resident GP fits
//   moments.hip  closed-form integrals, BQ moments, the acquisition entry points
//   pair.hip     the stacked pair of GPs at S hyper-parameter sets in one batched pass
//   probe.hip    hardware probes
#pragma once
#include <hip/hip_runtime.h>

#pragma GCC visibility push(default)
#include "../../include/bqhip.h"
#pragma GCC visibility pop

#include "types.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

// Guard bands (tests): with bq_set_guard(1) every device buffer allocated afterwards carries
// BQ_GUARD_BYTES of 0xA5 behind its last byte; bq_plan_check_guards counts the bytes of a plan's
// bands that a pass has overwritten (round 5 found a workspace sized for the wrong block width
// only because the next allocation happened to fault).
#define BQ_GUARD_BYTES 4096
inline bool &devbuf_guard()
{
    static bool on = std::getenv("BQ_GUARD") && std::atoi(std::getenv("BQ_GUARD"));
    return on;
}

// RAII device buffer
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    size_t guard = 0; // bytes of sentinel behind `bytes` (0: none)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        guard = 0;
    }
    hipError_t alloc(size_t b)
    {
        release();
        if (b == 0)
            b = 8;
        const size_t g = devbuf_guard() ? BQ_GUARD_BYTES : 0;
        hipError_t e = hipMalloc(&p, b + g);
        if (e == hipSuccess && g)
            e = hipMemset(static_cast<char *>(p) + b, 0xA5, g);
        if (e == hipSuccess) {
            bytes = b;
            guard = g;
        } else {
            if (p)
                (void)hipFree(p);
            p = nullptr;
        }
        return e;
    }
    // bytes of the guard band that no longer hold the sentinel (-1: the read failed)
    long guard_damage() const
    {
        if (!p || !guard)
            return 0;
        unsigned char h[BQ_GUARD_BYTES];
        if (hipMemcpy(h, static_cast<const char *>(p) + bytes, guard, hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
        long bad = 0;
        for (size_t i = 0; i < guard; ++i)
            bad += h[i] != 0xA5;
        return bad;
    }
    double *d() const { return static_cast<double *>(p); }
    int *i() const { return static_cast<int *>(p); }
};

struct ProfEvent {
    hipEvent_t a, b;
    int cls;
    double work;
    int on_aux; // recorded on the context's second stream
};

struct bq_ctx {
    int device = 0;
    hipStream_t stream = nullptr; // main stream: everything is ordered on it
    hipStream_t aux = nullptr;    // high-priority panel stream of the look-ahead Cholesky
    hipStream_t cur = nullptr;    // stream the launch helpers enqueue on (stream or aux)
    hipEvent_t ev_panel = nullptr, ev_next = nullptr, ev_fork = nullptr, ev_top = nullptr;
    int lookahead = 1;
    int split_batch = 1; // halves of a mid-sized batch on the two streams (BQ_SPLIT=0: lock-step)
    int diag_first = 1;  // batches: every outer block as diagonal factor, ONE panel solve, update
                         // (enqueue_potrf_dfirst; BQ_DIAG_FIRST=0: the recursive panels)
    int df_sweep = 1;    // the panel solve of an outer block in one launch (trsm_sweep_kernel; BQ_DF_SWEEP)
    int df_wg = -1;      // a batch's diagonal factor by one workgroup per matrix (potrf_wg_kernel): -1 by
                         // batch size (potrf.hip, dfirst_wg), 0 / 1 forced (BQ_DF_WG)
    int trsv_flow = 1;   // single-vector sweeps as one launch each, hand-offs through memory
                         // (trsvflow.h; BQ_TRSV_FLOW=0: one launch per block column)
    int trsv_flow_min = 2048; // ... from this many rows on (BQ_TRSV_FLOW_MIN)
    int *flow_abort = nullptr; // mapped host word a timed-out hand-off raises
    long n_flow_fallback = 0;  // solves re-issued on the per-block sweeps after such a time-out
                               // (bq_ctx_stats)
    int pair_border = 1; // bq_pair_esm as S factorisations + border rows (BQ_PAIR_BORDER=0: the S Ma
                         // full bordered systems)
    int df_wg_rows = 1000; // ... and, below that batch size, for the blocks with at least this many rows
                         // below them: the factors an update hides (BQ_DF_WG_ROWS; 0: never)
    int df_early = 1;    // the diagonal-first sweep forks before the panel solve: the next diagonal block's
                         // rows are solved, updated and factored beside the rest of the solve (BQ_DF_EARLY)
    int rows_tail = 128; // a large row sweep's last updates as split-k tiles: from this many LDS tiles down (BQ_ROWS_TAIL)
    int solve_kcopy = 1; // a one-vector solve's vector in / out and sentinel fill by kernels (BQ_SOLVE_KCOPY)
    double *hstage = nullptr; // mapped pinned staging of the small host-buffer calls (ctx_stage)
    size_t hstage_len = 0;
    int df_sharing = 0;  // gemm_lds_tile's sharing mode while a diagonal factor runs beside an update
                         // (0: the rule of a product alone -- C5 shard 5.73 ms against 6.05 with 1)
    int la_min = 3072;   // look-ahead only while the bulk update has at least this many rows (BQ_LA_MIN)
    DevBuf panel_ws;     // scratch panel columns of the eager linalg entry points
    DevBuf scratch;      // per-call temporaries of the acquisition / moment entry points, kept
                         // between calls (hipFree synchronises the device); bq_ctx_trim frees it
    int gemm_lds = 1;    // LDS-staged 128x128 trailing update (BQ_GEMM_LDS)
    int gemm_lds64 = 1;  // its 64x64-tile form for products that cannot fill the chip (BQ_GEMM_LDS64)
    int slab_nb_max = 3072; // one or two matrices below this size: one-launch steps throughout (BQ_SLAB_NB_MAX)
    int slab_max = 4800;    // ... and the last rows of a larger one, from this many on (BQ_SLAB_MAX)
    int fold_readout = 1; // one-launch sweeps carry their read-out (SlabOut; BQ_FOLD_READOUT)
    SlabOut slab_out{};  // set (scal != nullptr) by a caller whose slab sweep carries its read-out
    int asm_fuse = 1;    // a batched plan assembles only the first outer block's columns; the rest of
                         // the system is computed inside the first products that touch it
                         // (GramSeed; BQ_ASM_FUSE=0: the whole system is assembled first)
    GramSeed gram_seed{}; // set (pts != nullptr) by a caller that assembled only dfirst_seed_cols()
                         // columns: enqueue_potrf_dfirst seeds the first block's products with it
    int potf2_8w = 1;    // the one-launch steps' diagonal factor on eight waves where a step's workgroups
                         // have a CU each (BQ_POTF2_8W)
    int gemm_ksplit = 1; // eight-wave k-split forms of the 64-tile / job kernels (BQ_GEMM_KSPLIT)
    int gemm_tile = 0;   // 64 / 128: force the LDS kernel's workgroup tile (BQ_GEMM_TILE; measurements)
    int sharing = 0;     // how the chip is shared while the launches being queued run (gemm_lds_tile):
                         // 0 alone, 1 the two streams of a look-ahead, 2 the two halves of a batch
    int use_graph = 1;   // replay plans from a captured hipGraph (BQ_GRAPH=0 disables)
    bool own_stream = false;
    int cus = 256;
    int nb_override = 0;
    bq_plan *plan_cache = nullptr; // workspace of the last batched call, kept for the next one
    char err[512] = {0};
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool prof = false;
    std::vector<ProfEvent> prof_events;
    // bq_profile_timeline: (class, stream, start, end in ms since the first bracket) per launch
    bool prof_keep_timeline = false;
    hipEvent_t prof_origin = nullptr;
    std::vector<double> prof_timeline;
    double prof_ms[BQ_K_NCLASS] = {0};
    int64_t prof_n[BQ_K_NCLASS] = {0};
    double prof_work[BQ_K_NCLASS] = {0};
    DevBuf gbuf;   // GaussParams of the single-problem entry points (cached)
    GaussParams gbuf_host{};
    bool gbuf_valid = false;
    DevBuf dinv64; // potf2 reciprocal-diagonal scratch
    long long *stamp_buf = nullptr; // bq_probe_c2_timeline: 160 stamps per slab step
    int stamp_steps = 0;            // slab steps stamp_buf has room for
};

namespace bqh {

inline int fail(bq_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof c->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

#define HIPCHK(c, call)                                                                        \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return bqh::fail((c), e__ == hipErrorOutOfMemory ? BQ_ERR_NOMEM : BQ_ERR_HIP,      \
                             "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                             __LINE__);                                                        \
    } while (0)

#define BQCHK(call)                                                                            \
    do {                                                                                       \
        int s__ = (call);                                                                      \
        if (s__ != BQ_OK)                                                                      \
            return s__;                                                                        \
    } while (0)

inline long roundup(long v, long q) { return (v + q - 1) / q * q; }

// Every context field that decides which launches a sweep is made of, hashed: a captured
// hipGraph (plans, the pair's objective) is replayed only while this is what it was captured
// under -- whichever setter, environment switch or probe changed a field.
inline unsigned long long launch_config_key(const bq_ctx *c)
{
    const int f[] = {c->nb_override, c->lookahead, c->split_batch, c->la_min, c->gemm_lds,
                     c->gemm_lds64, c->slab_nb_max, c->slab_max, c->fold_readout, c->potf2_8w,
                     c->gemm_ksplit, c->gemm_tile, c->diag_first, c->df_sweep, c->df_wg,
                     c->df_sharing, c->rows_tail, c->df_early, c->df_wg_rows};
    unsigned long long h = 1469598103934665603ull;
    for (int v : f)
        h = (h ^ (unsigned long long)(unsigned)v) * 1099511628211ull;
    return h;
}

// leading dimension for an ntot x ntot column-major matrix: even, and nudged
// off large powers of two so that the 4 columns of an MFMA fragment do not
// all map to the same HBM channel / L2 set
inline long pick_ld(long ntot)
{
    long ld = ntot;
    if (ntot >= 1024 && (ntot % 512) == 0)
        ld += 64;
    return ld;
}

inline GaussParams make_params(int d, double h, const double *w, double s)
{
    GaussParams g;
    std::memset(&g, 0, sizeof g);
    double c = h * h;
    for (int k = 0; k < d; ++k) {
        c /= (std::sqrt(2.0 * M_PI) * w[k]);
        g.nh[k] = -0.5 / (w[k] * w[k]);
    }
    g.c = c;
    g.s2 = s * s;
    return g;
}

inline Layout make_layout(int n, int M, bool has_y)
{
    Layout L;
    L.n = n;
    L.npad = (int)roundup(n, 64);
    L.M = M;
    L.yrow = has_y ? L.npad + M : -1;
    L.ntot = (int)roundup(L.npad + M + (has_y ? 1 : 0), 64);
    return L;
}

// ---- profiling brackets: HIP events around a launch, on the stream it goes to ----
struct Bracket {
    bq_ctx *c;
    ProfEvent ev;
    bool on;
    Bracket(bq_ctx *ctx, int cls, double work = 0.0) : c(ctx), on(ctx->prof)
    {
        if (on) {
            ev.cls = cls;
            ev.work = work;
            ev.on_aux = ctx->cur != ctx->stream;
            if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) {
                on = false;
                return;
            }
            (void)hipEventRecord(ev.a, c->cur);
        }
    }
    ~Bracket()
    {
        if (on) {
            (void)hipEventRecord(ev.b, c->cur);
            c->prof_events.push_back(ev);
        }
    }
};

int prof_collect(bq_ctx *c); // ctx.hip

// Carves the per-call temporaries of one entry point out of the context's scratch buffer:
// sizes first (take), then one commit that grows the buffer if it must, then the pointers.
struct Scratch {
    bq_ctx *c;
    size_t total = 0;
    explicit Scratch(bq_ctx *ctx) : c(ctx) {}
    size_t take(size_t doubles)
    {
        const size_t off = total;
        total += (doubles + 31) & ~(size_t)31; // 256-byte granules
        return off;
    }
    int commit()
    {
        if (c->scratch.bytes < total * sizeof(double)) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, c->scratch.alloc(total * sizeof(double)));
        }
        return BQ_OK;
    }
    double *at(size_t off) const { return c->scratch.d() + off; }
};

int check_dims(bq_ctx *c, int64_t d, int64_t n);                          // ctx.hip
int check_w(bq_ctx *c, int64_t d, double h, const double *w, double s);   // ctx.hip

// ---- k_gram.hip -----------------------------------------------------------------------
int launch_gram_sym(bq_ctx *c, int d, const double *x, long xstride, const GaussParams *gp,
                    int gpstride, double *K, long ldk, long kstride, int n, int batch);
int launch_gram_cross(bq_ctx *c, int d, const double *x1, int n1, const double *x2, int n2,
                      const GaussParams &g, double *K, long ldk);
int launch_gram_cross_pad(bq_ctx *c, int d, const double *x1, int n1, int n1p, const double *x2,
                          int n2, int n2p, const GaussParams &g, double *K, long ldk);

// ---- k_panel.hip ----------------------------------------------------------------------
// fs: when set, the first launch of the slab sweep rides in the assembly (assemble_first_kernel)
struct FirstStep {
    double *S0 = nullptr;
    long lds = 0, sstride = 0;
    double *dinv = nullptr;
    int *info = nullptr;
    double *scal = nullptr; // SlabOut::scal: log|K| starts at zero here
};
// jcols > 0: only the first jcols columns (a multiple of 64) of every system
int launch_assemble(bq_ctx *c, int d, const double *pts, long pstride, const double *y,
                    long ystride, const GaussParams *gp, int gpstride, double *A, long lda,
                    long astride, Layout L, int batch, const FirstStep &fs = FirstStep(),
                    int jcols = 0);
// rows [sd.r, sd.r + m) x columns [sd.c, sd.c + n) of the systems sd describes (lower part)
int launch_assemble_region(bq_ctx *c, const GramSeed &sd, double *A, long lda, long astride, int m,
                           int n, int batch);
int launch_potf2(bq_ctx *c, double *A, long lda, long astride, int j0, double *dinv, long dstride,
                 int *info, int batch);
int launch_potrf_wg(bq_ctx *c, double *A, long lda, long astride, int kb, double *rec, long rstride,
                    int *info, int col0, int batch);
int launch_trsm_blk(bq_ctx *c, double *X, long ldx, long xstride, int m, const double *L11,
                    long ldl, long lstride, const double *dinv, long dstride, int batch);
// the 16 x 16 block inverses of every 64 x 64 diagonal block of a factor (npad / 64 records)
int launch_diag_winv(bq_ctx *c, const double *L, long ldl, int npad, double *dw);
int launch_panel_step(bq_ctx *c, double *A, long lda, long astride, int batch, int nrb,
                      double *Sin, double *Sout, long lds, long sstride, int K0, int j0,
                      double *dinv_in, double *dinv_out, int has_next, int first, double *SL,
                      int *info, double work);
int launch_slab_first(bq_ctx *c, double *A, long lda, long astride, int batch, double *S, long lds,
                      long sstride, int ntot, double *dinv, int *info, int col0,
                      long dstride = BQ_DINV_STRIDE);
int launch_slab_step(bq_ctx *c, double *A, long lda, long astride, int batch, double *Sin,
                     double *Sout, long lds, long sstride, int ntot, int j0, double *dinv_in,
                     double *dinv_out, int fnext, int last, int *info, int col0,
                     long long *stamps, double work, long dstride = BQ_DINV_STRIDE);

// ---- k_gemm.hip -----------------------------------------------------------------------
int gemm_init(bq_ctx *c); // function attributes of the LDS-staged kernels, once per context
bool gemm_uses_lds(const bq_ctx *c, int m, int n, int k, int lower, int batch);
// C(m x n) -= P(m x k) Q(n x k)^T; see k_gemm.hip
int launch_gemm(bq_ctx *c, int cls, double *C, long ldc, long cstride, const double *P, long ldp,
                long pstride, const double *Q, long qsj, long qsk, long qstride, int m, int n,
                int k, int lower, int batch, int fuse_j0 = -1, double *dinv = nullptr,
                long dstride = 0, int *info = nullptr, int ccut = 0,
                const GramSeed *seed = nullptr);
bool gemm_trsm_ok(const bq_ctx *c, int m, int n, int k);
int launch_gemm_trsm(bq_ctx *c, double *C, long ldc, long cstride, const double *P, long ldp,
                     long pstride, const double *Q, long ldq, long qstride, int m, int n, int k,
                     const double *Lss, long ldl, long lstride, const double *wrec, long wstride,
                     int batch);
int launch_trsm_sweep(bq_ctx *c, double *X, long ldx, long xstride, int m, const double *L11,
                      long ldl, long lstride, const double *rec, long rstride, int kb, int batch);
int launch_gemm_rows(bq_ctx *c, int cls, double *C, long ldc, const double *P, long ldp,
                     const double *Q, long qsj, long qsk, int m, int n, int k);
int launch_rows_step(bq_ctx *c, int mrows, const RowsJob &a, const RowsJob &b, double work);
int launch_rows_fused(bq_ctx *c, int mrows, const RowsJob &a, double *C, long ldc, const double *P,
                      long ldp, const double *Q, long ldq, int n, int k, bool qt, double work);

// ---- k_reduce.hip ---------------------------------------------------------------------
int launch_finalize(bq_ctx *c, const double *A, long lda, long astride, Layout L, double *scal,
                    double *mean, double *var, long mstride, int batch, double work = 0.0);
int launch_plan_readout(bq_ctx *c, const double *A, long lda, long astride, Layout L,
                        const GaussParams *gp, double *scal, double *mean, double *var,
                        long mstride, int batch);
int launch_rowdot(bq_ctx *c, const double *V, long ldv, int M, int Mp, int npad, const double *z,
                  double k0, double *mean, double *var, long zstride = 1);
int launch_predict_mean(bq_ctx *c, int d, const double *xo, int M, const double *pts, int n,
                        const double *alpha, const GaussParams &g, double *mean);
int launch_neg_identity(bq_ctx *c, double *nr, int B, int npad);
int launch_pad_identity(bq_ctx *c, double *A, long lda, int n, int ntot);
int launch_logdet(bq_ctx *c, const double *diag, long stride, int n, double *out);
int launch_transpose_pad(bq_ctx *c, const double *src, long lds, int rows, int cols, double *dst,
                         long ldd);
int launch_transpose_blocks(bq_ctx *c, const double *src, double *dst, int B, long bstride,
                            int bx, int by, int batch);
int launch_neg_sumsq(bq_ctx *c, const double *v, int n, double *out);
int launch_trsv_fwd(bq_ctx *c, const double *L, long ldl, int J, int bJ, int B, int nupd,
                    const double *nr, const double *tt, double *x, double *y, double work);
int launch_trsv_bwd(bq_ctx *c, const double *L, long ldl, int J, int bJ, int B, int bn, int nupd,
                    const double *nt, const double *uu, double *x, double *y, double work);
// a whole sweep in one launch (trsvflow.h); flow_check: BQ_ERR_HIP if a hand-off of a sweep that
// has completed on the stream timed out (call after synchronising)
bool trsv_flow_ok(const bq_ctx *c, int npad, int B, bool prefilled = false);
size_t trsv_flow_ws_doubles(int npad, int B);
// preset: ws and y hold the sentinel already (launch_flow_in: the right-hand side in from a mapped
// pinned vector + every hand-off slot of both sweeps, one launch; launch_flow_out: the solution out)
int launch_trsv_flow(bq_ctx *c, bool forward, const double *L, long ldl, int npad, int B,
                     const double *m1, const double *m2, const double *x0, double *y, double *ws,
                     bool preset = false);
int launch_flow_in(bq_ctx *c, const double *hsrc, int n, double *x, int npad, double *fill,
                   size_t nfill);
int launch_flow_out(bq_ctx *c, const double *x, int n, double *hdst);
int launch_gather_row(bq_ctx *c, double *dst, const double *src, long stride, int n);
// small host matrices through the context's mapped pinned staging buffer (trsvflow.h)
int launch_mat_in(bq_ctx *c, const double *stage, int n, double *A, long lda, int ntot, int *info);
int launch_mat_out(bq_ctx *c, double *out, const double *A, long lda, int n, const int *info);
int ctx_stage(bq_ctx *c, size_t words, double **host, double **dev);
// (L L^T) X = B, n <= 64, nrhs <= 64, one launch on the staging buffer [X | L | B]
int launch_small_potrs(bq_ctx *c, double *stage, int n, int nrhs);
// a small plan's inputs out of / results into one mapped pinned staging buffer (trsvflow.h)
int launch_plan_scatter(bq_ctx *c, const double *stage, int nprob, int d, int n, int M, int ntot,
                        int npad, int gw, double *gp, double *pts, double *yd);
int launch_plan_gather(bq_ctx *c, double *out, const double *scal, const int *info,
                       const double *mean, const double *var, int nb, int M);
// up to two small copies of 8-byte words in one launch (either side may be mapped pinned memory)
int launch_copy_words2(bq_ctx *c, void *d1, const void *s1, size_t n1, void *d2, const void *s2,
                       size_t n2);
int flow_check(bq_ctx *c);

// ---- potrf.hip ------------------------------------------------------------------------
int auto_nb(const bq_ctx *c, int ntot, int batch);
// columns a caller of enqueue_potrf_partial has to assemble before it when it hands the rest over
// as c->gram_seed (the first outer block's), 0: the whole system (another sweep will run)
int dfirst_seed_cols(const bq_ctx *c, int ntot, int ncols, int batch, size_t panel_ws_len);
size_t panel_ws_doubles(int ntot, int batch);
size_t sweep_ws_doubles(const bq_ctx *c, int ntot, int batch);
bool panel_ws_useful(const bq_ctx *c, int ntot, int batch);
bool sweep_is_slab(const bq_ctx *c, int ntot, int ncols, int batch, size_t panel_ws_len);
int enqueue_potrf_partial(bq_ctx *c, double *A, long lda, long astride, int batch, int ntot,
                          int ncols, double *dinv, int *info, double *panel_ws = nullptr,
                          size_t panel_ws_len = 0, bool first_done = false,
                          bool skip_border = false);

int enqueue_panel_solve(bq_ctx *c, double *A, long lda, long astride, int batch, int r0, int m2,
                        int K0, int KB, const double *rec, long rstride);

// ---- sweeps.hip -----------------------------------------------------------------------
struct WideInv {
    const double *nr = nullptr; // -W^T of every block
    // the single-vector sweeps (trsv.h):
    const double *nt = nullptr; // -W (the transposes)
    const double *tt = nullptr; // T_J^T, T_J = W_J L[J, J-B] (blocks J >= B)
    const double *uu = nullptr; // U_J = L[J+B, J] W_J (all blocks but the last)
    const double *t = nullptr;  // T_J itself (rows of T contiguous: the fused row-sweep step)
    int B = 0;
};
inline int wide_block(int npad)
{
    // (BQ_WIDE_B: measurements)
    static const int forced = std::getenv("BQ_WIDE_B") ? std::atoi(std::getenv("BQ_WIDE_B")) : 0;
    if (forced > 0)
        return std::min(npad, forced);
    // (round 5, tools/wide_b_check.py, 256 against 512 on resident fits of 1024 / 1536 points:
    // posterior mean + variance at 256 points 0.064 -> 0.056 / 0.094 -> 0.073 ms, one-vector solve
    // 0.053 -> 0.039 / 0.070 -> 0.047 -- half the dependent steps --; the inverses cost 0.06 ms
    // more to rebuild after a refit: 0.43 -> 0.50 ms for refit + first posterior)
    return npad < 1024 ? std::min(npad, 256) : 512;
}
inline size_t wide_doubles(int npad) { return (size_t)npad * wide_block(npad); }
// NR, NT, TT, UU and the scratch of T before its transposition (a full B x B per block)
inline size_t wide_alloc_doubles(int npad)
{
    const size_t B = (size_t)wide_block(npad);
    return 5 * wide_doubles(npad) + B * B;
}
WideInv wide_views(const double *base, int npad);
int compute_wide_inverses(bq_ctx *c, const double *L, long ldl, int npad, const double *dw,
                          double *nr);
// ws: trsv_flow_ws_doubles(npad, w.B) doubles for the one-launch form (nullptr: a launch per block)
int enqueue_forward_vec(bq_ctx *c, double *x, double *y, const double *L, long ldl, int npad,
                        WideInv w, double *ws = nullptr);
int enqueue_backward_vec(bq_ctx *c, double *x, double *y, const double *L, long ldl, int npad,
                         WideInv w, double *ws = nullptr);
int enqueue_forward_rows_blk(bq_ctx *c, double *X, long ldx, int mrows, const double *L, long ldl,
                             int npad, const double *dw);
int enqueue_forward_rows(bq_ctx *c, double *Xin, double *Xout, long ldx, int mrows,
                         const double *L, long ldl, int npad, WideInv w);
int enqueue_backward_rows(bq_ctx *c, double *Xin, double *Xout, long ldx, int mrows,
                          const double *L, long ldl, int npad, WideInv w);
int solve_rows_host(bq_ctx *c, const double *L, long ldl, int n, int npad, WideInv w,
                    const double *B, int64_t nrhs, double *X);

} // namespace bqh

// ---- handle types (plan.hip, fit.hip) ---------------------------------------------------
struct bq_plan {
    int nprob = 0, d = 0, n = 0, M = 0;
    Layout L{};
    long lda = 0, astride = 0;
    DevBuf A, pts, y, gp, dinv, info, scal, mean, var;
    DevBuf panel; // scratch panel columns of the one-launch slab sweep (small systems)
    std::vector<GaussParams> hgp;
    bool has_inputs = false;
    // the launch sequence of a plan is static: it is captured once into a hipGraph
    // and replayed (cuts the host launch cost of the ~50 short kernels of a step)
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    int graph_state = 0; // 0 = not tried, 1 = ready, -1 = unavailable (eager launches)
    unsigned long long graph_key = 0; // launch_config_key the graph was captured under
    double *hres = nullptr; // pinned staging of bq_plan_results: [scal 4 nb | info nb | mean | var]
    double *hin = nullptr;  // pinned staging of a small plan's inputs (bq_plan_set_inputs)
    size_t hin_len = 0;
    bool in_flight = false; // the scatter out of hin may not have run yet
    ~bq_plan()
    {
        if (hres)
            (void)hipHostFree(hres);
        if (hin)
            (void)hipHostFree(hin);
    }
};

struct bq_fit {
    int d = 0, n = 0, npad = 0;
    long ldl = 0;
    Layout L{}; // layout of the fit system (M = 0, y row)
    double h = 0, s = 0, w[BQ_MAXD] = {0};
    GaussParams g{};
    DevBuf A;     // ntot x ntot bordered factor: L in [0,npad)^2, z in row yrow
    DevBuf pts;   // d x ntot
    DevBuf y;     // npad
    DevBuf gp;    // GaussParams
    DevBuf dinv;  // npad reciprocal diagonal (+ BQ_DINV_STRIDE scratch for the factorisation)
    DevBuf panel; // scratch panel columns of the one-launch slab sweep
    DevBuf dw;    // diag_winv_kernel records of the resident factor (MFMA solves in the sweeps)
    DevBuf wide;  // -W^T of the B-wide diagonal blocks (row sweeps), valid if have_wide
    bool have_wide = false;
    bool have_dw = false; // dw is built on its first use: a loop that reads log-ML never pays
    DevBuf wV, wV2, wx, wout, wz; // prediction workspaces, grown on demand and kept
    DevBuf misc;  // info (int) + scal[4], then 2 x 64 doubles: the posterior of the border points
                  // of bq_gp_refit_predict (one read-back for all of it)
    DevBuf alpha; // npad, valid if have_alpha
    bool have_alpha = false;
    bool have_zc = false; // wz holds z = L^-1 y contiguously (gathered from the factor's y row on
                          // the first posterior after a (re)fit: the row reductions then read one
                          // line per 8 entries instead of one per entry)
    // the single-vector sweeps (trsv.h): x | y, 2 npad doubles, and their captured launch
    // chains -- [0] solve (forward + backward), [1] backward into alpha, [2] forward; the
    // pointers survive a refit, so the graphs do too
    DevBuf vec;
    double *hvec = nullptr; // pinned staging of one vector in / out (npad doubles): a copy from
                            // pageable memory is staged and synchronised by the runtime, which
                            // was half of a single-vector solve's wall time at N = 4096
    double *hio = nullptr;  // pinned staging of a prediction's points in and mean / variance out
    size_t hio_len = 0;
    double *hfit = nullptr; // pinned staging of a (re)fit: [results 8 + 128 | GaussParams | 63 d
                            // border points] -- the hyper-parameter loop's body is three small copies
    hipGraph_t vgraph[3] = {nullptr, nullptr, nullptr};
    hipGraphExec_t vgexec[3] = {nullptr, nullptr, nullptr};
    bool vg_failed[3] = {false, false, false};
    int vg_flow[3] = {0, 0, 0}; // c->trsv_flow when the slot's graph was captured
    ~bq_fit()
    {
        if (hvec)
            (void)hipHostFree(hvec);
        if (hio)
            (void)hipHostFree(hio);
        if (hfit)
            (void)hipHostFree(hfit);
        for (int i = 0; i < 3; ++i) {
            if (vgexec[i])
                (void)hipGraphExecDestroy(vgexec[i]);
            if (vgraph[i])
                (void)hipGraphDestroy(vgraph[i]);
        }
    }
    // false from the start of a (re)factorisation until it has succeeded: a refit that hits a
    // non-positive pivot leaves L, dinv, dw and the scalars overwritten with garbage
    bool valid = false;
    // true after bq_gp_set_y until the next (re)fit: the targets changed, the factor did not
    bool stale = false;
    double logml = 0, logdet = 0, qf = 0;
};

namespace bqh {
// fit.hip: shared with moments.hip
int check_fit(bq_ctx *c, const bq_fit *f);
int fit_dw(bq_ctx *c, bq_fit *f);
int fit_wide(bq_ctx *c, bq_fit *f, WideInv &w);
int fit_vec(bq_ctx *c, bq_fit *f);
int fit_alpha(bq_ctx *c, bq_fit *f);
// Replays a chain of sweep launches over a fit's own buffers from a captured hipGraph (a
// sweep is 2 npad / B launches of 2-8 us each: enqueued one by one the host is the
// bottleneck); eager when graphs are off, under the launch profiler, or if capture fails.
template <class F>
int fit_replay(bq_ctx *c, bq_fit *f, int slot, F &&enqueue)
{
    // (a one-launch sweep is two memsets and a kernel: enqueued directly it costs the host less
    // than a graph launch does -- BQ_FLOW_GRAPH=1 replays it from a graph all the same)
    static const bool flow_graph = std::getenv("BQ_FLOW_GRAPH") && std::atoi(std::getenv("BQ_FLOW_GRAPH"));
    if (!c->use_graph || c->prof || !c->own_stream || c->cur != c->stream ||
        (!flow_graph && trsv_flow_ok(c, f->npad, wide_block(f->npad))))
        return enqueue();
    // a graph captured with the one-launch sweeps inside is not what a fall-back retry (or a
    // caller that switched them off) asks for: such a call is enqueued eagerly
    if (f->vgexec[slot] && f->vg_flow[slot] != c->trsv_flow)
        return enqueue();
    if (!f->vgexec[slot] && !f->vg_failed[slot]) {
        f->vg_failed[slot] = true;
        f->vg_flow[slot] = c->trsv_flow;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
            const int st = enqueue();
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(c->stream, &g);
            if (st == BQ_OK && e == hipSuccess && g &&
                hipGraphInstantiate(&f->vgexec[slot], g, nullptr, nullptr, 0) == hipSuccess) {
                f->vgraph[slot] = g;
                f->vg_failed[slot] = false;
            } else {
                if (g)
                    (void)hipGraphDestroy(g);
                f->vgexec[slot] = nullptr;
                (void)hipGetLastError();
            }
        } else {
            (void)hipGetLastError();
        }
    }
    if (f->vgexec[slot]) {
        HIPCHK(c, hipGraphLaunch(f->vgexec[slot], c->stream));
        return BQ_OK;
    }
    return enqueue();
}
// A one-launch sweep whose hand-off timed out (trsvflow.h: a shared device, a lost slot) has raised
// the context's abort word and every spinner has left: the results of `attempt` are garbage.
// Degrade, do not fail: clear the word, count the event (bq_ctx_stats) and re-issue the SAME
// solve -- `attempt` restages its inputs and ends with the stream synchronised -- on the per-block
// kernels, which compute the same bits (test_flow_sweeps_same_bits_as_per_block_launches).
bool flow_timed_out(bq_ctx *c);
template <class F>
int with_flow_fallback(bq_ctx *c, F &&attempt)
{
    int st = attempt();
    if (st != BQ_OK || !flow_timed_out(c))
        return st;
    ++c->n_flow_fallback;
    const int saved = c->trsv_flow;
    c->trsv_flow = 0;
    st = attempt();
    c->trsv_flow = saved;
    // the retry must not have gone through a one-launch sweep again (a captured graph that still
    // holds one: fit_replay keys its graphs on trsv_flow) -- if the word is up again the results
    // are garbage and the call says so
    if (st == BQ_OK && flow_timed_out(c))
        return fail(c, BQ_ERR_HIP, "a sweep's hand-off timed out again on the per-block kernels");
    return st;
}
// plan.hip: new kernel parameters for every problem of a plan, nothing else re-uploaded
int plan_set_params(bq_ctx *c, bq_plan *p, const double *h, const double *w, const double *s);
// plan.hip: the launch sequence of one pass of a plan (bq_probe_c2_timeline runs it eagerly)
int plan_enqueue(bq_ctx *c, bq_plan *p);
} // namespace bqh
