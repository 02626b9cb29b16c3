// bqhip.hip -- host side of libbqhip.so: the C ABI declared in include/bqhip.h.
//
// Design: a GP "fit + posterior + log-ML" is ONE partial Cholesky
// of a bordered matrix
//
//        [ Kxx + s^2 I                       ]      n (padded to 64 with I)
//    B = [ K(xo,x)        K(xo,xo)           ]      M prediction points
//        [ y^T            0            0     ]      1 row (padded to 64)
//
// Eliminating the first npad columns with the blocked right-looking algorithm
// (potf2 on the 64x64 diagonal, row-per-lane panel solve, MFMA trailing
// update) leaves, in the Schur complement, the posterior covariance of xo,
// -mean(xo) in the y row and -y'Kxx^-1 y in its corner; z = L^-1 y appears in
// the y row of the factored panel.  No separate triangular solve is needed
// for the posterior or the log marginal likelihood; alpha = L^-T z is
// produced on demand by a backward sweep.  Every kernel is batched over
// independent problems (blockIdx.z).
#include "../../include/bqhip.h"
#include "kernels.h"

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

namespace {

// RAII device buffer
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
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
    }
    hipError_t alloc(size_t b)
    {
        release();
        if (b == 0)
            b = 8;
        hipError_t e = hipMalloc(&p, b);
        if (e == hipSuccess)
            bytes = b;
        else
            p = nullptr;
        return e;
    }
    double *d() const { return static_cast<double *>(p); }
    int *i() const { return static_cast<int *>(p); }
};

struct ProfEvent {
    hipEvent_t a, b;
    int cls;
    double work;
};

} // namespace

struct bq_ctx {
    int device = 0;
    hipStream_t stream = nullptr; // main stream: everything is ordered on it
    hipStream_t aux = nullptr;    // high-priority panel stream of the look-ahead Cholesky
    hipStream_t cur = nullptr;    // stream the launch helpers enqueue on (stream or aux)
    hipEvent_t ev_panel = nullptr, ev_next = nullptr, ev_fork = nullptr;
    int lookahead = 1;
    int split_batch = 1; // halves of a mid-sized batch on the two streams (BQ_SPLIT=0: lock-step)
    int la_min = 4096;   // look-ahead only while the bulk update has at least this many rows (BQ_LA_MIN)
    DevBuf panel_ws;     // scratch panel columns of the eager linalg entry points
    DevBuf scratch;      // per-call temporaries of the acquisition / moment entry points, kept
                         // between calls (hipFree synchronises the device); bq_ctx_trim frees it
    int gemm_lds = 1;    // LDS-staged 128x128 trailing update (BQ_GEMM_LDS)
    int use_graph = 1;   // replay plans from a captured hipGraph (BQ_GRAPH=0 disables)
    bool own_stream = false;
    int cus = 256;
    int nb_override = 0;
    bq_plan *plan_cache = nullptr; // workspace of the last batched call, kept for the next one
    char err[512] = {0};
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool prof = false;
    std::vector<ProfEvent> prof_events;
    double prof_ms[BQ_K_NCLASS] = {0};
    int64_t prof_n[BQ_K_NCLASS] = {0};
    double prof_work[BQ_K_NCLASS] = {0};
    DevBuf gbuf;   // GaussParams of the single-problem entry points (cached)
    GaussParams gbuf_host{};
    bool gbuf_valid = false;
    DevBuf dinv64; // potf2 reciprocal-diagonal scratch
    long long *stamp_buf = nullptr; // bq_probe_c2_timeline: 160 stamps per slab step
};

namespace {

int fail(bq_ctx *c, int code, const char *fmt, ...)
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
            return fail((c), e__ == hipErrorOutOfMemory ? BQ_ERR_NOMEM : BQ_ERR_HIP,           \
                        "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__,      \
                        __LINE__);                                                             \
    } while (0)

#define BQCHK(call)                                                                            \
    do {                                                                                       \
        int s__ = (call);                                                                      \
        if (s__ != BQ_OK)                                                                      \
            return s__;                                                                        \
    } while (0)

inline long roundup(long v, long q) { return (v + q - 1) / q * q; }

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

GaussParams make_params(int d, double h, const double *w, double s)
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

// ---- profiling brackets -------------------------------------------------
struct Bracket {
    bq_ctx *c;
    ProfEvent ev;
    bool on;
    Bracket(bq_ctx *ctx, int cls, double work = 0.0) : c(ctx), on(ctx->prof)
    {
        if (on) {
            ev.cls = cls;
            ev.work = work;
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

int prof_collect(bq_ctx *c)
{
    if (c->prof_events.empty())
        return BQ_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->aux)
        HIPCHK(c, hipStreamSynchronize(c->aux));
    for (auto &e : c->prof_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            c->prof_ms[e.cls] += ms;
            c->prof_n[e.cls] += 1;
            c->prof_work[e.cls] += e.work;
        }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    c->prof_events.clear();
    return BQ_OK;
}

// ---- launch helpers -------------------------------------------------------
template <int D>
void launch_gram_sym_d(bq_ctx *c, const double *x, long xstride, const GaussParams *gp,
                       int gpstride, double *K, long ldk, long kstride, int n, int batch)
{
    // whole 64 x 64 blocks: every exp once, block and transpose stored (N = 4096: 20 us against
    // 27 us with the full sweep below, which stays for ragged sizes)
    if ((n % 64) == 0 && (ldk % 2) == 0) {
        const int T = n / 64;
        hipLaunchKernelGGL(gram_tri_kernel<D>, dim3(T * (T + 1) / 2, 1, batch), dim3(256), 0, c->cur,
                           x, xstride, gp, gpstride, K, ldk, kstride, n);
        return;
    }
    dim3 grid((n + 127) / 128, (n + 63) / 64, batch);
    hipLaunchKernelGGL(gram_sym_kernel<D>, grid, dim3(256), 0, c->cur, x, xstride, gp, gpstride,
                       K, ldk, kstride, n);
}

int launch_gram_sym(bq_ctx *c, int d, const double *x, long xstride, const GaussParams *gp,
                    int gpstride, double *K, long ldk, long kstride, int n, int batch)
{
    Bracket br(c, BQ_K_GRAM, (8.0 * n * n + 8.0 * d * n) * batch);
    switch (d) {
    case 1: launch_gram_sym_d<1>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 2: launch_gram_sym_d<2>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 3: launch_gram_sym_d<3>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 4: launch_gram_sym_d<4>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 5: launch_gram_sym_d<5>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 6: launch_gram_sym_d<6>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 7: launch_gram_sym_d<7>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    case 8: launch_gram_sym_d<8>(c, x, xstride, gp, gpstride, K, ldk, kstride, n, batch); break;
    default: return fail(c, BQ_ERR_BAD_ARG, "d must be in 1..%d", BQ_MAXD);
    }
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

template <int D>
void launch_gram_cross_d(bq_ctx *c, const double *x1, int n1, const double *x2, int n2,
                         const GaussParams &g, double *K, long ldk)
{
    dim3 grid((n1 + 63) / 64, (n2 + 63) / 64, 1);
    hipLaunchKernelGGL(gram_cross_kernel<D>, grid, dim3(256), 0, c->cur, x1, n1, x2, n2, g, K,
                       ldk);
}

int launch_gram_cross(bq_ctx *c, int d, const double *x1, int n1, const double *x2, int n2,
                      const GaussParams &g, double *K, long ldk)
{
    if (n1 <= 0 || n2 <= 0)
        return BQ_OK;
    Bracket br(c, BQ_K_GRAM, 8.0 * n1 * n2);
    switch (d) {
    case 1: launch_gram_cross_d<1>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 2: launch_gram_cross_d<2>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 3: launch_gram_cross_d<3>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 4: launch_gram_cross_d<4>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 5: launch_gram_cross_d<5>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 6: launch_gram_cross_d<6>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 7: launch_gram_cross_d<7>(c, x1, n1, x2, n2, g, K, ldk); break;
    case 8: launch_gram_cross_d<8>(c, x1, n1, x2, n2, g, K, ldk); break;
    default: return fail(c, BQ_ERR_BAD_ARG, "d must be in 1..%d", BQ_MAXD);
    }
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// fs: when set, the first launch of the slab sweep rides in the assembly (assemble_first_kernel)
struct FirstStep {
    double *S0 = nullptr;
    long lds = 0, sstride = 0;
    double *dinv = nullptr;
    int *info = nullptr;
};

template <int D>
void launch_assemble_d(bq_ctx *c, const double *pts, long pstride, const double *y, long ystride,
                       const GaussParams *gp, int gpstride, double *A, long lda, long astride,
                       Layout L, int batch, const FirstStep &fs)
{
    dim3 grid((L.ntot + 127) / 128, (L.ntot + 63) / 64, batch);
    if (fs.S0)
        hipLaunchKernelGGL(assemble_first_kernel<D>, grid, dim3(256), 0, c->cur, pts, pstride, y,
                           ystride, gp, gpstride, A, lda, astride, L, fs.S0, fs.lds, fs.sstride,
                           fs.dinv, (long)BQ_DINV_STRIDE, fs.info);
    else
        hipLaunchKernelGGL(assemble_kernel<D>, grid, dim3(256), 0, c->cur, pts, pstride, y, ystride,
                           gp, gpstride, A, lda, astride, L);
}

int launch_assemble(bq_ctx *c, int d, const double *pts, long pstride, const double *y,
                    long ystride, const GaussParams *gp, int gpstride, double *A, long lda,
                    long astride, Layout L, int batch, const FirstStep &fs = FirstStep())
{
    Bracket br(c, BQ_K_GRAM, 8.0 * L.ntot * (L.ntot + 1.0) / 2.0 * batch);
    switch (d) {
    case 1: launch_assemble_d<1>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs); break;
    case 2: launch_assemble_d<2>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs); break;
    case 3: launch_assemble_d<3>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs); break;
    case 4: launch_assemble_d<4>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs); break;
    case 5: launch_assemble_d<5>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs); break;
    case 6: launch_assemble_d<6>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs); break;
    case 7: launch_assemble_d<7>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs); break;
    case 8: launch_assemble_d<8>(c, pts, pstride, y, ystride, gp, gpstride, A, lda, astride, L, batch, fs); break;
    default: return fail(c, BQ_ERR_BAD_ARG, "d must be in 1..%d", BQ_MAXD);
    }
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// C(m x n) -= P(m x k) Q(n x k)^T; tile shape from the amount of parallelism
// fuse_j0 >= 0: also factor the leading 64x64 block of C (global column fuse_j0) in the
// same launch (see gemm_sub_kernel); dinv / info as for launch_potf2
// whether C(m x n) -= P Q^T with unit-stride Q rows goes to the LDS-staged 128 x 128 kernel:
// whole 64 x 64 wave tiles, k in chunks of 32, and at least BQ_LDS_MIN_TILES workgroup tiles.
// (Round 1 asked for a full chip of tiles, 256.  The look-ahead's update of the next panel --
// m x 512 columns, 100-250 tiles, on the second stream BESIDE the bulk update -- then went to
// the register-streaming kernel at ~12 TFLOP/s and sat on the panel chain: with the LDS
// kernel N = 16384 takes 26.5 instead of 27.1 ms, 12288 13.17 instead of 13.35; the smaller
// sizes and the batched configs do not move.)
// Such an update never carries the fused diagonal factor (the factor would ride on the
// register-streaming kernel, which is slower by more than a potf2 launch costs).
#define BQ_LDS_MIN_TILES 96
static bool gemm_uses_lds(const bq_ctx *c, int m, int n, int k, int lower, int batch)
{
    long a = (long)((m + 127) / 128) * ((n + 127) / 128) * batch;
    if (lower)
        a = a / 2 + 1;
    return c->gemm_lds && a >= BQ_LDS_MIN_TILES && n >= 128 && (m % 64) == 0 &&
           (n % 64) == 0 && (k % 32) == 0;
}

int launch_gemm(bq_ctx *c, int cls, double *C, long ldc, long cstride, const double *P, long ldp,
                long pstride, const double *Q, long qsj, long qsk, long qstride, int m, int n,
                int k, int lower, int batch, int fuse_j0 = -1, double *dinv = nullptr,
                long dstride = 0, int *info = nullptr, int ccut = 0)
{
    // ccut > 0: columns >= ccut of C need no update (honoured by the LDS-staged kernel only)
    if (m <= 0 || n <= 0 || k <= 0)
        return BQ_OK;
    if ((m & 15) || (n & 15) || (k & 7))
        return fail(c, BQ_ERR_BAD_ARG, "gemm: m,n must be multiples of 16 and k of 8");
    auto tiles = [&](int t) {
        long a = (long)((m + t - 1) / t) * ((n + t - 1) / t) * batch;
        return lower ? a / 2 + 1 : a;
    };
    // algorithmic flops: full product 2mnk; lower trapezoid of a trailing block
    // 2k(mn - n^2/2), i.e. m^2 k for the square update
    const double flops = (lower ? 2.0 * k * ((double)m * n - 0.5 * (double)n * n)
                                : 2.0 * (double)m * n * k) * batch;
    if (cls == BQ_K_SYRK && tiles(128) < c->cus)
        cls = BQ_K_SYRK_SMALL;
    Bracket br(c, cls, flops);
    const long cu = c->cus;
    // square trailing updates launch only their lower workgroup tiles (mode 2)
    const bool tri = lower && m == n;
    const int mode = tri ? 2 : lower;
    auto grid_for = [&](int t) {
        const unsigned gm = (unsigned)((m + t - 1) / t), gn = (unsigned)((n + t - 1) / t);
        return tri ? dim3(gm * (gm + 1) / 2, 1, batch) : dim3(gm, gn, batch);
    };
    // the 4x4x4 four-block MFMA sustains ~1.5x the rate of the 16x16x4 form on gfx950; it
    // needs unit-stride Q rows and whole wave tiles (every padded system here has them)
    const bool f444 = qsj == 1 && (m % 64) == 0 && (n % 64) == 0;
#define BQ_GEMM_SUB(TM_, TN_, T_)                                                                  \
    do {                                                                                           \
        if (f444)                                                                                  \
            hipLaunchKernelGGL((gemm_sub_kernel<TM_, TN_, 1>), grid_for(T_), dim3(256), 0, c->cur, \
                               C, ldc, cstride, P, ldp, pstride, Q, qsj, qsk, qstride, m, n, k,    \
                               mode, fuse_j0, dinv, dstride, info);                                \
        else                                                                                       \
            hipLaunchKernelGGL((gemm_sub_kernel<TM_, TN_, 0>), grid_for(T_), dim3(256), 0, c->cur, \
                               C, ldc, cstride, P, ldp, pstride, Q, qsj, qsk, qstride, m, n, k,    \
                               mode, fuse_j0, dinv, dstride, info);                                \
    } while (0)
    // a 64-column slab has no use for 128-column workgroup tiles (half of their waves idle)
    if (f444 && fuse_j0 < 0 && n >= 128 && gemm_uses_lds(c, m, n, k, lower, batch)) {
        dim3 g = grid_for(128);
        hipLaunchKernelGGL(gemm_lds_kernel, g, dim3(256), BQ_LDS_BYTES, c->cur, C, ldc, cstride, P,
                           ldp, pstride, Q, qsk, qstride, m, n, k, mode,
                           ccut > 0 ? ccut : 0x7fffffff);
    } else if (tiles(128) >= cu && n >= 128) {
        BQ_GEMM_SUB(4, 4, 128);
    } else if (tiles(64) >= cu / 2) {
        if (k == 64)
            hipLaunchKernelGGL((gemm_k64_kernel<2, 2>), grid_for(64), dim3(256), 0, c->cur, C, ldc,
                               cstride, P, ldp, pstride, Q, qsj, qsk, qstride, m, n, mode, fuse_j0,
                               dinv, dstride, info);
        else
            BQ_GEMM_SUB(2, 2, 64);
    } else {
        if (k == 64)
            hipLaunchKernelGGL((gemm_k64_kernel<1, 1>), grid_for(32), dim3(256), 0, c->cur, C, ldc,
                               cstride, P, ldp, pstride, Q, qsj, qsk, qstride, m, n, mode, fuse_j0,
                               dinv, dstride, info);
        else
            BQ_GEMM_SUB(1, 1, 32);
    }
#undef BQ_GEMM_SUB
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

int launch_potf2(bq_ctx *c, double *A, long lda, long astride, int j0, double *dinv, long dstride,
                 int *info, int batch)
{
    Bracket br(c, BQ_K_POTF2, 64.0 * 64 * 64 / 3.0 * batch);
    hipLaunchKernelGGL(potf2_kernel, dim3(1, 1, batch), dim3(256), 0, c->cur, A, lda, astride,
                           j0, dinv, dstride, info);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// the MFMA panel solve (trsm_blk_kernel): needs the block inverses potf2f_body leaves
// behind the 64 reciprocal pivots
int launch_trsm_blk(bq_ctx *c, double *X, long ldx, long xstride, int m, const double *L11,
                    long ldl, long lstride, const double *dinv, long dstride, int batch)
{
    if (m <= 0)
        return BQ_OK;
    Bracket br(c, BQ_K_TRSM, 64.0 * 64 * (double)m * batch);
    hipLaunchKernelGGL(trsm_blk_kernel, dim3((m + 63) / 64, 1, batch), dim3(256), 0, c->cur, X, ldx,
                       xstride, m, L11, ldl, lstride, dinv, dstride);
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

// below this size one or two matrices sweep with the one-launch steps of outer block 64
// (tools/potrf_sizes.py, ms with blocks 64 / 128 / 256: N = 2048 0.57 / 0.76 / 0.80,
// 4096 1.69 / 1.93 / 1.92, 6144 4.21 / 3.72 / 3.57)
#define BQ_SLAB_MAX 4800

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
        // the shortest chain until the k = 64 updates cost more than it saves
        // (tools/potrf_sizes.py on one matrix: N=2048 0.73 / 0.81 ms, 3072 1.24 / 1.31,
        // 4096 2.03 / 2.00, 6144 4.37 / 3.83 with blocks 64 / 128; 8192 6.49 / 6.44 with 128 / 256)
        if (ntot < BQ_SLAB_MAX)
            return 64;
        // a wider block halves the trailing update's C traffic per flop (60 instead of 56
        // TFLOP/s at k = 512); it pays once the panel it lengthens hides behind the bulk
        // update (N = 8192: 5.43 / 5.56 ms with 256 / 512, 12288: 13.55 / 13.11, 16384: 28.0 / 26.9)
        return ntot < 12000 ? 256 : 512;
    }
    // (batches of mid-sized matrices, recursive panels: C5 shard 6.55 / 6.42 / 6.70 / 6.57 ms
    // with 256 / 320 / 384 / 512; 256 x C2 5.72 / 5.47 / 5.64 ms with 256 / 320 / 448)
    if (ntot >= 1024 && mb >= 100.0)
        return 320;
    if (ntot >= 512 && mb >= 30.0)
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
int enqueue_panel_rec(bq_ctx *c, double *A, long lda, long astride, int batch, int ntot, int j0,
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
int enqueue_panel(bq_ctx *c, double *A, long lda, long astride, int batch, int ntot, int K0,
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
            Bracket br(c, BQ_K_GEMM,
                       (m * 64.0 * 64.0 + (has_next ? 2.0 * m * 64.0 * 64.0 * (sidx + 1) : 0.0)) *
                           batch);
            const int par = sidx & 1;
            hipLaunchKernelGGL(panel_step_kernel, dim3(nrb, 1, batch), dim3(256), 0, c->cur, A,
                               lda, astride, S[par], S[par ^ 1], (long)ntot, sstride, K0, j0,
                               dinv + par * BQ_DINV_HALF, dinv + (par ^ 1) * BQ_DINV_HALF,
                               (long)BQ_DINV_STRIDE, has_next, sidx == 0, SL, info);
            HIPCHK(c, hipGetLastError());
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
int enqueue_slab_sweep(bq_ctx *c, double *A, long lda, long astride, int batch, int ntot,
                       int ncols, double *dinv, int *info, double *ws, int col0 = 0,
                       bool first_done = false)
{
    if (ntot <= 64)
        return launch_potf2(c, A - col0 - (long)col0 * lda, lda, astride, col0, dinv,
                            BQ_DINV_STRIDE, info, batch);
    const long sstride = 64L * ntot;
    double *S[2] = {ws, ws + sstride * batch};
    if (!first_done) {
        // the first diagonal factor and the staging of panel 0 share a launch (or ride in the
        // assembly: assemble_first_kernel)
        Bracket br(c, BQ_K_POTF2, 64.0 * 64 * 64 / 3.0 * batch);
        hipLaunchKernelGGL(slab_first_kernel, dim3(ntot / 64, 1, batch), dim3(256), 0, c->cur, A,
                           lda, astride, S[0], (long)ntot, sstride, ntot, dinv,
                           (long)BQ_DINV_STRIDE, info, col0);
        HIPCHK(c, hipGetLastError());
    }
    for (int j0 = 0, par = 0; j0 < ncols; j0 += 64, par ^= 1) {
        const int r0 = j0 + 64;
        if (r0 >= ntot)
            break;
        const int T = (ntot - r0) / 64;
        const int fnext = r0 < ncols;
        const double m = (double)(ntot - r0);
        // the tile updates (lower half of 2 m^2 64) and the solve of the panel (m 64^2)
        Bracket br(c, BQ_K_SYRK_SMALL, (m * m * 64.0 + m * 64.0 * 64.0) * batch);
        if (c->stamp_buf)
            hipLaunchKernelGGL(slab_step_kernel<true>, dim3(T * (T + 1) / 2, 1, batch), dim3(256),
                               0, c->cur, A, lda, astride, S[par], S[par ^ 1], (long)ntot, sstride,
                               ntot, j0, dinv + par * BQ_DINV_HALF,
                               dinv + (par ^ 1) * BQ_DINV_HALF, (long)BQ_DINV_STRIDE, fnext,
                               !fnext, info, col0, c->stamp_buf + 160 * (j0 / 64));
        else
            hipLaunchKernelGGL(slab_step_kernel<false>, dim3(T * (T + 1) / 2, 1, batch), dim3(256),
                               0, c->cur, A, lda, astride, S[par], S[par ^ 1], (long)ntot, sstride,
                               ntot, j0, dinv + par * BQ_DINV_HALF,
                               dinv + (par ^ 1) * BQ_DINV_HALF, (long)BQ_DINV_STRIDE, fnext,
                               !fnext, info, col0, (long long *)nullptr);
        HIPCHK(c, hipGetLastError());
    }
    return BQ_OK;
}

// nb_forced: the outer block of the whole batch when this call factors one half of it
// whether a factorisation of these sizes goes to the one-launch slab sweep from its first column
bool sweep_is_slab(const bq_ctx *c, int ntot, int ncols, int batch, size_t panel_ws_len)
{
    return auto_nb(c, ntot, batch) == 64 && panel_ws_len >= panel_ws_doubles(ntot, batch) &&
           ncols >= 64 && ntot > 64;
}

// skip_border: the caller reads its results off the border ROWS (plan_readout_kernel), so the
// trailing updates leave the border x border block alone
int enqueue_potrf_group(bq_ctx *c, double *A, long lda, long astride, int batch, int ntot,
                        int ncols, double *dinv, int *info, double *panel_ws, size_t panel_ws_len,
                        int nb_forced, bool first_done = false, bool skip_border = false)
{
    const int NB = nb_forced > 0 ? nb_forced : auto_nb(c, ntot, batch);
    double *ws = (panel_ws && panel_ws_len >= panel_ws_doubles(ntot, batch)) ? panel_ws : nullptr;
    if (NB == 64 && ws && ncols >= 64)
        return enqueue_slab_sweep(c, A, lda, astride, batch, ntot, ncols, dinv, info, ws, 0,
                                  first_done);
    const bool la = c->lookahead && c->aux && NB >= 128 && ncols > NB;
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
        const bool to_slab = batch <= 2 && ws && NB > 64 && r0 < ncols && ntot - r0 < BQ_SLAB_MAX;
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

// Eliminate the first ncols columns of `batch` matrices.  A batch of mid-sized matrices
// (config C5: 64 x N = 2048) sweeps in lock-step: every 64-column panel step is two short
// dependent launches that leave most of the chip idle, and a third of the sweep's time is
// such panel work.  The batch is therefore cut in two halves on the two streams: one half's
// panel chain runs beside the other half's MFMA trailing update (C5 shard: 7.21 -> 6.78 ms;
// three or four groups on more streams were slower, 7.8 / 7.5 ms).  (Large single matrices
// use the second stream for the look-ahead instead, small ones the one-launch steps.)
int enqueue_potrf_partial(bq_ctx *c, double *A, long lda, long astride, int batch, int ntot,
                          int ncols, double *dinv, int *info, double *panel_ws = nullptr,
                          size_t panel_ws_len = 0, bool first_done = false,
                          bool skip_border = false)
{
    if ((ntot & 63) || (ncols & 63) || ncols > ntot)
        return fail(c, BQ_ERR_BAD_ARG, "potrf: sizes must be multiples of 64");
    const int NB = auto_nb(c, ntot, batch);
    const bool la = c->lookahead && c->aux && NB >= 128 && ncols > NB &&
                    ntot - std::min(NB, ncols) >= c->la_min;
    if (c->split_batch && c->lookahead && c->aux && c->cur == c->stream && batch >= 8 &&
        NB >= 128 && !la) {
        const int b0 = batch / 2, b1 = batch - b0;
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

// ---------------------------------------------------------------------------
// Row sweeps over a RESIDENT factor (cho_solve, alpha, posterior variance, the bordered
// acquisition update): X <- X L^-T (forward) and X <- X L^-1 (backward), right-hand sides as
// the rows of X (mrows x npad, ld ldx).
//
// With 64-column steps a sweep is 2 npad / 64 dependent launches and nothing else -- 2 ms at
// N = 4096 for 134 MB of factor.  The steps are therefore B = 256 or 512 columns wide and a
// step's triangular solve is a product with the explicit inverse of its B x B diagonal block
// (MAGMA's trtri-based trsm; the 16 x 16 inverses of the panel solve one level up):
//     forward   Y_J = X_J W_J^T,   X[:, J+B:] -= Y_J L[J+B:, J]^T
//     backward  Y_J = X_J W_J,     X[:, :J]   -= Y_J L[J, :J]
// -- two MFMA GEMM launches per B columns, input and output in two buffers so that no step
// copies.  NR = -W^T of every diagonal block (npad x B doubles, block J at NR + J B, ld B) is
// built once per factor by the 64-column sweep itself applied to -I, batched over the blocks.
// cond(W_J) <= cond(L) = sqrt(cond(K)): 1e-12 relative at the worst-conditioned configs.
// ---------------------------------------------------------------------------
struct WideInv {
    const double *nr = nullptr; // -W^T of every block
    // the single-vector sweeps (trsv.h):
    const double *nt = nullptr; // -W (the transposes)
    const double *tt = nullptr; // T_J^T, T_J = W_J L[J, J-B] (blocks J >= B)
    const double *uu = nullptr; // U_J = L[J+B, J] W_J (all blocks but the last)
    const double *t = nullptr;  // T_J itself (rows of T contiguous: the fused row-sweep step)
    int B = 0;
};

// the sweeps' products (few rows, a long k).  A kernel of their own with eight k-steps of
// fragment loads in flight (instead of gemm_sub_kernel's one) was measured and gained nothing:
// 36 us per launch at k = 512 either way -- the factor panel streams from HBM behind one
// block of prefetch, not from L2.
int launch_gemm_rows(bq_ctx *c, int cls, double *C, long ldc, const double *P, long ldp,
                     const double *Q, long qsj, long qsk, int m, int n, int k)
{
    // small products (posterior variance at C2 size): split-k tiles, gemm_splitk_kernel
    if ((m % 32) == 0 && (n % 32) == 0 && (k % 64) == 0 && k <= 2048 &&
        (long)(m / 32) * (n / 32) <= 4L * c->cus) {
        Bracket br(c, cls, 2.0 * (double)m * n * k);
        hipLaunchKernelGGL(gemm_splitk_kernel, dim3(m / 32, n / 32), dim3(256), 0, c->cur, C, ldc,
                           P, ldp, Q, qsj, qsk, k);
        HIPCHK(c, hipGetLastError());
        return BQ_OK;
    }
    return launch_gemm(c, cls, C, ldc, 0, P, ldp, 0, Q, qsj, qsk, 0, m, n, k, 0, 1);
}

inline int wide_block(int npad) { return npad < 2048 ? std::min(npad, 256) : 512; }
inline size_t wide_doubles(int npad) { return (size_t)npad * wide_block(npad); }
// NR, NT, TT, UU and the scratch of T before its transposition (a full B x B per block)
inline size_t wide_alloc_doubles(int npad)
{
    const size_t B = (size_t)wide_block(npad);
    return 5 * wide_doubles(npad) + B * B;
}
inline WideInv wide_views(const double *base, int npad)
{
    WideInv w;
    const size_t n = wide_doubles(npad);
    w.nr = base;
    w.nt = base + n;
    w.tt = base + 2 * n;
    w.uu = base + 3 * n;
    w.t = base + 4 * n;
    w.B = wide_block(npad);
    return w;
}

// dw: the per-64-block records of diag_winv_kernel (npad / 64 of them)
int compute_wide_inverses(bq_ctx *c, const double *L, long ldl, int npad, const double *dw,
                          double *nr)
{
    const int B = wide_block(npad);
    HIPCHK(c, hipMemsetAsync(nr, 0, sizeof(double) * wide_doubles(npad), c->stream));
    hipLaunchKernelGGL(neg_identity_kernel, dim3((npad + 255) / 256), dim3(256), 0, c->stream, nr,
                       B, npad);
    HIPCHK(c, hipGetLastError());
    const int nfull = npad / B, rem = npad - nfull * B;
    for (int part = 0; part < 2; ++part) {
        const int batch = part == 0 ? nfull : (rem ? 1 : 0), bs = part == 0 ? B : rem;
        const int J0 = part == 0 ? 0 : nfull * B;
        if (batch == 0)
            continue;
        double *X = nr + (size_t)J0 * B; // bs x bs per block, ld B
        const double *Ld = L + J0 + (long)J0 * ldl;
        const long xs = (long)B * B, ls = (long)B * (1 + ldl), ds = (long)(B / 64) * BQ_DINV_HALF;
        for (int jb = 0; jb < bs; jb += 64) {
            const double *L11 = Ld + jb + (long)jb * ldl;
            BQCHK(launch_trsm_blk(c, X + (long)jb * B, B, xs, bs, L11, ldl, ls,
                                  dw + (long)((J0 + jb) / 64) * BQ_DINV_HALF, ds, batch));
            const int rest = bs - jb - 64;
            if (rest > 0)
                BQCHK(launch_gemm(c, BQ_K_GEMM, X + (long)(jb + 64) * B, B, xs, X + (long)jb * B, B,
                                  xs, L11 + 64, 1, ldl, ls, bs, rest, 64, 0, batch));
        }
    }
    // NT = the blocks' transposes, behind NR
    double *nt = nr + wide_doubles(npad), *tt = nt + wide_doubles(npad),
           *uu = tt + wide_doubles(npad);
    for (int part = 0; part < 2; ++part) {
        const int batch = part == 0 ? nfull : (rem ? 1 : 0), bs = part == 0 ? B : rem;
        if (batch == 0)
            continue;
        const size_t off = part == 0 ? 0 : (size_t)nfull * B * B;
        hipLaunchKernelGGL(transpose_blocks_kernel, dim3(bs / 64, bs / 64, batch), dim3(256), 0,
                           c->stream, nr + off, nt + off, B, (long)B * B);
        HIPCHK(c, hipGetLastError());
    }
    // The couplings of neighbouring blocks for the one-launch steps of trsv.h:
    //   T_J = W_J L[J, J-B]  (bJ x B; kept transposed)   and   U_J = L[J+B, J] W_J  (bn x B).
    const int nblk = nfull + (rem ? 1 : 0);
    if (nblk > 1) {
        // T before its transposition: a full B x B per block, behind UU
        double *tmp = uu + wide_doubles(npad);
        HIPCHK(c, hipMemsetAsync(tmp, 0, sizeof(double) * (size_t)nblk * B * B, c->stream));
        HIPCHK(c, hipMemsetAsync(uu, 0, sizeof(double) * wide_doubles(npad), c->stream));
        const long bb = (long)B * B, ls = (long)B * (1 + ldl);
        // full blocks J = B .. (nfull - 1) B, then the partial last one
        if (nfull > 1)
            BQCHK(launch_gemm(c, BQ_K_GEMM, tmp + bb, B, bb, nt + bb, B, bb, L + B, ldl, 1, ls, B, B,
                              B, 0, nfull - 1));
        if (rem) {
            const long J = (long)nfull * B;
            BQCHK(launch_gemm(c, BQ_K_GEMM, tmp + nfull * bb, B, 0, nt + J * B, B, 0,
                              L + J + (J - B) * ldl, ldl, 1, 0, rem, B, rem, 0, 1));
        }
        for (int part = 0; part < 2; ++part) {
            const int batch = part == 0 ? nfull - 1 : (rem ? 1 : 0), bs = part == 0 ? B : rem;
            if (batch <= 0)
                continue;
            const size_t off = (part == 0 ? 1 : (size_t)nfull) * bb;
            hipLaunchKernelGGL(transpose_blocks_kernel, dim3(bs / 64, B / 64, batch), dim3(256), 0,
                               c->stream, tmp + off, tt + off, B, bb);
            HIPCHK(c, hipGetLastError());
        }
        // U_J for the blocks with a full neighbour below, then the one above the partial block
        if (nfull > 1)
            BQCHK(launch_gemm(c, BQ_K_GEMM, uu, B, bb, L + B, ldl, ls, nr, 1, B, bb, B, B, B, 0,
                              nfull - 1));
        if (rem) {
            const long J = (long)(nfull - 1) * B;
            BQCHK(launch_gemm(c, BQ_K_GEMM, uu + J * B, B, 0, L + J + B + J * ldl, ldl, 0,
                              nr + J * B, 1, B, 0, rem, B, B, 0, 1));
        }
    }
    return BQ_OK;
}

// One right-hand side: x (npad, consumed) -> y = L^-1 x, one launch per B columns (trsv.h)
int enqueue_forward_vec(bq_ctx *c, double *x, double *y, const double *L, long ldl, int npad,
                        WideInv w)
{
    for (int J = 0; J < npad; J += w.B) {
        const int bJ = std::min(w.B, npad - J);
        const int nupd = J > 0 ? (npad - J - bJ) / 64 : 0;
        Bracket br(c, BQ_K_GEMM, (double)bJ * bJ + 2.0 * w.B * (J > 0 ? bJ + 64.0 * nupd : 0));
#define BQ_TRSV_FWD(NB_)                                                                           \
    hipLaunchKernelGGL((trsv_fwd_step_kernel<NB_>), dim3(bJ / 16 + nupd), dim3(1024), 0, c->cur,  \
                       L, ldl, J, bJ, w.B, w.nr + (size_t)J * w.B, w.tt + (size_t)J * w.B, x, y)
        if (w.B == 512)
            BQ_TRSV_FWD(8);
        else if (w.B == 256)
            BQ_TRSV_FWD(4);
        else
            BQ_TRSV_FWD(0);
#undef BQ_TRSV_FWD
        HIPCHK(c, hipGetLastError());
    }
    return BQ_OK;
}

// x (npad, consumed) -> y = L^-T x
int enqueue_backward_vec(bq_ctx *c, double *x, double *y, const double *L, long ldl, int npad,
                         WideInv w)
{
    const int last = (npad - 1) / w.B * w.B;
    for (int J = last; J >= 0; J -= w.B) {
        const int bJ = std::min(w.B, npad - J);
        const int bn = J < last ? std::min(w.B, npad - J - w.B) : 0;
        const int nupd = bn > 0 ? J / 64 : 0;
        Bracket br(c, BQ_K_GEMM, (double)bJ * bJ + 2.0 * bn * (bJ + 64.0 * nupd));
        hipLaunchKernelGGL(trsv_bwd_step_kernel, dim3(bJ / 16 + nupd), dim3(1024), 0, c->cur, L, ldl,
                           J, bJ, w.B, bn, w.nt + (size_t)J * w.B, w.uu + (size_t)J * w.B, x, y);
        HIPCHK(c, hipGetLastError());
    }
    return BQ_OK;
}

// X <- X L^-T in place, 64 columns per step from the 16 x 16 block inverses (the panel solve
// of the factorisation): twice the launches of the wide steps, but every product is with the
// inverse of a 16 x 16 block only.  For right-hand sides that nearly lie in the span of the
// factor's own columns -- the borders of the acquisition update, whose Schur complement
// k0 - |L^-1 k|^2 cancels to 1e-7 -- the 64- to 512-wide explicit inverses lose cond(L_JJ)
// (3e-10 against 2e-12 on test_acquisition_and_posterior_vs_extended_precision).
int enqueue_forward_rows_blk(bq_ctx *c, double *X, long ldx, int mrows, const double *L, long ldl,
                             int npad, const double *dw)
{
    for (int jb = 0; jb < npad; jb += 64) {
        const double *L11 = L + jb + (long)jb * ldl;
        BQCHK(launch_trsm_blk(c, X + (long)jb * ldx, ldx, 0, mrows, L11, ldl, 0,
                              dw + (long)(jb / 64) * BQ_DINV_HALF, 0, 1));
        const int rest = npad - jb - 64;
        if (rest > 0)
            BQCHK(launch_gemm(c, BQ_K_GEMM, X + (long)(jb + 64) * ldx, ldx, 0, X + (long)jb * ldx,
                              ldx, 0, L11 + 64, 1, ldl, 0, mrows, rest, 64, 0, 1));
    }
    return BQ_OK;
}

// Xout <- Xin L^-T; Xin is overwritten with partial sums
int enqueue_forward_rows(bq_ctx *c, double *Xin, double *Xout, long ldx, int mrows,
                         const double *L, long ldl, int npad, WideInv w)
{
    // small systems: one launch per step (rows_step_kernel), every entry of Xout written
    if ((mrows % 32) == 0 && (w.B % 64) == 0 &&
        (long)(mrows / 32) * (npad / 32) <= 4L * c->cus) {
        for (int J = 0; J < npad; J += w.B) {
            const int bJ = std::min(w.B, npad - J), rest = npad - J - bJ;
            RowsJob a{}, b{};
            a.C = Xout + (long)J * ldx;
            a.ldc = ldx;
            a.P1 = Xin + (long)J * ldx;
            a.ldp1 = ldx;
            a.Q1 = w.nt + (size_t)J * w.B;
            a.qsj1 = 1;
            a.qsk1 = w.B;
            a.k1 = bJ;
            a.ny = bJ / 32;
            a.write = 1;
            // (unused operand pairs point at valid memory: k2 = 0 never dereferences them)
            a.P2 = a.P1, a.Q2 = a.Q1, a.ldp2 = ldx, a.qsj2 = 1, a.qsk2 = w.B;
            b = a;
            b.ny = 0;
            if (J > 0) {
                a.P2 = Xout + (long)(J - w.B) * ldx;
                a.Q2 = w.t + (size_t)J * w.B;
                a.k2 = w.B;
                if (rest > 0) {
                    b.C = Xin + (long)(J + bJ) * ldx;
                    b.P1 = a.P2;
                    b.Q1 = L + J + bJ + (long)(J - w.B) * ldl;
                    b.qsj1 = 1;
                    b.qsk1 = ldl;
                    b.k1 = w.B;
                    b.k2 = 0;
                    b.ny = rest / 32;
                    b.write = 0;
                }
            }
            Bracket br(c, BQ_K_GEMM,
                       2.0 * mrows * ((double)bJ * (a.k1 + a.k2) + (double)b.ny * 32 * b.k1));
            hipLaunchKernelGGL(rows_step_kernel, dim3(mrows / 32, a.ny + b.ny), dim3(256), 0, c->cur,
                               a, b);
            HIPCHK(c, hipGetLastError());
        }
        return BQ_OK;
    }
    HIPCHK(c, hipMemsetAsync(Xout, 0, sizeof(double) * (size_t)ldx * npad, c->cur));
    for (int J = 0; J < npad; J += w.B) {
        const int bJ = std::min(w.B, npad - J);
        // (NR[k, j] read through its transposed copy: unit stride across the output columns)
        BQCHK(launch_gemm_rows(c, BQ_K_TRSM, Xout + (long)J * ldx, ldx, Xin + (long)J * ldx, ldx,
                               w.nt + (size_t)J * w.B, 1, w.B, mrows, bJ, bJ));
        const int rest = npad - J - bJ;
        if (rest > 0)
            BQCHK(launch_gemm_rows(c, BQ_K_GEMM, Xin + (long)(J + bJ) * ldx, ldx,
                                   Xout + (long)J * ldx, ldx, L + J + bJ + (long)J * ldl, 1, ldl,
                                   mrows, rest, bJ));
    }
    return BQ_OK;
}

// Xout (zeroed here) <- Xin L^-1 (the L^T sweep of dpotrs in row form); Xin is overwritten
int enqueue_backward_rows(bq_ctx *c, double *Xin, double *Xout, long ldx, int mrows,
                          const double *L, long ldl, int npad, WideInv w)
{
    HIPCHK(c, hipMemsetAsync(Xout, 0, sizeof(double) * (size_t)ldx * npad, c->cur));
    const int last = (npad - 1) / w.B * w.B;
    for (int J = last; J >= 0; J -= w.B) {
        const int bJ = std::min(w.B, npad - J);
        BQCHK(launch_gemm_rows(c, BQ_K_TRSM, Xout + (long)J * ldx, ldx, Xin + (long)J * ldx, ldx,
                               w.nr + (size_t)J * w.B, 1, w.B, mrows, bJ, bJ));
        if (J > 0) // Xin[:, 0:J] -= Xout[:, J:J+bJ] L[J:J+bJ, 0:J]
            BQCHK(launch_gemm_rows(c, BQ_K_GEMM, Xin, ldx, Xout + (long)J * ldx, ldx, L + J, ldl, 1,
                                   mrows, J, bJ));
    }
    return BQ_OK;
}

// (L L^T) X = B for nrhs host columns through the row-form sweeps: the columns go up as they
// are, a device transposition puts them into the rows of the sweep buffers (mpad x npad,
// X_dev[r, j] = B[j, r]) and the solution back
int solve_rows_host(bq_ctx *c, const double *L, long ldl, int n, int npad, WideInv w,
                    const double *B, int64_t nrhs, double *X)
{
    const int mpad = (int)roundup(nrhs, 64);
    DevBuf Bd, Xd, X2;
    HIPCHK(c, Bd.alloc(sizeof(double) * (size_t)n * nrhs));
    HIPCHK(c, Xd.alloc(sizeof(double) * (size_t)mpad * npad));
    HIPCHK(c, X2.alloc(sizeof(double) * (size_t)mpad * npad));
    HIPCHK(c, hipMemcpyAsync(Bd.p, B, Bd.bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(Xd.p, 0, Xd.bytes, c->stream));
    const dim3 g((n + 63) / 64, (unsigned)((nrhs + 63) / 64));
    hipLaunchKernelGGL(transpose_pad_kernel, g, dim3(256), 0, c->stream, Bd.d(), (long)n, n,
                       (int)nrhs, Xd.d(), (long)mpad);
    HIPCHK(c, hipGetLastError());
    BQCHK(enqueue_forward_rows(c, Xd.d(), X2.d(), mpad, mpad, L, ldl, npad, w));
    BQCHK(enqueue_backward_rows(c, X2.d(), Xd.d(), mpad, mpad, L, ldl, npad, w));
    const dim3 gb((unsigned)((nrhs + 63) / 64), (n + 63) / 64);
    hipLaunchKernelGGL(transpose_pad_kernel, gb, dim3(256), 0, c->stream, Xd.d(), (long)mpad,
                       (int)nrhs, n, Bd.d(), (long)n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(X, Bd.p, Bd.bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

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

int check_dims(bq_ctx *c, int64_t d, int64_t n)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (d < 1 || d > BQ_MAXD)
        return fail(c, BQ_ERR_BAD_ARG, "d must be in 1..%d", BQ_MAXD);
    if (n < 1 || n > (1 << 20))
        return fail(c, BQ_ERR_BAD_ARG, "n out of range");
    return BQ_OK;
}

int check_w(bq_ctx *c, int64_t d, double h, const double *w, double s)
{
    if (!w)
        return fail(c, BQ_ERR_BAD_ARG, "w is NULL");
    if (!(std::isfinite(h)) || !(std::isfinite(s)))
        return fail(c, BQ_ERR_BAD_ARG, "h and s must be finite");
    for (int k = 0; k < d; ++k)
        if (!(w[k] > 0.0) || !std::isfinite(w[k]))
            return fail(c, BQ_ERR_BAD_ARG, "w must be positive and finite");
    return BQ_OK;
}

} // namespace

// ===========================================================================
// contexts
// ===========================================================================
extern "C" int bq_device_count(int *count)
{
    if (!count)
        return BQ_ERR_BAD_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        n = 0;
    *count = n;
    return BQ_OK;
}

static int ctx_init(bq_ctx *c, int device)
{
    HIPCHK(c, hipSetDevice(device));
    c->device = device;
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, device));
    c->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(c, hipEventCreate(&c->t0));
    HIPCHK(c, hipEventCreate(&c->t1));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_panel, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_next, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    int lo = 0, hi = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(c, hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, hi));
    // the LDS-staged trailing update uses 72 KiB of dynamic LDS per workgroup
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_lds_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, BQ_LDS_BYTES));
    if (const char *e = std::getenv("BQ_LOOKAHEAD"))
        c->lookahead = std::atoi(e);
    if (const char *e = std::getenv("BQ_SPLIT"))
        c->split_batch = std::atoi(e);
    if (const char *e = std::getenv("BQ_LA_MIN"))
        c->la_min = std::atoi(e);
    if (const char *e = std::getenv("BQ_GEMM_LDS"))
        c->gemm_lds = std::atoi(e);
    if (const char *e = std::getenv("BQ_GRAPH"))
        c->use_graph = std::atoi(e);
    return BQ_OK;
}

extern "C" int bq_ctx_create(int device, bq_ctx **out)
{
    if (!out)
        return BQ_ERR_BAD_ARG;
    *out = nullptr;
    bq_ctx *c = new (std::nothrow) bq_ctx();
    if (!c)
        return BQ_ERR_NOMEM;
    int st = ctx_init(c, device);
    if (st == BQ_OK) {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess)
            st = fail(c, BQ_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        c->own_stream = true;
        c->cur = c->stream;
    }
    if (st != BQ_OK) {
        fprintf(stderr, "bq_ctx_create: %s\n", c->err);
        delete c;
        return st;
    }
    *out = c;
    return BQ_OK;
}

extern "C" int bq_ctx_create_on_stream(int device, void *hip_stream, bq_ctx **out)
{
    if (!out)
        return BQ_ERR_BAD_ARG;
    *out = nullptr;
    bq_ctx *c = new (std::nothrow) bq_ctx();
    if (!c)
        return BQ_ERR_NOMEM;
    int st = ctx_init(c, device);
    if (st != BQ_OK) {
        delete c;
        return st;
    }
    c->stream = static_cast<hipStream_t>(hip_stream);
    c->cur = c->stream;
    c->own_stream = false;
    *out = c;
    return BQ_OK;
}

extern "C" void bq_ctx_destroy(bq_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->plan_cache) {
        bq_plan_destroy(c, c->plan_cache);
        c->plan_cache = nullptr;
    }
    for (auto &e : c->prof_events) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    if (c->t0)
        (void)hipEventDestroy(c->t0);
    if (c->t1)
        (void)hipEventDestroy(c->t1);
    for (hipEvent_t e : {c->ev_panel, c->ev_next, c->ev_fork})
        if (e)
            (void)hipEventDestroy(e);
    if (c->aux) {
        (void)hipStreamSynchronize(c->aux);
        (void)hipStreamDestroy(c->aux);
    }
    if (c->own_stream && c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int bq_ctx_sync(bq_ctx *c)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

extern "C" const char *bq_last_error(const bq_ctx *c) { return c ? c->err : "null context"; }

extern "C" int bq_device_info(bq_ctx *c, char *name, int *cus, size_t *hbm_bytes, int *clock_khz)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
    if (name) {
        std::snprintf(name, 64, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (cus)
        *cus = prop.multiProcessorCount;
    if (hbm_bytes)
        *hbm_bytes = prop.totalGlobalMem;
    if (clock_khz)
        *clock_khz = prop.clockRate;
    return BQ_OK;
}

extern "C" int bq_set_block(bq_ctx *c, int nb)
{
    if (!c || nb < 0 || (nb & 63))
        return c ? fail(c, BQ_ERR_BAD_ARG, "block must be a multiple of 64") : BQ_ERR_BAD_ARG;
    c->nb_override = nb;
    return BQ_OK;
}

extern "C" int bq_set_lookahead(bq_ctx *c, int on)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    c->lookahead = on ? 1 : 0;
    return BQ_OK;
}

extern "C" int bq_set_lookahead_rows(bq_ctx *c, int min_rows)
{
    if (!c || min_rows < 0)
        return c ? fail(c, BQ_ERR_BAD_ARG, "min_rows must be >= 0") : BQ_ERR_BAD_ARG;
    c->la_min = min_rows;
    return BQ_OK;
}

// ===========================================================================
// memory, timers, profiling
// ===========================================================================
extern "C" int bq_dev_alloc(bq_ctx *c, size_t bytes, void **dptr)
{
    if (!c || !dptr)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(dptr, bytes ? bytes : 8));
    return BQ_OK;
}

extern "C" int bq_dev_free(bq_ctx *c, void *dptr)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (dptr) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(dptr));
    }
    return BQ_OK;
}

extern "C" int bq_upload(bq_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || (!dst && bytes) || (!src && bytes))
        return BQ_ERR_BAD_ARG;
    if (bytes == 0)
        return BQ_OK;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

extern "C" int bq_download(bq_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (!c || (!dst && bytes) || (!src && bytes))
        return BQ_ERR_BAD_ARG;
    if (bytes == 0)
        return BQ_OK;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

extern "C" int bq_memset(bq_ctx *c, void *dst, int byte, size_t bytes)
{
    if (!c || !dst)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipMemsetAsync(dst, byte, bytes, c->stream));
    return BQ_OK;
}

extern "C" int bq_timer_start(bq_ctx *c)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipEventRecord(c->t0, c->stream));
    return BQ_OK;
}

extern "C" int bq_timer_stop_ms(bq_ctx *c, float *ms)
{
    if (!c || !ms)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipEventRecord(c->t1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->t1));
    HIPCHK(c, hipEventElapsedTime(ms, c->t0, c->t1));
    return BQ_OK;
}

extern "C" int bq_profile_enable(bq_ctx *c, int on)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    BQCHK(prof_collect(c));
    c->prof = on != 0;
    return BQ_OK;
}

extern "C" int bq_profile_reset(bq_ctx *c)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    BQCHK(prof_collect(c));
    for (int k = 0; k < BQ_K_NCLASS; ++k) {
        c->prof_ms[k] = 0;
        c->prof_n[k] = 0;
        c->prof_work[k] = 0;
    }
    return BQ_OK;
}

extern "C" int bq_profile_read(bq_ctx *c, double *ms, int64_t *launches, double *work)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    BQCHK(prof_collect(c));
    for (int k = 0; k < BQ_K_NCLASS; ++k) {
        if (ms)
            ms[k] = c->prof_ms[k];
        if (launches)
            launches[k] = c->prof_n[k];
        if (work)
            work[k] = c->prof_work[k];
    }
    return BQ_OK;
}

// ===========================================================================
// linalg_c drop-ins (host buffers)
// ===========================================================================
namespace {

// upload an n x n host matrix (ld n) into a padded ntot x ntot device matrix
int upload_padded(bq_ctx *c, const double *H, int n, DevBuf &A, int &ntot, long &lda)
{
    ntot = (int)roundup(n, 64);
    lda = pick_ld(ntot);
    HIPCHK(c, A.alloc(sizeof(double) * (size_t)lda * ntot));
    HIPCHK(c, hipMemcpy2DAsync(A.p, sizeof(double) * lda, H, sizeof(double) * n,
                               sizeof(double) * n, n, hipMemcpyHostToDevice, c->stream));
    if (ntot > n) {
        hipLaunchKernelGGL(pad_identity_kernel, dim3((ntot + 255) / 256, ntot), dim3(256), 0,
                           c->stream, A.d(), lda, n, ntot);
        HIPCHK(c, hipGetLastError());
    }
    return BQ_OK;
}

} // namespace

extern "C" int bq_cho_factor(bq_ctx *c, const double *C, double *L, int64_t n, int64_t *info_out)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (info_out)
        *info_out = 0;
    if (!C || !L || n < 0)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (n == 0)
        return BQ_OK;
    if (n > 65536)
        return fail(c, BQ_ERR_BAD_ARG, "n too large");
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf A, ws;
    int ntot;
    long lda;
    BQCHK(upload_padded(c, C, (int)n, A, ntot, lda));
    HIPCHK(c, ws.alloc(BQ_DINV_STRIDE * sizeof(double) + 64));
    double *dinv = ws.d();
    int *info = reinterpret_cast<int *>(ws.d() + BQ_DINV_STRIDE);
    HIPCHK(c, hipMemsetAsync(info, 0, sizeof(int), c->stream));
    if (c->panel_ws.bytes < sizeof(double) * panel_ws_doubles(ntot, 1))
        HIPCHK(c, c->panel_ws.alloc(sizeof(double) * panel_ws_doubles(ntot, 1)));
    BQCHK(enqueue_potrf_partial(c, A.d(), lda, 0, 1, ntot, ntot, dinv, info, c->panel_ws.d(),
                                c->panel_ws.bytes / sizeof(double)));
    int hinfo = 0;
    HIPCHK(c, hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (hinfo != 0) {
        if (info_out)
            *info_out = hinfo;
        return fail(c, BQ_ERR_NOT_PD, "matrix is not positive definite");
    }
    // copy back only the lower triangle; the strict upper part of L keeps what
    // the caller had there (C's values after the reference's C -> L copy)
    if (C != L) {
        // straight into L, then the caller's strict upper triangle over what the device left there
        HIPCHK(c, hipMemcpy2D(L, sizeof(double) * n, A.p, sizeof(double) * lda, sizeof(double) * n,
                              n, hipMemcpyDeviceToHost));
        for (int64_t j = 1; j < n; ++j)
            std::memcpy(L + j * n, C + j * n, sizeof(double) * (size_t)j);
        return BQ_OK;
    }
    std::vector<double> tmp((size_t)n * n);
    HIPCHK(c, hipMemcpy2D(tmp.data(), sizeof(double) * n, A.p, sizeof(double) * lda,
                          sizeof(double) * n, n, hipMemcpyDeviceToHost));
    for (int64_t j = 0; j < n; ++j)
        std::memcpy(L + j + j * n, tmp.data() + j + j * n, sizeof(double) * (size_t)(n - j));
    return BQ_OK;
}

extern "C" int bq_cho_solve(bq_ctx *c, const double *L, const double *B, double *X, int64_t n,
                            int64_t nrhs)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!L || !B || !X || n < 0 || nrhs < 0)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (n == 0 || nrhs == 0)
        return BQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf A, ws, Xd;
    int npad;
    long ldl;
    // the strict upper triangle of L is never read by the sweeps
    BQCHK(upload_padded(c, L, (int)n, A, npad, ldl));
    // the block inverses of the factor's diagonal: 16 x 16 (panel solve), then B wide
    DevBuf wide, X2;
    HIPCHK(c, ws.alloc(sizeof(double) * BQ_DINV_HALF * (size_t)(npad / 64)));
    HIPCHK(c, wide.alloc(sizeof(double) * wide_alloc_doubles(npad)));
    hipLaunchKernelGGL(diag_winv_kernel, dim3(npad / 64), dim3(256), 0, c->stream, A.d(), ldl,
                       ws.d());
    HIPCHK(c, hipGetLastError());
    BQCHK(compute_wide_inverses(c, A.d(), ldl, npad, ws.d(), wide.d()));
    const WideInv w = wide_views(wide.d(), npad);
    if (nrhs == 1) {
        // one right-hand side: the GEMV sweeps (trsv.h)
        HIPCHK(c, Xd.alloc(sizeof(double) * 2 * (size_t)npad));
        HIPCHK(c, hipMemsetAsync(Xd.p, 0, Xd.bytes, c->stream));
        HIPCHK(c, hipMemcpyAsync(Xd.p, B, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
        double *x = Xd.d(), *y = Xd.d() + npad;
        BQCHK(enqueue_forward_vec(c, x, y, A.d(), ldl, npad, w));
        BQCHK(enqueue_backward_vec(c, y, x, A.d(), ldl, npad, w));
        HIPCHK(c, hipMemcpyAsync(X, x, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return BQ_OK;
    }
    return solve_rows_host(c, A.d(), ldl, (int)n, npad, w, B, nrhs, X);
}

extern "C" int bq_logdet(bq_ctx *c, const double *L, int64_t n, double *out)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!L || !out || n < 0)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    HIPCHK(c, hipSetDevice(c->device));
    if (n == 0) {
        *out = 0.0;
        return BQ_OK;
    }
    // only the diagonal travels
    std::vector<double> diag((size_t)n);
    for (int64_t i = 0; i < n; ++i)
        diag[(size_t)i] = L[i + i * n];
    DevBuf dv;
    HIPCHK(c, dv.alloc(sizeof(double) * (n + 1)));
    HIPCHK(c, hipMemcpyAsync(dv.p, diag.data(), sizeof(double) * n, hipMemcpyHostToDevice,
                             c->stream));
    hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(256), 0, c->stream, dv.d(), 0L, (int)n,
                       dv.d() + n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, dv.d() + n, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

// ===========================================================================
// Gram
// ===========================================================================
extern "C" int bq_gram_gauss_dev(bq_ctx *c, const double *x_dev, int64_t d, int64_t n, double h,
                                 const double *w, double s, double *K_dev, int64_t ldk)
{
    BQCHK(check_dims(c, d, n));
    BQCHK(check_w(c, d, h, w, s));
    if (!x_dev || !K_dev || ldk < n)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    // parameters ride in a tiny device buffer so the same kernel serves the
    // batched callers; uploaded only when they change
    GaussParams g = make_params((int)d, h, w, s);
    if (!c->gbuf.p) {
        HIPCHK(c, c->gbuf.alloc(sizeof(GaussParams)));
        c->gbuf_valid = false;
    }
    if (!c->gbuf_valid || std::memcmp(&c->gbuf_host, &g, sizeof g) != 0) {
        // synchronous on purpose: the staging copy of `g` must not outlive this frame
        HIPCHK(c, hipMemcpyAsync(c->gbuf.p, &g, sizeof g, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->gbuf_host = g;
        c->gbuf_valid = true;
    }
    return launch_gram_sym(c, (int)d, x_dev, 0, static_cast<GaussParams *>(c->gbuf.p), 0, K_dev,
                           ldk, 0, (int)n, 1);
}

extern "C" int bq_gram_gauss(bq_ctx *c, const double *x, int64_t d, int64_t n, double h,
                             const double *w, double s, double *K_out)
{
    BQCHK(check_dims(c, d, n));
    BQCHK(check_w(c, d, h, w, s));
    if (!x || !K_out)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf xd, Kd, gd;
    HIPCHK(c, xd.alloc(sizeof(double) * d * n));
    HIPCHK(c, Kd.alloc(sizeof(double) * (size_t)n * n));
    HIPCHK(c, gd.alloc(sizeof(GaussParams)));
    GaussParams g = make_params((int)d, h, w, s);
    HIPCHK(c, hipMemcpyAsync(xd.p, x, sizeof(double) * d * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(gd.p, &g, sizeof g, hipMemcpyHostToDevice, c->stream));
    BQCHK(launch_gram_sym(c, (int)d, xd.d(), 0, static_cast<GaussParams *>(gd.p), 0, Kd.d(), n, 0,
                          (int)n, 1));
    HIPCHK(c, hipMemcpyAsync(K_out, Kd.p, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

extern "C" int bq_gram_gauss_cross(bq_ctx *c, const double *x1, int64_t n1, const double *x2,
                                   int64_t n2, int64_t d, double h, const double *w, double *K_out)
{
    BQCHK(check_dims(c, d, n1 > 0 ? n1 : 1));
    BQCHK(check_w(c, d, h, w, 0.0));
    if (n1 < 0 || n2 < 0 || (n1 && !x1) || (n2 && !x2) || (!K_out && n1 * n2))
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (n1 == 0 || n2 == 0)
        return BQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf a, b, Kd;
    HIPCHK(c, a.alloc(sizeof(double) * d * n1));
    HIPCHK(c, b.alloc(sizeof(double) * d * n2));
    HIPCHK(c, Kd.alloc(sizeof(double) * (size_t)n1 * n2));
    HIPCHK(c, hipMemcpyAsync(a.p, x1, sizeof(double) * d * n1, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(b.p, x2, sizeof(double) * d * n2, hipMemcpyHostToDevice, c->stream));
    GaussParams g = make_params((int)d, h, w, 0.0);
    BQCHK(launch_gram_cross(c, (int)d, a.d(), (int)n1, b.d(), (int)n2, g, Kd.d(), n1));
    HIPCHK(c, hipMemcpyAsync(K_out, Kd.p, sizeof(double) * (size_t)n1 * n2, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

// ===========================================================================
// device-resident Cholesky
// ===========================================================================
extern "C" int bq_potrf_dev(bq_ctx *c, double *A_dev, int64_t n, int64_t lda, int32_t *info_dev)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!A_dev || !info_dev || n <= 0 || (n & 63) || lda < n || (lda & 1))
        return fail(c, BQ_ERR_BAD_ARG, "potrf_dev: n must be a positive multiple of 64, lda even");
    if (!c->dinv64.p)
        HIPCHK(c, c->dinv64.alloc(BQ_DINV_STRIDE * sizeof(double)));
    HIPCHK(c, hipMemsetAsync(info_dev, 0, sizeof(int32_t), c->stream));
    if (c->panel_ws.bytes < sizeof(double) * panel_ws_doubles((int)n, 1))
        HIPCHK(c, c->panel_ws.alloc(sizeof(double) * panel_ws_doubles((int)n, 1)));
    return enqueue_potrf_partial(c, A_dev, lda, 0, 1, (int)n, (int)n, c->dinv64.d(), info_dev,
                                 c->panel_ws.d(), c->panel_ws.bytes / sizeof(double));
}

// ===========================================================================
// plans: resident batched "fit + posterior + log-ML"
// ===========================================================================
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
    int graph_nb = 0, graph_la = 0, graph_pw = 0;
};

namespace {

Layout make_layout(int n, int M, bool has_y)
{
    Layout L;
    L.n = n;
    L.npad = (int)roundup(n, 64);
    L.M = M;
    L.yrow = has_y ? L.npad + M : -1;
    L.ntot = (int)roundup(L.npad + M + (has_y ? 1 : 0), 64);
    return L;
}

} // namespace

extern "C" int bq_plan_create(bq_ctx *c, int64_t nprob, int64_t d, int64_t n, int64_t M,
                              bq_plan **out)
{
    if (!out)
        return BQ_ERR_BAD_ARG;
    *out = nullptr;
    BQCHK(check_dims(c, d, n));
    if (nprob < 1 || M < 0 || M > (1 << 20) || nprob > 65535)
        return fail(c, BQ_ERR_BAD_ARG, "illegal batch / M");
    HIPCHK(c, hipSetDevice(c->device));
    bq_plan *p = new (std::nothrow) bq_plan();
    if (!p)
        return fail(c, BQ_ERR_NOMEM, "out of host memory");
    p->nprob = (int)nprob;
    p->d = (int)d;
    p->n = (int)n;
    p->M = (int)M;
    p->L = make_layout((int)n, (int)M, true);
    p->lda = pick_ld(p->L.ntot);
    p->astride = p->lda * (long)p->L.ntot;
    hipError_t e = hipSuccess;
    auto A = [&](DevBuf &b, size_t bytes) {
        if (e == hipSuccess)
            e = b.alloc(bytes);
    };
    A(p->A, sizeof(double) * (size_t)p->astride * nprob);
    A(p->pts, sizeof(double) * (size_t)d * p->L.ntot * nprob);
    A(p->y, sizeof(double) * (size_t)p->L.npad * nprob);
    A(p->gp, sizeof(GaussParams) * (size_t)nprob);
    A(p->dinv, sizeof(double) * BQ_DINV_STRIDE * (size_t)nprob);
    A(p->panel, panel_ws_useful(c, p->L.ntot, (int)nprob)
                    ? sizeof(double) * panel_ws_doubles(p->L.ntot, (int)nprob)
                    : 0);
    A(p->info, sizeof(int) * (size_t)nprob);
    A(p->scal, sizeof(double) * 4 * (size_t)nprob);
    A(p->mean, sizeof(double) * (size_t)std::max<int64_t>(M, 1) * nprob);
    A(p->var, sizeof(double) * (size_t)std::max<int64_t>(M, 1) * nprob);
    if (e != hipSuccess) {
        delete p;
        return fail(c, e == hipErrorOutOfMemory ? BQ_ERR_NOMEM : BQ_ERR_HIP,
                    "plan allocation failed: %s", hipGetErrorString(e));
    }
    HIPCHK(c, hipMemsetAsync(p->pts.p, 0, p->pts.bytes, c->stream));
    HIPCHK(c, hipMemsetAsync(p->y.p, 0, p->y.bytes, c->stream));
    *out = p;
    return BQ_OK;
}

namespace {
void plan_drop_graph(bq_plan *p);
}

extern "C" void bq_plan_destroy(bq_ctx *c, bq_plan *p)
{
    if (!p)
        return;
    if (c) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
    }
    plan_drop_graph(p);
    delete p;
}

namespace {

// The batched host entry points (bq_batch_fit_predict, bq_gp_logml_grid) keep their plan --
// up to half of the free HBM -- in the context between calls: a hyper-parameter loop calls
// them again and again with the same shapes, and allocating and releasing tens of GB per
// call costs milliseconds every time and, now and then, hundreds (observed on the C3 grid:
// 230 ms typical, 0.5 - 1.6 s spikes).  bq_ctx_trim() releases it.
int plan_acquire(bq_ctx *c, int64_t nprob, int64_t d, int64_t n, int64_t M, bq_plan **out)
{
    bq_plan *p = c->plan_cache;
    c->plan_cache = nullptr;
    if (p && p->nprob == nprob && p->d == d && p->n == n && p->M == M) {
        *out = p;
        return BQ_OK;
    }
    if (p)
        bq_plan_destroy(c, p);
    return bq_plan_create(c, nprob, d, n, M, out);
}

void plan_release(bq_ctx *c, bq_plan *p)
{
    if (!p)
        return;
    if (c->plan_cache)
        bq_plan_destroy(c, c->plan_cache);
    c->plan_cache = p;
}

} // namespace

extern "C" int bq_ctx_trim(bq_ctx *c)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (c->plan_cache) {
        bq_plan_destroy(c, c->plan_cache);
        c->plan_cache = nullptr;
    }
    (void)hipStreamSynchronize(c->stream);
    c->scratch.release();
    return BQ_OK;
}

extern "C" int bq_plan_bytes(bq_plan *p, size_t *bytes)
{
    if (!p || !bytes)
        return BQ_ERR_BAD_ARG;
    *bytes = p->A.bytes + p->pts.bytes + p->y.bytes + p->gp.bytes + p->dinv.bytes +
             p->info.bytes + p->scal.bytes + p->mean.bytes + p->var.bytes + p->panel.bytes;
    return BQ_OK;
}

extern "C" int bq_plan_set_inputs(bq_ctx *c, bq_plan *p, const double *x, const double *y,
                                  const double *xo, const double *h, const double *w,
                                  const double *s)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!p)
        return fail(c, BQ_ERR_BAD_ARG, "null plan handle");
    if (!x || !y || (!xo && p->M) || !h || !w || !s)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    HIPCHK(c, hipSetDevice(c->device));
    const int d = p->d, n = p->n, M = p->M, ntot = p->L.ntot, npad = p->L.npad;
    p->hgp.resize(p->nprob);
    for (int b = 0; b < p->nprob; ++b) {
        BQCHK(check_w(c, d, h[b], w + (size_t)b * d, s[b]));
        p->hgp[b] = make_params(d, h[b], w + (size_t)b * d, s[b]);
    }
    HIPCHK(c, hipMemcpyAsync(p->gp.p, p->hgp.data(), sizeof(GaussParams) * p->nprob,
                             hipMemcpyHostToDevice, c->stream));
    // points: x at columns [0,n), xo at [npad, npad+M) of each problem's d x ntot block
    HIPCHK(c, hipMemcpy2DAsync(p->pts.p, sizeof(double) * d * ntot, x, sizeof(double) * d * n,
                               sizeof(double) * d * n, p->nprob, hipMemcpyHostToDevice,
                               c->stream));
    if (M > 0)
        HIPCHK(c, hipMemcpy2DAsync(p->pts.d() + (size_t)d * npad, sizeof(double) * d * ntot, xo,
                                   sizeof(double) * d * M, sizeof(double) * d * M, p->nprob,
                                   hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpy2DAsync(p->y.p, sizeof(double) * npad, y, sizeof(double) * n,
                               sizeof(double) * n, p->nprob, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    p->has_inputs = true;
    return BQ_OK;
}

namespace {

int plan_enqueue(bq_ctx *c, bq_plan *p)
{
    // a small system's first sweep launch (and the clearing of the failure flags) rides in
    // the assembly
    FirstStep fs;
    const bool fuse = sweep_is_slab(c, p->L.ntot, p->L.npad, p->nprob,
                                    p->panel.bytes / sizeof(double));
    if (fuse) {
        fs.S0 = p->panel.d();
        fs.lds = p->L.ntot;
        fs.sstride = 64L * p->L.ntot;
        fs.dinv = p->dinv.d();
        fs.info = p->info.i();
    } else {
        HIPCHK(c, hipMemsetAsync(p->info.p, 0, sizeof(int) * p->nprob, c->stream));
    }
    BQCHK(launch_assemble(c, p->d, p->pts.d(), (long)p->d * p->L.ntot, p->y.d(), p->L.npad,
                          static_cast<GaussParams *>(p->gp.p), 1, p->A.d(), p->lda, p->astride,
                          p->L, p->nprob, fs));
    // A blocked sweep (outer block >= 128) reads its results off the border rows and skips the
    // border x border block in its trailing updates; the one-launch steps of small systems
    // update everything and read the Schur complement.
    const bool by_rows = !fuse && p->L.yrow >= 0 && auto_nb(c, p->L.ntot, p->nprob) >= 128;
    BQCHK(enqueue_potrf_partial(c, p->A.d(), p->lda, p->astride, p->nprob, p->L.ntot, p->L.npad,
                                p->dinv.d(), p->info.i(), p->panel.d(),
                                p->panel.bytes / sizeof(double), fuse, by_rows));
    if (by_rows) {
        Bracket br(c, BQ_K_REDUCE, 8.0 * (p->M + 1.0) * p->L.npad * p->nprob);
        hipLaunchKernelGGL(plan_readout_kernel, dim3((p->M + 16) / 16, 1, p->nprob), dim3(1024), 0,
                           c->stream, p->A.d(), p->lda, p->astride, p->L,
                           static_cast<const GaussParams *>(p->gp.p), p->scal.d(), p->mean.d(),
                           p->var.d(), (long)std::max(p->M, 1));
        HIPCHK(c, hipGetLastError());
    } else {
        Bracket br(c, BQ_K_REDUCE, 8.0 * (p->n + 2.0 * p->M) * p->nprob);
        hipLaunchKernelGGL(finalize_kernel, dim3(1, 1, p->nprob), dim3(256), 0, c->stream,
                           p->A.d(), p->lda, p->astride, p->L, p->scal.d(), p->mean.d(),
                           p->var.d(), (long)std::max(p->M, 1));
        HIPCHK(c, hipGetLastError());
    }
    return BQ_OK;
}

void plan_drop_graph(bq_plan *p)
{
    if (p->gexec)
        (void)hipGraphExecDestroy(p->gexec);
    if (p->graph)
        (void)hipGraphDestroy(p->graph);
    p->gexec = nullptr;
    p->graph = nullptr;
}

} // namespace

extern "C" int bq_plan_run(bq_ctx *c, bq_plan *p)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!p)
        return fail(c, BQ_ERR_BAD_ARG, "null plan handle");
    if (!p->has_inputs)
        return fail(c, BQ_ERR_BAD_ARG, "plan has no inputs");
    if (c->prof || !c->use_graph || !c->own_stream)
        return plan_enqueue(c, p);
    // settings that change the launch sequence invalidate the captured graph
    if (p->graph_state == 1 && (p->graph_nb != c->nb_override || p->graph_la != c->lookahead ||
                                p->graph_pw != c->la_min * 4 + c->split_batch * 2 + c->gemm_lds)) {
        plan_drop_graph(p);
        p->graph_state = 0;
    }
    if (p->graph_state == 0) {
        p->graph_nb = c->nb_override;
        p->graph_la = c->lookahead;
        p->graph_pw = c->la_min * 4 + c->split_batch * 2 + c->gemm_lds;
        p->graph_state = -1;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
            const int st = plan_enqueue(c, p);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(c->stream, &g);
            if (st == BQ_OK && e == hipSuccess && g &&
                hipGraphInstantiate(&p->gexec, g, nullptr, nullptr, 0) == hipSuccess) {
                p->graph = g;
                p->graph_state = 1;
            } else {
                if (g)
                    (void)hipGraphDestroy(g);
                (void)hipGetLastError(); // clear; fall back to eager launches
            }
        } else {
            (void)hipGetLastError();
        }
    }
    if (p->graph_state == 1) {
        HIPCHK(c, hipGraphLaunch(p->gexec, c->stream));
        return BQ_OK;
    }
    return plan_enqueue(c, p);
}

extern "C" int bq_plan_results(bq_ctx *c, bq_plan *p, double *mean, double *var, double *logml,
                               int32_t *status)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!p)
        return fail(c, BQ_ERR_BAD_ARG, "null plan handle");
    const int nb = p->nprob, M = p->M;
    std::vector<double> scal((size_t)nb * 4);
    std::vector<int> info((size_t)nb);
    HIPCHK(c, hipMemcpyAsync(scal.data(), p->scal.p, sizeof(double) * 4 * nb,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(info.data(), p->info.p, sizeof(int) * nb, hipMemcpyDeviceToHost,
                             c->stream));
    if (mean && M)
        HIPCHK(c, hipMemcpyAsync(mean, p->mean.p, sizeof(double) * (size_t)M * nb,
                                 hipMemcpyDeviceToHost, c->stream));
    if (var && M)
        HIPCHK(c, hipMemcpyAsync(var, p->var.p, sizeof(double) * (size_t)M * nb,
                                 hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int b = 0; b < nb; ++b) {
        if (status)
            status[b] = info[b];
        if (logml)
            logml[b] = info[b] ? -std::numeric_limits<double>::infinity() : scal[(size_t)b * 4];
    }
    return BQ_OK;
}

// ===========================================================================
// one-shot and batched host entry points built on plans
// ===========================================================================
extern "C" int bq_batch_fit_predict(bq_ctx *c, int64_t nprob, const double *x, const double *y,
                                    int64_t d, int64_t n, double h, const double *w, double s,
                                    const double *xo, int64_t M, double *mean, double *var,
                                    double *logml, int32_t *status)
{
    BQCHK(check_dims(c, d, n));
    BQCHK(check_w(c, d, h, w, s));
    if (nprob < 1)
        return fail(c, BQ_ERR_BAD_ARG, "nprob < 1");
    // bound the resident working set: chunks of problems
    const Layout L = make_layout((int)n, (int)M, true);
    const size_t per = sizeof(double) * (size_t)pick_ld(L.ntot) * L.ntot;
    size_t freeb = 0, totalb = 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemGetInfo(&freeb, &totalb));
    int64_t chunk = std::max<int64_t>(1, (int64_t)((freeb / 2) / per));
    chunk = std::min<int64_t>(chunk, nprob);
    // free memory was read with the cached workspace still allocated: keep its size
    if (const bq_plan *q = c->plan_cache)
        if (q->d == d && q->n == n && q->M == M && q->nprob <= nprob && q->nprob >= chunk)
            chunk = q->nprob;
    bq_plan *p = nullptr, *big = nullptr;
    BQCHK(plan_acquire(c, chunk, d, n, M, &p));
    big = p;
    std::vector<double> hh((size_t)chunk, h), ss((size_t)chunk, s), ww((size_t)chunk * d);
    for (int64_t b = 0; b < chunk; ++b)
        for (int64_t k = 0; k < d; ++k)
            ww[(size_t)(b * d + k)] = w[k];
    int st = BQ_OK;
    for (int64_t p0 = 0; p0 < nprob && st == BQ_OK; p0 += chunk) {
        const int64_t nb = std::min(chunk, nprob - p0);
        if (nb != chunk) { // last, smaller chunk: a temporary plan of the right size
            p = nullptr;
            st = bq_plan_create(c, nb, d, n, M, &p);
            if (st != BQ_OK)
                break;
        }
        st = bq_plan_set_inputs(c, p, x + (size_t)p0 * d * n, y + (size_t)p0 * n,
                                xo ? xo + (size_t)p0 * d * M : nullptr, hh.data(), ww.data(),
                                ss.data());
        if (st == BQ_OK)
            st = bq_plan_run(c, p);
        if (st == BQ_OK)
            st = bq_plan_results(c, p, mean ? mean + (size_t)p0 * M : nullptr,
                                 var ? var + (size_t)p0 * M : nullptr,
                                 logml ? logml + p0 : nullptr, status ? status + p0 : nullptr);
    }
    if (p != big)
        bq_plan_destroy(c, p);
    plan_release(c, big);
    return st;
}

extern "C" int bq_fit_predict(bq_ctx *c, const double *x, const double *y, int64_t d, int64_t n,
                              double h, const double *w, double s, const double *xo, int64_t M,
                              double *mean, double *var, double *logml)
{
    int32_t status = 0;
    BQCHK(bq_batch_fit_predict(c, 1, x, y, d, n, h, w, s, xo, M, mean, var, logml, &status));
    if (status != 0)
        return fail(c, BQ_ERR_NOT_PD, "matrix is not positive definite");
    return BQ_OK;
}

namespace {

// G hyper-parameter points in chunks of batched plans: log-ML per point (-inf where the
// factorisation fails) and, on request, its two ingredients log|K| and y^T K^-1 y
int logml_grid_core(bq_ctx *c, const double *x, const double *y, int64_t d, int64_t n,
                    const double *h, const double *w, double s, int64_t G, int64_t chunk,
                    double *lm, double *logdet, double *qf)
{
    const Layout L = make_layout((int)n, 0, true);
    const size_t per = sizeof(double) * (size_t)pick_ld(L.ntot) * L.ntot;
    if (chunk <= 0) {
        size_t freeb = 0, totalb = 0;
        HIPCHK(c, hipMemGetInfo(&freeb, &totalb));
        chunk = std::max<int64_t>(1, (int64_t)((freeb / 2) / per));
    }
    chunk = std::min<int64_t>(chunk, G);
    if (const bq_plan *q = c->plan_cache)
        if (q->d == d && q->n == n && q->M == 0 && q->nprob <= G && q->nprob >= chunk)
            chunk = q->nprob;
    bq_plan *p = nullptr, *big = nullptr;
    BQCHK(plan_acquire(c, chunk, d, n, 0, &p));
    big = p;
    // the data are shared: replicate x, y once for the chunk
    std::vector<double> xr((size_t)chunk * d * n), yr((size_t)chunk * n), ss((size_t)chunk, s);
    for (int64_t b = 0; b < chunk; ++b) {
        std::memcpy(&xr[(size_t)b * d * n], x, sizeof(double) * d * n);
        std::memcpy(&yr[(size_t)b * n], y, sizeof(double) * n);
    }
    std::vector<double> scal((size_t)chunk * 4);
    std::vector<int> info((size_t)chunk);
    int st = BQ_OK;
    for (int64_t g0 = 0; g0 < G && st == BQ_OK; g0 += chunk) {
        const int64_t nb = std::min(chunk, G - g0);
        if (nb != chunk) { // last, smaller chunk: a temporary plan of the right size
            p = nullptr;
            st = bq_plan_create(c, nb, d, n, 0, &p);
            if (st != BQ_OK)
                break;
        }
        st = bq_plan_set_inputs(c, p, xr.data(), yr.data(), nullptr, h + g0, w + (size_t)g0 * d,
                                ss.data());
        if (st == BQ_OK)
            st = bq_plan_run(c, p);
        if (st != BQ_OK)
            break;
        hipError_t e = hipMemcpyAsync(scal.data(), p->scal.p, sizeof(double) * 4 * nb,
                                      hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(info.data(), p->info.p, sizeof(int) * nb, hipMemcpyDeviceToHost,
                               c->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            st = fail(c, BQ_ERR_HIP, "%s", hipGetErrorString(e));
            break;
        }
        for (int64_t b = 0; b < nb; ++b) {
            const bool bad = info[(size_t)b] != 0;
            lm[g0 + b] = bad ? -std::numeric_limits<double>::infinity() : scal[(size_t)b * 4];
            if (logdet)
                logdet[g0 + b] = scal[(size_t)b * 4 + 1];
            if (qf)
                qf[g0 + b] = scal[(size_t)b * 4 + 2];
        }
    }
    if (p != big)
        bq_plan_destroy(c, p);
    plan_release(c, big);
    return st;
}

} // namespace

extern "C" int bq_gp_logml_grid(bq_ctx *c, const double *x, const double *y, int64_t d, int64_t n,
                                const double *h, const double *w, double s, int64_t G, double *out,
                                int64_t chunk)
{
    BQCHK(check_dims(c, d, n));
    if (!x || !y || !h || !w || !out || G < 1)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    HIPCHK(c, hipSetDevice(c->device));
    // Without a noise term K = h^2 G(w): chol(K) = h chol(G), so every output scale h of one
    // length scale w shares ONE factorisation:
    //   log|K| = log|G| + 2 n log h,   y^T K^-1 y = y^T G^-1 y / h^2.
    // Only the distinct w are factored, at h = 1.  (A matrix on the edge of numerical
    // definiteness could pass at one h and fail at another when factored separately; here
    // all h of one w succeed or fail together.)
    if (s == 0.0 && G > 1) {
        std::vector<int64_t> rep((size_t)G), uniq;
        for (int64_t g = 0; g < G; ++g) {
            int64_t r = -1;
            for (size_t u = 0; u < uniq.size() && r < 0; ++u)
                if (std::memcmp(w + (size_t)uniq[u] * d, w + (size_t)g * d, sizeof(double) * d) == 0)
                    r = (int64_t)u;
            if (r < 0) {
                r = (int64_t)uniq.size();
                uniq.push_back(g);
            }
            rep[(size_t)g] = r;
        }
        bool hpos = true;
        for (int64_t g = 0; g < G; ++g)
            hpos = hpos && h[g] > 0.0;
        if ((int64_t)uniq.size() < G && hpos) {
            const int64_t U = (int64_t)uniq.size();
            std::vector<double> hu((size_t)U, 1.0), wu((size_t)U * d), lm((size_t)U),
                ld((size_t)U), qf((size_t)U);
            for (int64_t u = 0; u < U; ++u)
                std::memcpy(&wu[(size_t)u * d], w + (size_t)uniq[(size_t)u] * d, sizeof(double) * d);
            BQCHK(logml_grid_core(c, x, y, d, n, hu.data(), wu.data(), 0.0, U, chunk, lm.data(),
                                  ld.data(), qf.data()));
            for (int64_t g = 0; g < G; ++g) {
                const size_t u = (size_t)rep[(size_t)g];
                const double hh = h[g];
                out[g] = std::isinf(lm[u])
                             ? lm[u]
                             : -0.5 * qf[u] / (hh * hh) -
                                   0.5 * (ld[u] + 2.0 * (double)n * std::log(hh)) -
                                   0.5 * (double)n * 1.8378770664093453;
            }
            return BQ_OK;
        }
    }
    return logml_grid_core(c, x, y, d, n, h, w, s, G, chunk, out, nullptr, nullptr);
}

// ===========================================================================
// GP fit objects
// ===========================================================================
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
    // the single-vector sweeps (trsv.h): x | y, 2 npad doubles, and their captured launch
    // chains -- [0] solve (forward + backward), [1] backward into alpha, [2] forward; the
    // pointers survive a refit, so the graphs do too
    DevBuf vec;
    hipGraph_t vgraph[3] = {nullptr, nullptr, nullptr};
    hipGraphExec_t vgexec[3] = {nullptr, nullptr, nullptr};
    bool vg_failed[3] = {false, false, false};
    ~bq_fit()
    {
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
    double logml = 0, logdet = 0, qf = 0;
};

namespace {

// pm / pv: device buffers for the posterior mean / variance of the layout's M border points
// (bq_gp_refit_predict), or null
int fit_factor(bq_ctx *c, bq_fit *f, double *pm = nullptr, double *pv = nullptr,
               double *hpost = nullptr) // hpost: host copy of misc[8 .. 8 + 128) on return
{
    const int ntot = f->L.ntot;
    int *info = f->misc.i();
    double *scal = f->misc.d() + 2;
    f->valid = false;
    f->have_alpha = false;
    f->have_wide = false;
    f->have_dw = false;
    HIPCHK(c, hipMemcpyAsync(f->gp.p, &f->g, sizeof f->g, hipMemcpyHostToDevice, c->stream));
    double *scratch = f->dinv.d() + f->npad;
    FirstStep fs;
    const bool fuse = sweep_is_slab(c, ntot, f->npad, 1, f->panel.bytes / sizeof(double));
    if (fuse) {
        fs.S0 = f->panel.d();
        fs.lds = ntot;
        fs.sstride = 64L * ntot;
        fs.dinv = scratch;
        fs.info = info;
    } else {
        HIPCHK(c, hipMemsetAsync(info, 0, sizeof(int), c->stream));
    }
    BQCHK(launch_assemble(c, f->d, f->pts.d(), 0, f->y.d(), 0, static_cast<GaussParams *>(f->gp.p),
                          0, f->A.d(), f->ldl, 0, f->L, 1, fs));
    BQCHK(enqueue_potrf_partial(c, f->A.d(), f->ldl, 0, 1, ntot, f->npad, scratch, info,
                                f->panel.d(), f->panel.bytes / sizeof(double), fuse));
    {
        Bracket br(c, BQ_K_REDUCE);
        hipLaunchKernelGGL(finalize_kernel, dim3(1, 1, 1), dim3(256), 0, c->stream, f->A.d(),
                           f->ldl, 0L, f->L, scal, pm, pv, 64L);
        HIPCHK(c, hipGetLastError());
    }
    // one read-back: misc = [info (int, 8 bytes) | pad | scal[4] | pad | mean[64] | var[64]]
    double hm[8 + 128];
    HIPCHK(c, hipMemcpyAsync(hm, f->misc.p, sizeof(double) * (hpost ? 8 + 128 : 6),
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (hpost)
        std::memcpy(hpost, hm + 8, sizeof(double) * 128);
    int hinfo = 0;
    std::memcpy(&hinfo, hm, sizeof hinfo);
    const double *hs = hm + 2;
    f->have_alpha = false;
    if (hinfo != 0)
        return fail(c, BQ_ERR_NOT_PD, "matrix is not positive definite");
    f->logml = hs[0];
    f->logdet = hs[1];
    f->qf = hs[2];
    f->valid = true;
    return BQ_OK;
}

// every consumer of a fit: the handle exists and its last factorisation succeeded
int check_fit(bq_ctx *c, const bq_fit *f)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!f)
        return fail(c, BQ_ERR_BAD_ARG, "null fit handle");
    if (!f->valid)
        return fail(c, BQ_ERR_NOT_PD,
                    "fit holds no valid factor: its last (re)fit was not positive definite");
    return BQ_OK;
}

// the 16 x 16 block inverses of the resident factor's diagonal (the record trsm_blk_kernel
// wants), built on their first use after a (re)fit
int fit_dw(bq_ctx *c, bq_fit *f)
{
    if (!f->have_dw) {
        hipLaunchKernelGGL(diag_winv_kernel, dim3(f->npad / 64), dim3(256), 0, c->stream, f->A.d(),
                           f->ldl, f->dw.d());
        HIPCHK(c, hipGetLastError());
        f->have_dw = true;
    }
    return BQ_OK;
}

// the wide block inverses of the resident factor, built on the first sweep after a (re)fit: a
// hyper-parameter loop that only reads log-ML never pays for them
int fit_wide(bq_ctx *c, bq_fit *f, WideInv &w)
{
    if (!f->have_wide) {
        BQCHK(fit_dw(c, f));
        if (f->wide.bytes < sizeof(double) * wide_alloc_doubles(f->npad))
            HIPCHK(c, f->wide.alloc(sizeof(double) * wide_alloc_doubles(f->npad)));
        BQCHK(compute_wide_inverses(c, f->A.d(), f->ldl, f->npad, f->dw.d(), f->wide.d()));
        f->have_wide = true;
    }
    w = wide_views(f->wide.d(), f->npad);
    return BQ_OK;
}

// the single-vector workspace of a fit (x at vec, y at vec + npad)
int fit_vec(bq_ctx *c, bq_fit *f)
{
    if (f->vec.bytes < sizeof(double) * 2 * (size_t)f->npad)
        HIPCHK(c, f->vec.alloc(sizeof(double) * 2 * (size_t)f->npad));
    return BQ_OK;
}

// Replays a chain of sweep launches over a fit's own buffers from a captured hipGraph (a
// sweep is 2 npad / B launches of 2-8 us each: enqueued one by one the host is the
// bottleneck); eager when graphs are off, under the launch profiler, or if capture fails.
template <class F>
int fit_replay(bq_ctx *c, bq_fit *f, int slot, F &&enqueue)
{
    if (!c->use_graph || c->prof || !c->own_stream || c->cur != c->stream)
        return enqueue();
    if (!f->vgexec[slot] && !f->vg_failed[slot]) {
        f->vg_failed[slot] = true;
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

int fit_alpha(bq_ctx *c, bq_fit *f)
{
    if (f->have_alpha)
        return BQ_OK;
    // alpha = L^-T z, z = A[yrow, 0:npad] (the forward-solved y of the bordered system)
    WideInv w;
    BQCHK(fit_wide(c, f, w));
    BQCHK(fit_vec(c, f));
    // (the gather stays outside the captured chain: the y row moves when the fit carries
    // border points, bq_gp_refit_predict)
    HIPCHK(c, hipMemcpy2DAsync(f->vec.p, sizeof(double), f->A.d() + f->L.yrow,
                               sizeof(double) * f->ldl, sizeof(double), f->npad,
                               hipMemcpyDeviceToDevice, c->stream));
    BQCHK(fit_replay(c, f, 1, [&]() -> int {
        return enqueue_backward_vec(c, f->vec.d(), f->alpha.d(), f->A.d(), f->ldl, f->npad, w);
    }));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    f->have_alpha = true;
    return BQ_OK;
}

} // namespace

extern "C" int bq_gp_fit(bq_ctx *c, const double *x, const double *y, int64_t d, int64_t n,
                         double h, const double *w, double s, bq_fit **out)
{
    if (!out)
        return BQ_ERR_BAD_ARG;
    *out = nullptr;
    BQCHK(check_dims(c, d, n));
    BQCHK(check_w(c, d, h, w, s));
    if (!x || !y)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    HIPCHK(c, hipSetDevice(c->device));
    bq_fit *f = new (std::nothrow) bq_fit();
    if (!f)
        return fail(c, BQ_ERR_NOMEM, "out of host memory");
    f->d = (int)d;
    f->n = (int)n;
    f->L = make_layout((int)n, 0, true);
    f->npad = f->L.npad;
    f->ldl = pick_ld(f->L.ntot);
    f->h = h;
    f->s = s;
    for (int k = 0; k < d; ++k)
        f->w[k] = w[k];
    f->g = make_params((int)d, h, w, s);
    hipError_t e = hipSuccess;
    auto A = [&](DevBuf &b, size_t bytes) {
        if (e == hipSuccess)
            e = b.alloc(bytes);
    };
    A(f->A, sizeof(double) * (size_t)f->ldl * f->L.ntot);
    A(f->pts, sizeof(double) * (size_t)d * f->L.ntot);
    A(f->y, sizeof(double) * (size_t)f->npad);
    A(f->gp, sizeof(GaussParams));
    A(f->dinv, sizeof(double) * ((size_t)f->npad + BQ_DINV_STRIDE));
    A(f->panel, panel_ws_useful(c, f->L.ntot, 1) ? sizeof(double) * panel_ws_doubles(f->L.ntot, 1)
                                                 : 0);
    A(f->dw, sizeof(double) * BQ_DINV_HALF * (size_t)(f->npad / 64));
    A(f->misc, sizeof(double) * (8 + 128));
    A(f->alpha, sizeof(double) * (size_t)f->npad);
    if (e != hipSuccess) {
        delete f;
        return fail(c, e == hipErrorOutOfMemory ? BQ_ERR_NOMEM : BQ_ERR_HIP,
                    "fit allocation failed: %s", hipGetErrorString(e));
    }
    int st = BQ_OK;
    auto H = [&](hipError_t err) {
        if (st == BQ_OK && err != hipSuccess)
            st = fail(c, BQ_ERR_HIP, "fit upload failed: %s", hipGetErrorString(err));
    };
    H(hipMemsetAsync(f->pts.p, 0, f->pts.bytes, c->stream));
    H(hipMemsetAsync(f->y.p, 0, f->y.bytes, c->stream));
    H(hipMemcpyAsync(f->pts.p, x, sizeof(double) * d * n, hipMemcpyHostToDevice, c->stream));
    H(hipMemcpyAsync(f->y.p, y, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    if (st == BQ_OK)
        st = fit_factor(c, f);
    if (st != BQ_OK) {
        (void)hipStreamSynchronize(c->stream);
        delete f;
        return st;
    }
    *out = f;
    return BQ_OK;
}

extern "C" int bq_gp_refit(bq_ctx *c, bq_fit *f, double h, const double *w, double s)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!f)
        return fail(c, BQ_ERR_BAD_ARG, "null fit handle");
    BQCHK(check_w(c, f->d, h, w, s));
    HIPCHK(c, hipSetDevice(c->device));
    f->h = h;
    f->s = s;
    for (int k = 0; k < f->d; ++k)
        f->w[k] = w[k];
    f->g = make_params(f->d, h, w, s);
    f->L = make_layout(f->n, 0, true);
    return fit_factor(c, f);
}

// New targets for the same points: the hyper-parameter loop hands GP2 new targets l_sc =
// [l_s, exp(mean of GP1 at the candidates)] on every evaluation (bq.py:948-954) -- a new fit
// object per evaluation costs 0.10 ms at the reference's sizes and 0.57 ms at N = 1034, a
// refit 0.05 / 0.32.  The fit is invalid until its next bq_gp_refit / bq_gp_refit_predict.
extern "C" int bq_gp_set_y(bq_ctx *c, bq_fit *f, const double *y)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!f)
        return fail(c, BQ_ERR_BAD_ARG, "null fit handle");
    if (!y)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    HIPCHK(c, hipSetDevice(c->device));
    f->valid = false;
    f->have_alpha = false;
    HIPCHK(c, hipMemcpyAsync(f->y.p, y, sizeof(double) * f->n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // y is the caller's buffer
    return BQ_OK;
}

// New hyper-parameters AND the posterior at M points in the same sweep -- the body of the
// hyper-parameter loop (bq.py:933-947: refit GP1, re-predict the candidates' mean and
// variance).  The M points ride as border rows of the fit's own bordered system, in the
// 64-row block that holds the y row anyway: no launch beyond the refit's, where a separate
// bq_gp_predict after a refit first rebuilds the factor's block inverses (N = 1024: 0.75 ms
// for refit + predict, 0.31 for this).  M <= 63; more points take the two-call route.
extern "C" int bq_gp_refit_predict(bq_ctx *c, bq_fit *f, double h, const double *w, double s,
                                   const double *xo, int64_t M, double *mean, double *var)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!f)
        return fail(c, BQ_ERR_BAD_ARG, "null fit handle");
    if (M < 0 || (M > 0 && (!xo || (!mean && !var))))
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (M == 0)
        return bq_gp_refit(c, f, h, w, s);
    if (M > 63) {
        BQCHK(bq_gp_refit(c, f, h, w, s));
        return bq_gp_predict(c, f, xo, M, mean, var, nullptr);
    }
    BQCHK(check_w(c, f->d, h, w, s));
    HIPCHK(c, hipSetDevice(c->device));
    f->h = h;
    f->s = s;
    for (int k = 0; k < f->d; ++k)
        f->w[k] = w[k];
    f->g = make_params(f->d, h, w, s);
    f->L = make_layout(f->n, (int)M, true); // same ntot: the points share the y row's block
    HIPCHK(c, hipMemcpyAsync(f->pts.d() + (size_t)f->d * f->npad, xo, sizeof(double) * f->d * M,
                             hipMemcpyHostToDevice, c->stream));
    double hv[128];
    BQCHK(fit_factor(c, f, f->misc.d() + 8, f->misc.d() + 8 + 64, hv));
    for (int64_t i = 0; i < M; ++i) {
        if (mean)
            mean[i] = hv[i];
        if (var)
            var[i] = hv[64 + i];
    }
    return BQ_OK;
}

extern "C" void bq_fit_destroy(bq_ctx *c, bq_fit *f)
{
    if (!f)
        return;
    if (c) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
    }
    delete f;
}

extern "C" int bq_gp_logml(bq_ctx *c, bq_fit *f, double *out)
{
    BQCHK(check_fit(c, f));
    if (!out)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    *out = f->logml;
    return BQ_OK;
}

extern "C" int bq_gp_get(bq_ctx *c, bq_fit *f, int which, double *out)
{
    BQCHK(check_fit(c, f));
    if (!out)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    HIPCHK(c, hipSetDevice(c->device));
    const int n = f->n;
    switch (which) {
    case 0: { // L, strict upper zeroed
        HIPCHK(c, hipMemcpy2DAsync(out, sizeof(double) * n, f->A.p, sizeof(double) * f->ldl,
                                   sizeof(double) * n, n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int j = 1; j < n; ++j)
            for (int i = 0; i < j; ++i)
                out[i + (size_t)j * n] = 0.0;
        return BQ_OK;
    }
    case 1:
        BQCHK(fit_alpha(c, f));
        HIPCHK(c, hipMemcpyAsync(out, f->alpha.p, sizeof(double) * n, hipMemcpyDeviceToHost,
                                 c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return BQ_OK;
    case 2:
        HIPCHK(c, hipMemcpy2DAsync(out, sizeof(double), f->A.d() + f->L.yrow,
                                   sizeof(double) * f->ldl, sizeof(double), n,
                                   hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return BQ_OK;
    case 3: {
        DevBuf K;
        HIPCHK(c, K.alloc(sizeof(double) * (size_t)n * n));
        BQCHK(launch_gram_sym(c, f->d, f->pts.d(), 0, static_cast<GaussParams *>(f->gp.p), 0,
                              K.d(), n, 0, n, 1));
        HIPCHK(c, hipMemcpyAsync(out, K.p, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost,
                                 c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return BQ_OK;
    }
    default:
        return fail(c, BQ_ERR_BAD_ARG, "unknown item %d", which);
    }
}

extern "C" int bq_gp_predict(bq_ctx *c, bq_fit *f, const double *xo, int64_t M, double *mean,
                             double *var, double *cov)
{
    BQCHK(check_fit(c, f));
    if (M < 0 || (M && !xo))
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (M == 0)
        return BQ_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const int d = f->d, n = f->n, npad = f->npad;
    const int Mp = (int)roundup(M, 64);
    // workspaces live in the fit and only grow: a BQ object predicts thousands of times
    auto grow = [&](DevBuf &b, size_t bytes) -> hipError_t {
        return b.bytes >= bytes ? hipSuccess : b.alloc(bytes);
    };
    DevBuf &xod = f->wx, &out = f->wout;
    HIPCHK(c, grow(xod, sizeof(double) * d * M));
    HIPCHK(c, grow(out, sizeof(double) * 2 * (size_t)Mp));
    HIPCHK(c, hipMemcpyAsync(xod.p, xo, sizeof(double) * d * M, hipMemcpyHostToDevice, c->stream));
    GaussParams g = f->g;
    if (!var && !cov) {
        // mean only: fused cross-Gram x alpha
        BQCHK(fit_alpha(c, f));
        Bracket br(c, BQ_K_REDUCE);
        dim3 grid((unsigned)((M + 3) / 4));
#define PM(D_)                                                                                     \
    hipLaunchKernelGGL(predict_mean_kernel<D_>, grid, dim3(256), 0, c->stream, xod.d(), (int)M,    \
                       f->pts.d(), n, f->alpha.d(), g, out.d())
        switch (d) {
        case 1: PM(1); break;
        case 2: PM(2); break;
        case 3: PM(3); break;
        case 4: PM(4); break;
        case 5: PM(5); break;
        case 6: PM(6); break;
        case 7: PM(7); break;
        default: PM(8); break;
        }
#undef PM
        HIPCHK(c, hipGetLastError());
    } else {
        // V = K(xo, x) L^-T by a forward sweep with rows = prediction points
        WideInv wi;
        BQCHK(fit_wide(c, f, wi));
        DevBuf &V0 = f->wV, &V = f->wV2;
        HIPCHK(c, grow(V0, sizeof(double) * (size_t)Mp * npad));
        HIPCHK(c, grow(V, sizeof(double) * (size_t)Mp * npad));
        HIPCHK(c, hipMemsetAsync(V0.p, 0, sizeof(double) * (size_t)Mp * npad, c->stream));
        BQCHK(launch_gram_cross(c, d, xod.d(), (int)M, f->pts.d(), n, g, V0.d(), Mp));
        BQCHK(enqueue_forward_rows(c, V0.d(), V.d(), Mp, Mp, f->A.d(), f->ldl, npad, wi));
        // z lives in row yrow of the factor with stride ldl: gather it
        DevBuf &z = f->wz;
        HIPCHK(c, grow(z, sizeof(double) * npad));
        HIPCHK(c, hipMemcpy2DAsync(z.p, sizeof(double), f->A.d() + f->L.yrow,
                                   sizeof(double) * f->ldl, sizeof(double), npad,
                                   hipMemcpyDeviceToDevice, c->stream));
        {
            Bracket br(c, BQ_K_REDUCE);
            hipLaunchKernelGGL(rowdot_kernel, dim3(Mp / 16), dim3(1024), 0, c->stream, V.d(),
                               (long)Mp, (int)M, npad, z.d(), g.c, out.d(), out.d() + Mp);
            HIPCHK(c, hipGetLastError());
        }
        if (cov) {
            // cov = K(xo,xo) - V V^T  (Mp x Mp on device, M x M out)
            DevBuf Cd, gd;
            HIPCHK(c, Cd.alloc(sizeof(double) * (size_t)Mp * Mp));
            HIPCHK(c, gd.alloc(sizeof(GaussParams)));
            GaussParams g0 = g;
            g0.s2 = 0.0;
            HIPCHK(c, hipMemsetAsync(Cd.p, 0, Cd.bytes, c->stream));
            HIPCHK(c, hipMemcpyAsync(gd.p, &g0, sizeof g0, hipMemcpyHostToDevice, c->stream));
            BQCHK(launch_gram_sym(c, d, xod.d(), 0, static_cast<GaussParams *>(gd.p), 0, Cd.d(),
                                  Mp, 0, (int)M, 1));
            BQCHK(launch_gemm(c, BQ_K_GEMM, Cd.d(), Mp, 0, V.d(), Mp, 0, V.d(), 1, Mp, 0, Mp, Mp,
                              npad, 0, 1));
            HIPCHK(c, hipMemcpy2DAsync(cov, sizeof(double) * M, Cd.p, sizeof(double) * Mp,
                                       sizeof(double) * M, M, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream)); // Cd goes out of scope
    }
    if (mean)
        HIPCHK(c, hipMemcpyAsync(mean, out.p, sizeof(double) * M, hipMemcpyDeviceToHost,
                                 c->stream));
    if (var)
        HIPCHK(c, hipMemcpyAsync(var, out.d() + Mp, sizeof(double) * M, hipMemcpyDeviceToHost,
                                 c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

#include "integrals.inc"

// ===========================================================================
// hardware probes
// ===========================================================================
extern "C" int bq_probe_mfma_f64(bq_ctx *c, double *tflops)
{
    if (!c || !tflops)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf o;
    HIPCHK(c, o.alloc(64));
    const int iters = 4096, blocks = c->cus * 8; // 2 waves per SIMD
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(blocks), dim3(256), 0, c->stream, o.d(), 64);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    BQCHK(bq_timer_start(c));
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(blocks), dim3(256), 0, c->stream, o.d(), iters);
    BQCHK(bq_timer_stop_ms(c, &ms));
    const double flops = (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * (16.0 * 16 * 4 * 2);
    *tflops = flops / (ms * 1e-3) / 1e12;
    return BQ_OK;
}

extern "C" int bq_probe_fma_f64(bq_ctx *c, double *tflops)
{
    if (!c || !tflops)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf o;
    HIPCHK(c, o.alloc(64));
    const int iters = 1 << 16, blocks = c->cus * 8;
    hipLaunchKernelGGL(probe_fma_kernel, dim3(blocks), dim3(256), 0, c->stream, o.d(), 64);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    BQCHK(bq_timer_start(c));
    hipLaunchKernelGGL(probe_fma_kernel, dim3(blocks), dim3(256), 0, c->stream, o.d(), iters);
    BQCHK(bq_timer_stop_ms(c, &ms));
    const double flops = (double)blocks * 256 * (double)iters * 8 * 2;
    *tflops = flops / (ms * 1e-3) / 1e12;
    return BQ_OK;
}

extern "C" int bq_probe_hbm(bq_ctx *c, size_t bytes, double *write_gbs, double *copy_gbs)
{
    if (!c || bytes < 4096)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf a, b;
    HIPCHK(c, a.alloc(bytes));
    HIPCHK(c, b.alloc(bytes));
    const size_t n2 = bytes / 16;
    const int blocks = c->cus * 8;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        BQCHK(bq_timer_start(c));
        for (int i = 0; i < 5; ++i)
            hipLaunchKernelGGL(probe_write_kernel, dim3(blocks), dim3(256), 0, c->stream,
                               static_cast<double2_t *>(a.p), n2);
        BQCHK(bq_timer_stop_ms(c, &ms));
    }
    if (write_gbs)
        *write_gbs = 5.0 * bytes / (ms * 1e-3) / 1e9;
    for (int rep = 0; rep < 2; ++rep) {
        BQCHK(bq_timer_start(c));
        for (int i = 0; i < 5; ++i)
            hipLaunchKernelGGL(probe_copy_kernel, dim3(blocks), dim3(256), 0, c->stream,
                               static_cast<double2_t *>(b.p), static_cast<const double2_t *>(a.p),
                               n2);
        BQCHK(bq_timer_stop_ms(c, &ms));
    }
    if (copy_gbs)
        *copy_gbs = 5.0 * 2.0 * bytes / (ms * 1e-3) / 1e9;
    return BQ_OK;
}

// kind 0: v_mfma_f64_16x16x4_f64, 1: v_mfma_f64_4x4x4_4b_f64; nacc in {1,2,4,8};
// blocks_per_cu 256-thread blocks per CU (= waves per SIMD)
extern "C" int bq_probe_mfma_variant(bq_ctx *c, int kind, int nacc, int blocks_per_cu,
                                     double *tflops)
{
    if (!c || !tflops || blocks_per_cu < 1 || blocks_per_cu > 8)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf o;
    HIPCHK(c, o.alloc(64));
    if (kind >= 2) { // the GEMM inner step, kind 2: no rotations, 3: with rotations
        const int it = 512, blocks = c->cus * blocks_per_cu;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            BQCHK(bq_timer_start(c));
            if (kind == 2)
                hipLaunchKernelGGL(probe_mfma_step_kernel<0>, dim3(blocks), dim3(256), 0,
                                   c->stream, o.d(), it);
            else
                hipLaunchKernelGGL(probe_mfma_step_kernel<1>, dim3(blocks), dim3(256), 0,
                                   c->stream, o.d(), it);
            BQCHK(bq_timer_stop_ms(c, &ms));
        }
        *tflops = (double)blocks * 4 * (double)it * 64 * 512.0 / (ms * 1e-3) / 1e12;
        return BQ_OK;
    }
    const int iters = 8192 / nacc, blocks = c->cus * blocks_per_cu;
    auto launch = [&](int it) {
#define PV(K_, N_)                                                                                 \
    hipLaunchKernelGGL((probe_mfma_var_kernel<K_, N_>), dim3(blocks), dim3(256), 0, c->stream,     \
                       o.d(), it)
        if (kind == 0) {
            switch (nacc) {
            case 1: PV(0, 1); break;
            case 2: PV(0, 2); break;
            case 4: PV(0, 4); break;
            default: PV(0, 8); break;
            }
        } else {
            switch (nacc) {
            case 1: PV(1, 1); break;
            case 2: PV(1, 2); break;
            case 4: PV(1, 4); break;
            default: PV(1, 8); break;
            }
        }
#undef PV
    };
    launch(16);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    BQCHK(bq_timer_start(c));
    launch(iters);
    BQCHK(bq_timer_stop_ms(c, &ms));
    const double per = kind == 0 ? 16.0 * 16 * 4 * 2 : 4.0 * 4 * 4 * 4 * 2;
    const int na = (nacc == 1 || nacc == 2 || nacc == 4) ? nacc : 8;
    *tflops = (double)blocks * 4 * (double)iters * na * per / (ms * 1e-3) / 1e12;
    return BQ_OK;
}

extern "C" int bq_probe_mfma444_layout(bq_ctx *c, int cbsz, int abid, int32_t *out8192)
{
    if (!c || !out8192)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf o;
    HIPCHK(c, o.alloc(8192 * sizeof(int)));
#define PL(C_, A_)                                                                                 \
    hipLaunchKernelGGL((probe_layout444_kernel<C_, A_>), dim3(64, 64), dim3(64), 0, c->stream, o.i())
    if (cbsz == 0) PL(0, 0);
    else if (cbsz == 1 && abid == 0) PL(1, 0);
    else if (cbsz == 1) PL(1, 1);
    else if (abid == 0) PL(2, 0);
    else if (abid == 1) PL(2, 1);
    else if (abid == 2) PL(2, 2);
    else PL(2, 3);
#undef PL
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out8192, o.p, 8192 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

extern "C" int bq_probe_exp(bq_ctx *c, const double *x, int64_t n, double *out)
{
    if (!c || !x || !out || n < 1 || n > (1 << 28))
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf xd, od;
    HIPCHK(c, xd.alloc(sizeof(double) * n));
    HIPCHK(c, od.alloc(sizeof(double) * n));
    HIPCHK(c, hipMemcpyAsync(xd.p, x, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(probe_exp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       xd.d(), od.d(), (int)n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, od.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

extern "C" int bq_probe_rsq(bq_ctx *c, const double *x, int64_t n, double *err3)
{
    if (!c || !x || !err3 || n < 1)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf xd, od;
    HIPCHK(c, xd.alloc(sizeof(double) * n));
    HIPCHK(c, od.alloc(sizeof(double) * 3 * n));
    HIPCHK(c, hipMemcpyAsync(xd.p, x, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(probe_rsq_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       xd.d(), od.d(), (int)n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(err3, od.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

// One eager (not graph-replayed) pass of a plan with the profiling instantiation of the slab
// step: stamps[160 * step + k] = s_memtime of workgroup 0 at (0) entry, (1) factor fragments
// loaded, (2) panel rows solved, (3) tile loaded + Q in LDS, (4) tile updated, (5..9) the
// diagonal factor's entry / block in registers / pivot chain done / sub-blocks in LDS / end.
extern "C" int bq_probe_c2_timeline(bq_ctx *c, bq_plan *p, int64_t *stamps, int64_t nsteps)
{
    if (!c || !p || !stamps || nsteps < 1 || nsteps > 1024)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf st;
    HIPCHK(c, st.alloc(sizeof(long long) * 160 * (size_t)nsteps));
    HIPCHK(c, hipMemsetAsync(st.p, 0, st.bytes, c->stream));
    c->stamp_buf = static_cast<long long *>(st.p);
    int rc = plan_enqueue(c, p);
    c->stamp_buf = nullptr;
    BQCHK(rc);
    HIPCHK(c, hipMemcpyAsync(stamps, st.p, st.bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

// The diagonal factor alone: A (64 x 64 host, column-major) is factored `reps` times from a
// resident copy; L_out / dinv_out (BQ_DINV_HALF doubles) / info_out are the last launch's
// results, us_per_launch the HIP-event average, stamps5 the in-kernel s_memtime stamps
// (entry, block loaded, pivot chain done, sub-blocks in LDS, end; shader cycles) followed at
// [8 + 2 (4 P + w) + k] by wave w's arrival at (k = 0) / release from (k = 1) the barrier that
// publishes panel P: 136 values.
extern "C" int bq_probe_potf2(bq_ctx *c, const double *A, int from_lds, int64_t reps,
                              double *L_out, double *dinv_out, int32_t *info_out,
                              double *us_per_launch, int64_t *stamps5)
{
    if (!c || !A || reps < 1)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf ain, a, dv, inf, st;
    HIPCHK(c, ain.alloc(sizeof(double) * 4096));
    HIPCHK(c, a.alloc(sizeof(double) * 4096));
    HIPCHK(c, dv.alloc(sizeof(double) * BQ_DINV_STRIDE));
    HIPCHK(c, inf.alloc(64));
    HIPCHK(c, st.alloc(sizeof(long long) * 136));
    HIPCHK(c, hipMemcpyAsync(ain.p, A, sizeof(double) * 4096, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(inf.p, 0, 64, c->stream));
    HIPCHK(c, hipMemsetAsync(a.p, 0, sizeof(double) * 4096, c->stream));
    auto launch = [&]() {
        hipLaunchKernelGGL(potf2_probe_kernel, dim3(1), dim3(256), 0, c->stream, ain.d(), a.d(),
                           64L, dv.d(), inf.i(), static_cast<long long *>(st.p), from_lds);
    };
    for (int i = 0; i < 5; ++i)
        launch();
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemsetAsync(inf.p, 0, 64, c->stream));
    float ms = 0;
    BQCHK(bq_timer_start(c));
    for (int64_t i = 0; i < reps; ++i)
        launch();
    BQCHK(bq_timer_stop_ms(c, &ms));
    HIPCHK(c, hipGetLastError());
    if (us_per_launch)
        *us_per_launch = ms * 1e3 / (double)reps;
    if (L_out)
        HIPCHK(c, hipMemcpyAsync(L_out, a.p, sizeof(double) * 4096, hipMemcpyDeviceToHost,
                                 c->stream));
    if (dinv_out)
        HIPCHK(c, hipMemcpyAsync(dinv_out, dv.p, sizeof(double) * BQ_DINV_HALF,
                                 hipMemcpyDeviceToHost, c->stream));
    if (info_out)
        HIPCHK(c, hipMemcpyAsync(info_out, inf.p, sizeof(int32_t), hipMemcpyDeviceToHost,
                                 c->stream));
    if (stamps5)
        HIPCHK(c, hipMemcpyAsync(stamps5, st.p, sizeof(int64_t) * 136, hipMemcpyDeviceToHost,
                                 c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

extern "C" int bq_probe_launch(bq_ctx *c, int64_t n, double *us_per_launch)
{
    if (!c || !us_per_launch || n < 1)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf o;
    HIPCHK(c, o.alloc(64));
    for (int i = 0; i < 10; ++i)
        hipLaunchKernelGGL(probe_empty_kernel, dim3(1), dim3(64), 0, c->stream, o.d());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    BQCHK(bq_timer_start(c));
    for (int64_t i = 0; i < n; ++i)
        hipLaunchKernelGGL(probe_empty_kernel, dim3(1), dim3(64), 0, c->stream, o.d());
    BQCHK(bq_timer_stop_ms(c, &ms));
    *us_per_launch = ms * 1e3 / (double)n;
    return BQ_OK;
}

extern "C" int bq_probe_mfma_layout(bq_ctx *c, double *out256)
{
    if (!c || !out256)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf o;
    HIPCHK(c, o.alloc(256 * sizeof(double)));
    hipLaunchKernelGGL(probe_layout_kernel, dim3(1), dim3(64), 0, c->stream, o.d());
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out256, o.p, 256 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}
