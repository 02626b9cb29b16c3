// fit.hip -- device-resident GP fits (bq_fit): factor, z = L^-1 y, log-ML; alpha, the block
// inverses and the sweep workspaces on demand (gp.GP.Lxx / inv_Kxx_y / log_lh / mean / cov:
// bq.py:200,227-228,282,334-335,546,942-943).
#include "host.h"

// ===========================================================================
// GP fit objects
// ===========================================================================
namespace bqh {

// pm / pv: device buffers for the posterior mean / variance of the layout's M border points
// (bq_gp_refit_predict), or null
// npts_words: that many doubles of border points wait in the staging buffer (hfit + HF_PTS) for
// their place behind the fit's own points
static int fit_factor(bq_ctx *c, bq_fit *f, double *pm = nullptr, double *pv = nullptr,
                      double *hpost = nullptr, // hpost: host copy of misc[8 .. 8 + 128) on return
                      size_t npts_words = 0)
{
    const int ntot = f->L.ntot;
    int *info = f->misc.i();
    double *scal = f->misc.d() + 2;
    f->valid = false;
    f->stale = false;
    f->have_alpha = false;
    f->have_zc = false;
    f->have_wide = false;
    f->have_dw = false;
    // pinned staging: [0, 136) results, then the kernel parameters, then border points
    constexpr size_t HF_PAR = 8 + 128, HF_PTS = HF_PAR + (sizeof(GaussParams) + 7) / 8;
    if (!f->hfit)
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&f->hfit),
                                sizeof(double) * (HF_PTS + 64 * BQ_MAXD)));
    std::memcpy(f->hfit + HF_PAR, &f->g, sizeof f->g);
    // The call's small transfers -- kernel parameters (and border points) in, the record out -- go
    // through one kernel each on the mapped staging buffer: a copy-engine operation costs the
    // stream 8-9 us, a further kernel 2.9 (tools/stream_ops_bench.hip); a refit at the reference's
    // own sizes is 40 us in all.
    double *hmap = nullptr;
    if (c->solve_kcopy)
        HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&hmap), f->hfit, 0));
    static_assert(sizeof(GaussParams) % 8 == 0, "GaussParams is copied in 8-byte words");
    if (hmap) {
        BQCHK(launch_copy_words2(c, f->gp.p, hmap + HF_PAR, sizeof f->g / 8,
                                 f->pts.d() + (size_t)f->d * f->npad, hmap + HF_PTS, npts_words));
    } else {
        HIPCHK(c, hipMemcpyAsync(f->gp.p, f->hfit + HF_PAR, sizeof f->g, hipMemcpyHostToDevice,
                                 c->stream));
        if (npts_words)
            HIPCHK(c, hipMemcpyAsync(f->pts.d() + (size_t)f->d * f->npad, f->hfit + HF_PTS,
                                     sizeof(double) * npts_words, hipMemcpyHostToDevice,
                                     c->stream));
    }
    double *scratch = f->dinv.d() + f->npad;
    FirstStep fs;
    const bool fuse = sweep_is_slab(c, ntot, f->npad, 1, f->panel.bytes / sizeof(double));
    if (fuse) {
        fs.S0 = f->panel.d();
        fs.lds = ntot;
        fs.sstride = 64L * ntot;
        fs.dinv = scratch;
        fs.info = info;
        fs.scal = c->fold_readout ? scal : nullptr;
    } else {
        HIPCHK(c, hipMemsetAsync(info, 0, sizeof(int), c->stream));
    }
    BQCHK(launch_assemble(c, f->d, f->pts.d(), 0, f->y.d(), 0, static_cast<GaussParams *>(f->gp.p),
                          0, f->A.d(), f->ldl, 0, f->L, 1, fs));
    const bool folded = fuse && f->L.yrow >= 0 && c->fold_readout;
    if (folded) // the one-launch sweep carries the read-out (SlabOut)
        c->slab_out = SlabOut{scal, pm, pv, 64L, f->L.n, f->L.npad, f->L.M, f->L.yrow};
    const int st_sweep = enqueue_potrf_partial(c, f->A.d(), f->ldl, 0, 1, ntot, f->npad, scratch,
                                               info, f->panel.d(),
                                               f->panel.bytes / sizeof(double), fuse);
    c->slab_out = SlabOut{};
    BQCHK(st_sweep);
    if (!folded)
        BQCHK(launch_finalize(c, f->A.d(), f->ldl, 0L, f->L, scal, pm, pv, 64L, 1));
    // one read-back: misc = [info (int, 8 bytes) | pad | scal[4] | pad | mean[64] | var[64]]
    double *hm = f->hfit;
    if (hmap)
        BQCHK(launch_copy_words2(c, hmap, f->misc.p, hpost ? 8 + 128 : 6, nullptr, nullptr, 0));
    else
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
    if (f->stale)
        return fail(c, BQ_ERR_BAD_ARG,
                    "fit has new targets (bq_gp_set_y): refit required before its next use");
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
        BQCHK(launch_diag_winv(c, f->A.d(), f->ldl, f->npad, f->dw.d()));
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
    // x | y | the one-launch sweeps' workspace (ticket + x versions) | a solve's second vector
    // and second workspace (bq_gp_solve: both sweeps' slots are filled in one launch)
    const size_t need = 3 * (size_t)f->npad + 2 * trsv_flow_ws_doubles(f->npad, wide_block(f->npad));
    if (f->vec.bytes < sizeof(double) * need)
        HIPCHK(c, f->vec.alloc(sizeof(double) * need));
    if (!f->hvec)
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&f->hvec),
                                sizeof(double) * (size_t)f->npad));
    return BQ_OK;
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
    BQCHK(with_flow_fallback(c, [&]() -> int {
        BQCHK(launch_gather_row(c, f->vec.d(), f->A.d() + f->L.yrow, f->ldl, f->npad));
        BQCHK(fit_replay(c, f, 1, [&]() -> int {
            return enqueue_backward_vec(c, f->vec.d(), f->alpha.d(), f->A.d(), f->ldl, f->npad, w,
                                        f->vec.d() + 2 * (size_t)f->npad);
        }));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return BQ_OK;
    }));
    f->have_alpha = true;
    return BQ_OK;
}

} // namespace bqh

using namespace bqh;

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
// refit 0.05 / 0.32.  Until its next bq_gp_refit / bq_gp_refit_predict every consumer of the
// fit returns BQ_ERR_BAD_ARG ("refit required") -- not BQ_ERR_NOT_PD: nothing failed.
extern "C" int bq_gp_set_y(bq_ctx *c, bq_fit *f, const double *y)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!f)
        return fail(c, BQ_ERR_BAD_ARG, "null fit handle");
    if (!y)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    HIPCHK(c, hipSetDevice(c->device));
    f->stale = true; // the factor is intact, z / alpha / log-ML belong to the old targets
    f->have_alpha = false;
    f->have_zc = false;
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
    {
        constexpr size_t HF_PTS = 8 + 128 + (sizeof(GaussParams) + 7) / 8;
        if (!f->hfit)
            HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&f->hfit),
                                    sizeof(double) * (HF_PTS + 64 * BQ_MAXD)));
        std::memcpy(f->hfit + HF_PTS, xo, sizeof(double) * f->d * M);
    }
    double hv[128];
    BQCHK(fit_factor(c, f, f->misc.d() + 8, f->misc.d() + 8 + 64, hv, (size_t)f->d * (size_t)M));
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
    // points up and results down through pinned staging (asynchronous for real)
    const size_t need = (size_t)d * M + 2 * (size_t)Mp;
    if (f->hio_len < need) {
        if (f->hio)
            (void)hipHostFree(f->hio);
        f->hio = nullptr;
        f->hio_len = 0;
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&f->hio), sizeof(double) * need));
        f->hio_len = need;
    }
    std::memcpy(f->hio, xo, sizeof(double) * d * M);
    // (points in and results out through kernels on the mapped staging buffer: a copy-engine
    // operation costs the stream 8-9 us, a further kernel 2.9 -- tools/stream_ops_bench.hip)
    double *hmap = nullptr;
    if (c->solve_kcopy)
        HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&hmap), f->hio, 0));
    // mean + variance without the covariance (BQ.l_var, _set_gp_log_l_params: bq.py:227-228,
    // 942-943) is six launches: the cross Gram reads the points out of the mapped staging buffer
    // and writes its own zero padding, the row reductions write the results into it (round 5:
    // flow_in, the memset and flow_out gone, 59 -> 48 us of kernels at N = 1024, M = 256)
    // (every workgroup column of the cross Gram re-reads the points over PCIe: npad / 64 uncached
    // passes over d M doubles.  Measured a win at N = 1024, M = 256 / 1000; above a few MB of such
    // reads the points go down once through flow_in instead -- ADVICE r05)
    const bool direct = hmap && var && !cov &&
                        (size_t)d * M * sizeof(double) * (size_t)(npad / 64) <= ((size_t)4 << 20);
    if (direct)
        ;
    else if (hmap)
        BQCHK(launch_flow_in(c, hmap, d * (int)M, xod.d(), d * (int)M, nullptr, 0));
    else
        HIPCHK(c, hipMemcpyAsync(xod.p, f->hio, sizeof(double) * d * M, hipMemcpyHostToDevice,
                                 c->stream));
    GaussParams g = f->g;
    if (!var && !cov) {
        // mean only: fused cross-Gram x alpha
        BQCHK(fit_alpha(c, f));
        BQCHK(launch_predict_mean(c, d, xod.d(), (int)M, f->pts.d(), n, f->alpha.d(), g, out.d()));
    } else {
        // V = K(xo, x) L^-T by a forward sweep with rows = prediction points
        WideInv wi;
        BQCHK(fit_wide(c, f, wi));
        DevBuf &V0 = f->wV, &V = f->wV2;
        HIPCHK(c, grow(V0, sizeof(double) * (size_t)Mp * npad));
        HIPCHK(c, grow(V, sizeof(double) * (size_t)Mp * npad));
        if (direct) {
            BQCHK(launch_gram_cross_pad(c, d, hmap, (int)M, Mp, f->pts.d(), n, npad, g, V0.d(), Mp));
        } else {
            HIPCHK(c, hipMemsetAsync(V0.p, 0, sizeof(double) * (size_t)Mp * npad, c->stream));
            BQCHK(launch_gram_cross(c, d, xod.d(), (int)M, f->pts.d(), n, g, V0.d(), Mp));
        }
        BQCHK(enqueue_forward_rows(c, V0.d(), V.d(), Mp, Mp, f->A.d(), f->ldl, npad, wi));
        // z lives in row yrow of the factor with stride ldl: gathered once per (re)fit
        if (!f->have_zc) {
            HIPCHK(c, grow(f->wz, sizeof(double) * (size_t)npad));
            BQCHK(launch_gather_row(c, f->wz.d(), f->A.d() + f->L.yrow, f->ldl, npad));
            f->have_zc = true;
        }
        double *omean = direct ? hmap + (size_t)d * M : out.d();
        BQCHK(launch_rowdot(c, V.d(), (long)Mp, (int)M, Mp, npad, f->wz.d(), g.c, omean, omean + Mp,
                            1));
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
            HIPCHK(c, hipStreamSynchronize(c->stream)); // Cd goes out of scope
        }
    }
    double *hres = f->hio + (size_t)d * M; // [mean (Mp) | var (Mp)], one copy
    if (direct)
        ;
    else if (hmap)
        BQCHK(launch_flow_out(c, out.d(), 2 * Mp, hmap + (size_t)d * M));
    else
        HIPCHK(c, hipMemcpyAsync(hres, out.p, sizeof(double) * 2 * (size_t)Mp,
                                 hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (mean)
        std::memcpy(mean, hres, sizeof(double) * M);
    if (var)
        std::memcpy(var, hres + Mp, sizeof(double) * M);
    return BQ_OK;
}
