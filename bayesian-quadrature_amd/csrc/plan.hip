// plan.hip -- resident batched "fit + posterior + log-ML" plans and the batched / hyper-grid
// entry points built on them (bq_plan_*, bq_batch_fit_predict, bq_fit_predict,
// bq_gp_logml_grid).
#include "host.h"

using namespace bqh;

// ===========================================================================
// plans: resident batched "fit + posterior + log-ML"
// ===========================================================================
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
    A(p->panel, sizeof(double) * sweep_ws_doubles(c, p->L.ntot, (int)nprob));
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

static void plan_drop_graph(bq_plan *p);

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

extern "C" int bq_set_guard(int on)
{
    devbuf_guard() = on != 0;
    return BQ_OK;
}

extern "C" int bq_plan_check_guards(bq_ctx *c, bq_plan *p, int64_t *guarded, int64_t *damaged)
{
    if (!c || !p || !guarded || !damaged)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *guarded = *damaged = 0;
    for (const DevBuf *b : {&p->A, &p->pts, &p->y, &p->gp, &p->dinv, &p->panel, &p->info, &p->scal,
                            &p->mean, &p->var}) {
        if (!b->guard)
            continue;
        ++*guarded;
        const long bad = b->guard_damage();
        if (bad < 0)
            return fail(c, BQ_ERR_HIP, "guard band read-back failed");
        *damaged += bad;
    }
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
    // A small plan (below 256 KB of inputs: a latency-bound call): everything through ONE mapped
    // pinned staging buffer and one kernel that puts it in place -- four copy operations out of
    // pageable memory cost the call 40 us.  The buffer is the plan's; it is rewritten by the next
    // call only, which waits for this scatter first (in_flight).
    static_assert(sizeof(GaussParams) % 8 == 0, "GaussParams is copied in 8-byte words");
    constexpr size_t GW = sizeof(GaussParams) / 8;
    const size_t words = (size_t)p->nprob * (GW + (size_t)d * n + (size_t)d * M + (size_t)n);
    if (c->solve_kcopy && words * 8 <= (256u << 10)) {
        if (p->in_flight)
            HIPCHK(c, hipStreamSynchronize(c->stream));
        p->in_flight = false;
        if (p->hin_len < words) {
            if (p->hin)
                (void)hipHostFree(p->hin);
            p->hin = nullptr;
            p->hin_len = 0;
            HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&p->hin), sizeof(double) * words));
            p->hin_len = words;
        }
        double *q = p->hin;
        std::memcpy(q, p->hgp.data(), sizeof(GaussParams) * p->nprob);
        q += GW * p->nprob;
        std::memcpy(q, x, sizeof(double) * (size_t)d * n * p->nprob);
        q += (size_t)d * n * p->nprob;
        if (M > 0)
            std::memcpy(q, xo, sizeof(double) * (size_t)d * M * p->nprob);
        q += (size_t)d * M * p->nprob;
        std::memcpy(q, y, sizeof(double) * (size_t)n * p->nprob);
        double *hmap = nullptr;
        HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&hmap), p->hin, 0));
        BQCHK(launch_plan_scatter(c, hmap, p->nprob, d, n, M, ntot, npad, (int)GW, p->gp.d(),
                                  p->pts.d(), p->y.d()));
        p->in_flight = true;
        p->has_inputs = true;
        return BQ_OK;
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

namespace bqh {

// New (h, w, s) for every problem of a plan whose points and targets stay resident (the
// hyper-parameter loop: same data, new parameters on every evaluation)
int plan_set_params(bq_ctx *c, bq_plan *p, const double *h, const double *w, const double *s)
{
    if (!p || !p->has_inputs)
        return fail(c, BQ_ERR_BAD_ARG, "plan has no inputs");
    const int d = p->d;
    p->hgp.resize(p->nprob);
    for (int b = 0; b < p->nprob; ++b) {
        BQCHK(check_w(c, d, h[b], w + (size_t)b * d, s[b]));
        p->hgp[b] = make_params(d, h[b], w + (size_t)b * d, s[b]);
    }
    // (hgp outlives the copy: it belongs to the plan)
    HIPCHK(c, hipMemcpyAsync(p->gp.p, p->hgp.data(), sizeof(GaussParams) * p->nprob,
                             hipMemcpyHostToDevice, c->stream));
    return BQ_OK;
}

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
        fs.scal = c->fold_readout ? p->scal.d() : nullptr;
    } else {
        HIPCHK(c, hipMemsetAsync(p->info.p, 0, sizeof(int) * p->nprob, c->stream));
    }
    // a batch that sweeps diagonal block first assembles only its first outer block's columns: the
    // first products that touch the rest compute it themselves (GramSeed, potrf.hip)
    const int jcols = fuse ? 0
                           : dfirst_seed_cols(c, p->L.ntot, p->L.npad, p->nprob,
                                              p->panel.bytes / sizeof(double));
    BQCHK(launch_assemble(c, p->d, p->pts.d(), (long)p->d * p->L.ntot, p->y.d(), p->L.npad,
                          static_cast<GaussParams *>(p->gp.p), 1, p->A.d(), p->lda, p->astride,
                          p->L, p->nprob, fs, jcols));
    if (jcols > 0)
        c->gram_seed = GramSeed{p->pts.d(), (long)p->d * p->L.ntot, p->y.d(), (long)p->L.npad,
                                static_cast<const GaussParams *>(p->gp.p), 1, p->L, p->d, 0, 0};
    // A blocked sweep (outer block >= 128) reads its results off the border rows and skips the
    // border x border block in its trailing updates; the one-launch steps of small systems
    // update everything and read the Schur complement.
    const bool by_rows = !fuse && p->L.yrow >= 0 && auto_nb(c, p->L.ntot, p->nprob) >= 128;
    // a sweep of one-launch steps carries the read-out itself (SlabOut: no finalize launch)
    const bool folded = fuse && p->L.yrow >= 0 && c->fold_readout;
    if (folded)
        c->slab_out = SlabOut{p->scal.d(), p->mean.d(), p->var.d(), (long)std::max(p->M, 1),
                              p->L.n, p->L.npad, p->L.M, p->L.yrow};
    const int st_sweep =
        enqueue_potrf_partial(c, p->A.d(), p->lda, p->astride, p->nprob, p->L.ntot, p->L.npad,
                              p->dinv.d(), p->info.i(), p->panel.d(),
                              p->panel.bytes / sizeof(double), fuse, by_rows);
    c->slab_out = SlabOut{};
    c->gram_seed = GramSeed{};
    BQCHK(st_sweep);
    if (folded)
        return BQ_OK;
    if (by_rows)
        return launch_plan_readout(c, p->A.d(), p->lda, p->astride, p->L,
                                   static_cast<const GaussParams *>(p->gp.p), p->scal.d(),
                                   p->mean.d(), p->var.d(), (long)std::max(p->M, 1), p->nprob);
    return launch_finalize(c, p->A.d(), p->lda, p->astride, p->L, p->scal.d(), p->mean.d(),
                           p->var.d(), (long)std::max(p->M, 1), p->nprob,
                           8.0 * (p->n + 2.0 * p->M) * p->nprob);
}

} // namespace bqh

static void plan_drop_graph(bq_plan *p)
{
    if (p->gexec)
        (void)hipGraphExecDestroy(p->gexec);
    if (p->graph)
        (void)hipGraphDestroy(p->graph);
    p->gexec = nullptr;
    p->graph = nullptr;
}

extern "C" int bq_plan_run(bq_ctx *c, bq_plan *p)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!p)
        return fail(c, BQ_ERR_BAD_ARG, "null plan handle");
    if (!p->has_inputs)
        return fail(c, BQ_ERR_BAD_ARG, "plan has no inputs");
    HIPCHK(c, hipSetDevice(c->device)); // (a context may be driven from any one thread)
    if (c->prof || !c->use_graph || !c->own_stream)
        return plan_enqueue(c, p);
    // settings that change the launch sequence invalidate the captured graph
    if (p->graph_state == 1 && p->graph_key != launch_config_key(c)) {
        plan_drop_graph(p);
        p->graph_state = 0;
    }
    if (p->graph_state == 0) {
        p->graph_key = launch_config_key(c);
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
    HIPCHK(c, hipSetDevice(c->device));
    // through pinned staging: [scal 4 nb | info nb (as doubles' storage) | mean M nb | var M nb]
    const size_t o_info = (size_t)4 * nb, o_mean = o_info + (size_t)nb, o_var = o_mean + (size_t)M * nb;
    if (!p->hres)
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&p->hres),
                                sizeof(double) * (o_var + (size_t)M * nb + 1)));
    double *scal = p->hres;
    int *info = reinterpret_cast<int *>(p->hres + o_info);
    if (c->solve_kcopy && (o_var + (size_t)M * nb) * 8 <= (256u << 10)) {
        // (a small plan's record in one kernel on the mapped staging instead of up to four copies)
        double *hmap = nullptr;
        HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&hmap), p->hres, 0));
        BQCHK(launch_plan_gather(c, hmap, p->scal.d(), p->info.i(), (mean && M) ? p->mean.d() : nullptr,
                                 (var && M) ? p->var.d() : nullptr, nb, M));
    } else {
        HIPCHK(c, hipMemcpyAsync(scal, p->scal.p, sizeof(double) * 4 * nb, hipMemcpyDeviceToHost,
                                 c->stream));
        HIPCHK(c, hipMemcpyAsync(info, p->info.p, sizeof(int) * nb, hipMemcpyDeviceToHost,
                                 c->stream));
        if (mean && M)
            HIPCHK(c, hipMemcpyAsync(p->hres + o_mean, p->mean.p, sizeof(double) * (size_t)M * nb,
                                     hipMemcpyDeviceToHost, c->stream));
        if (var && M)
            HIPCHK(c, hipMemcpyAsync(p->hres + o_var, p->var.p, sizeof(double) * (size_t)M * nb,
                                     hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    p->in_flight = false;
    if (mean && M)
        std::memcpy(mean, p->hres + o_mean, sizeof(double) * (size_t)M * nb);
    if (var && M)
        std::memcpy(var, p->hres + o_var, sizeof(double) * (size_t)M * nb);
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

// One eager (not graph-replayed) pass of a plan with the profiling instantiation of the slab
// step: stamps[160 * step + k] = s_memtime of workgroup 0 at (0) entry, (1) factor fragments
// loaded, (2) panel rows solved, (3) tile loaded + Q in LDS, (4) tile updated, (5..9) the
// diagonal factor's entry / block in registers / pivot chain done / sub-blocks in LDS / end.
extern "C" int bq_probe_c2_timeline(bq_ctx *c, bq_plan *p, int64_t *stamps, int64_t nsteps)
{
    if (!c || !p || !stamps || nsteps < 1 || nsteps > 1024)
        return BQ_ERR_BAD_ARG;
    // only the one-launch slab sweep carries the stamped instantiation, and it stamps one
    // record per step into the caller's nsteps (the sweep itself skips steps beyond them)
    if (!sweep_is_slab(c, p->L.ntot, p->L.npad, p->nprob, p->panel.bytes / sizeof(double)))
        return fail(c, BQ_ERR_BAD_ARG, "timeline: this plan does not sweep with the one-launch steps");
    if (nsteps < p->L.npad / 64)
        return fail(c, BQ_ERR_BAD_ARG, "timeline: %d steps, room for %d", p->L.npad / 64, (int)nsteps);
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf st;
    HIPCHK(c, st.alloc(sizeof(long long) * 160 * (size_t)nsteps));
    HIPCHK(c, hipMemsetAsync(st.p, 0, st.bytes, c->stream));
    c->stamp_buf = static_cast<long long *>(st.p);
    c->stamp_steps = (int)nsteps;
    int rc = plan_enqueue(c, p);
    c->stamp_buf = nullptr;
    c->stamp_steps = 0;
    BQCHK(rc);
    HIPCHK(c, hipMemcpyAsync(stamps, st.p, st.bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}

