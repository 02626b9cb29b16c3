// ctx.hip -- contexts, device memory, timers and the launch profiler of the C ABI.
#include "host.h"

namespace bqh {

int prof_collect(bq_ctx *c)
{
    if (c->prof_events.empty())
        return BQ_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->aux)
        HIPCHK(c, hipStreamSynchronize(c->aux));
    // the timeline's origin: the start of the first launch bracketed since the recording was
    // armed -- kept across drains (profile_read / _reset / _enable in between), so every row
    // is measured against the same instant
    if (c->prof_keep_timeline && !c->prof_origin)
        c->prof_origin = c->prof_events.front().a;
    for (auto &e : c->prof_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            c->prof_ms[e.cls] += ms;
            c->prof_n[e.cls] += 1;
            c->prof_work[e.cls] += e.work;
            if (c->prof_keep_timeline) {
                float t0 = 0.f;
                (void)hipEventElapsedTime(&t0, c->prof_origin, e.a);
                c->prof_timeline.insert(c->prof_timeline.end(),
                                        {(double)e.cls, (double)e.on_aux, (double)t0,
                                         (double)t0 + ms, e.work});
            }
        }
    }
    for (auto &e : c->prof_events) {
        if (e.a != c->prof_origin)
            (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    c->prof_events.clear();
    return BQ_OK;
}

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


} // namespace bqh

using namespace bqh;

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
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_top, hipEventDisableTiming));

    int lo = 0, hi = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(c, hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, hi));
    BQCHK(gemm_init(c));
    // the one-launch single-vector sweeps: the abort word lives in mapped
    // host memory (the host reads it after a synchronisation without a copy)
    {
        void *h = nullptr;
        HIPCHK(c, hipHostMalloc(&h, 64, hipHostMallocMapped));
        c->flow_abort = static_cast<int *>(h);
        *c->flow_abort = 0;
    }
    if (const char *e = std::getenv("BQ_TRSV_FLOW"))
        c->trsv_flow = std::atoi(e);
    if (const char *e = std::getenv("BQ_TRSV_FLOW_MIN"))
        c->trsv_flow_min = std::atoi(e);
    if (const char *e = std::getenv("BQ_LOOKAHEAD"))
        c->lookahead = std::atoi(e);
    if (const char *e = std::getenv("BQ_SPLIT"))
        c->split_batch = std::atoi(e);
    if (const char *e = std::getenv("BQ_DIAG_FIRST"))
        c->diag_first = std::atoi(e);
    if (const char *e = std::getenv("BQ_ASM_FUSE"))
        c->asm_fuse = std::atoi(e);
    if (const char *e = std::getenv("BQ_DF_SWEEP"))
        c->df_sweep = std::atoi(e);
    if (const char *e = std::getenv("BQ_DF_WG"))
        c->df_wg = std::atoi(e);
    if (const char *e = std::getenv("BQ_PAIR_BORDER"))
        c->pair_border = std::atoi(e);
    if (const char *e = std::getenv("BQ_DF_WG_ROWS"))
        c->df_wg_rows = std::atoi(e);
    if (const char *e = std::getenv("BQ_DF_EARLY"))
        c->df_early = std::atoi(e);
    if (const char *e = std::getenv("BQ_ROWS_TAIL"))
        c->rows_tail = std::atoi(e);
    if (const char *e = std::getenv("BQ_SOLVE_KCOPY"))
        c->solve_kcopy = std::atoi(e);
    if (const char *e = std::getenv("BQ_DF_SHARING"))
        c->df_sharing = std::atoi(e);
    if (const char *e = std::getenv("BQ_LA_MIN"))
        c->la_min = std::atoi(e);
    if (const char *e = std::getenv("BQ_GEMM_LDS"))
        c->gemm_lds = std::atoi(e);
    if (const char *e = std::getenv("BQ_GEMM_LDS64"))
        c->gemm_lds64 = std::atoi(e);
    if (const char *e = std::getenv("BQ_SLAB_NB_MAX"))
        c->slab_nb_max = std::atoi(e);
    if (const char *e = std::getenv("BQ_SLAB_MAX"))
        c->slab_max = std::atoi(e);
    if (const char *e = std::getenv("BQ_FOLD_READOUT"))
        c->fold_readout = std::atoi(e);
    if (const char *e = std::getenv("BQ_POTF2_8W"))
        c->potf2_8w = std::atoi(e);
    if (const char *e = std::getenv("BQ_GEMM_KSPLIT"))
        c->gemm_ksplit = std::atoi(e);
    if (const char *e = std::getenv("BQ_GEMM_TILE"))
        c->gemm_tile = std::atoi(e);
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
        if (e.a != c->prof_origin)
            (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    if (c->prof_origin)
        (void)hipEventDestroy(c->prof_origin);
    if (c->t0)
        (void)hipEventDestroy(c->t0);
    if (c->t1)
        (void)hipEventDestroy(c->t1);
    if (c->hstage)
        (void)hipHostFree(c->hstage);
    c->hstage = nullptr;
    for (hipEvent_t e : {c->ev_panel, c->ev_next, c->ev_fork, c->ev_top})
        if (e)
            (void)hipEventDestroy(e);
    if (c->flow_abort)
        (void)hipHostFree(c->flow_abort);
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
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return flow_check(c);
}

extern "C" const char *bq_last_error(const bq_ctx *c) { return c ? c->err : "null context"; }

extern "C" int bq_device_info(bq_ctx *c, char *name, int *cus, size_t *hbm_bytes, int *clock_khz)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
    if (name) {
        // (on the GPU box the marketing name comes back empty: say what is known)
        std::snprintf(name, 64, "%s (%s)", prop.name[0] ? prop.name : "AMD GPU", prop.gcnArchName);
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

extern "C" int bq_get_config(bq_ctx *c, int *nb, int *lookahead, int *min_rows)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (nb)
        *nb = c->nb_override;
    if (lookahead)
        *lookahead = c->lookahead;
    if (min_rows)
        *min_rows = c->la_min;
    return BQ_OK;
}

extern "C" int bq_ctx_stats(bq_ctx *c, int64_t *out, int n)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!out || n < 0)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    for (int i = 0; i < n; ++i)
        out[i] = 0;
    if (n > 0)
        out[0] = c->n_flow_fallback;
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
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->t0, c->stream));
    return BQ_OK;
}

extern "C" int bq_timer_stop_ms(bq_ctx *c, float *ms)
{
    if (!c || !ms)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
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

// The launches bracketed since the recording was armed, as a timeline: rows of {class, stream
// (0 main, 1 second), start ms, end ms, algorithmic work}, times since the first launch's start.
// keep = 1 and out = NULL arms (and clears) the recording; keep = 0 stops it: with out = NULL
// that call only reports *nrows, with out it copies up to max_rows rows.
extern "C" int bq_profile_timeline(bq_ctx *c, int keep, double *out, int64_t max_rows,
                                   int64_t *nrows)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    BQCHK(prof_collect(c));
    if ((keep && !out) || !keep) {
        // (re)armed or stopped: the next recording starts from its own first launch
        if (c->prof_origin)
            (void)hipEventDestroy(c->prof_origin);
        c->prof_origin = nullptr;
    }
    if (keep && !out)
        c->prof_timeline.clear();
    const int64_t n = (int64_t)c->prof_timeline.size() / 5;
    if (nrows)
        *nrows = n;
    if (out && max_rows > 0)
        std::memcpy(out, c->prof_timeline.data(),
                    sizeof(double) * 5 * (size_t)std::min<int64_t>(n, max_rows));
    c->prof_keep_timeline = keep != 0;
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
