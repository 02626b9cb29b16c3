// pair.hip -- the stacked pair of GPs of Bayesian quadrature at S hyper-parameter sets in ONE
// batched pass: GP1 over log l at the samples, GP2 over exp(log l) at samples + candidates.
//
// The reference evaluates the hyper-parameter objective (bq.py:536-550: refit GP1, re-predict the
// candidates, hand GP2 its new targets, refit GP2, add the two log-MLs -- bq.py:933-965) and the
// acquisition under sampled hyper-parameters (marginalize / choose_next, bq.py:604-662) one
// parameter set after the other; on the device each of those is a chain of a few short dependent
// launches.  The S sets are independent, and every kernel of the engine takes a batch in
// blockIdx.z: here they run as
//   stage 1  a bordered plan over S copies of GP1's system, the candidates (and the
//            acquisition points x_a) as border points -> log-ML, mean / variance at x_c, x_a;
//   targets  l_c = exp(mean) with the reference's overflow guard, GP2's targets [l_s, l_c];
//   stage 2  bq_pair_llh: a plan over S copies of GP2's system -> log-ML;
//            bq_pair_esm: the S x Ma bordered (nsc + 1)^2 systems of the acquisition
//            (bq.py:447-527), the reference's own recipe with its jitter, one batch.
#include "host.h"
#include "pair.h"

using namespace bqh;

struct bq_pair {
    int ns = 0, nc = 0, ma = 0, S = 0, nsc = 0;
    bq_plan *p1 = nullptr; // GP1: n = ns, M = nc + ma
    bq_plan *p2 = nullptr; // GP2: n = nsc, M = 0 (the objective; the acquisition has its own systems)
    // nc = 0 (every candidate filtered out: dense samples) and no acquisition points: GP2's
    // targets are the samples' own values, nothing of it depends on GP1, and both have ns
    // points -- ONE plan of 2 S systems (p1; first S: GP1, last S: GP2), one sweep instead of two
    bool merged = false;
    // the whole device side of a bq_pair_llh pass -- parameters up from pinned staging, both
    // plans' launches, targets, the record per set, the record down -- as ONE graph launch
    hipGraph_t lgraph = nullptr;
    hipGraphExec_t lgexec = nullptr;
    int lgraph_state = 0; // 0 = not tried, 1 = ready, -1 = unavailable (eager launches)
    unsigned long long lgraph_key = 0;
    DevBuf l_s, x_sc, x_a, y2, flag;
    // stage 2 of bq_pair_esm, kept between calls (choose_next calls it once per step with the
    // same shapes; allocating and releasing its tens of GB per call costs more than the pass)
    DevBuf gpd, pard, ik, ika, dj1, dj2, Ad, dinv, info, outd, panel;
    int64_t chunk = 0;
    bool esm_small = false; // gpd .. dj2 are allocated
    // the border route (S factorisations + border rows): the sets' systems, the S Ma small ones
    DevBuf bA, bdinv, binfo, bws, sA, sdinv, sinfo, sws, sout;
    int64_t bchunk = 0; // parameter sets per pass
    std::vector<double> hx_s, hx_c, hx_a;
    // bq_pair_llh runs hundreds of times per optimisation / chain on tiny systems, where the
    // host's share of a pass matters: parameters go up from PINNED staging (asynchronous for
    // real), results come back as one record per set
    GaussParams *hpar = nullptr; // 2 S (pinned)
    double *hres = nullptr;      // S (5 + nc) (pinned)
    DevBuf dres;
    ~bq_pair()
    {
        if (lgexec)
            (void)hipGraphExecDestroy(lgexec);
        if (lgraph)
            (void)hipGraphDestroy(lgraph);
        if (hpar)
            (void)hipHostFree(hpar);
        if (hres)
            (void)hipHostFree(hres);
    }
};

namespace {

constexpr double LOG_2PI = 1.8378770664093453;

// log of the largest double the reference lets exp() see: log(2^(maxexp - 4)) (bq.py:14-16)
double max_log() { return std::log(std::exp2((double)(std::numeric_limits<double>::max_exponent - 4))); }

int check_params(bq_ctx *c, const double *p, int64_t S, const char *what)
{
    for (int64_t b = 0; b < S; ++b) {
        const double w[1] = {p[3 * b + 1]};
        if (check_w(c, 1, p[3 * b], w, p[3 * b + 2]) != BQ_OK)
            return fail(c, BQ_ERR_BAD_ARG, "%s: parameter set %d is invalid", what, (int)b);
    }
    return BQ_OK;
}

// stage 1 + targets: leaves mean / var of GP1 at [x_c, x_a] in p1->mean / p1->var, GP2's targets
// in `y2` (stride ystride) and the overflow flags in pr->flag
int run_stage1(bq_ctx *c, bq_pair *pr, const double *p_tl, double *y2, long ystride)
{
    const int S = pr->S;
    std::vector<double> h((size_t)S), w((size_t)S), s((size_t)S);
    for (int b = 0; b < S; ++b) {
        h[(size_t)b] = p_tl[3 * b];
        w[(size_t)b] = p_tl[3 * b + 1];
        s[(size_t)b] = p_tl[3 * b + 2];
    }
    BQCHK(plan_set_params(c, pr->p1, h.data(), w.data(), s.data()));
    BQCHK(bq_plan_run(c, pr->p1));
    HIPCHK(c, hipMemsetAsync(pr->flag.p, 0, sizeof(int) * S, c->stream));
    const long ms = std::max(pr->nc + pr->ma, 1);
    hipLaunchKernelGGL(pair_targets_kernel, dim3((pr->nsc + 255) / 256, S), dim3(256), 0, c->stream,
                       pr->l_s.d(), pr->ns, pr->nc, pr->p1->mean.d(), pr->p1->var.d(), ms,
                       max_log(), y2, ystride, pr->flag.i());
    HIPCHK(c, hipGetLastError());
    return BQ_OK;
}

} // namespace

extern "C" int bq_pair_create(bq_ctx *c, const double *x_s, const double *tl_s, const double *l_s,
                              int64_t ns, const double *x_c, int64_t nc, const double *x_a,
                              int64_t ma, int64_t S, bq_pair **out)
{
    if (!out)
        return BQ_ERR_BAD_ARG;
    *out = nullptr;
    BQCHK(check_dims(c, 1, ns));
    if (!x_s || !tl_s || !l_s || nc < 0 || ma < 0 || S < 1 || S > 65535 || (nc && !x_c) ||
        (ma && !x_a) || nc + ma > (1 << 20))
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    HIPCHK(c, hipSetDevice(c->device));
    bq_pair *pr = new (std::nothrow) bq_pair();
    if (!pr)
        return fail(c, BQ_ERR_NOMEM, "out of host memory");
    pr->ns = (int)ns;
    pr->nc = (int)nc;
    pr->ma = (int)ma;
    pr->S = (int)S;
    pr->nsc = (int)(ns + nc);
    pr->hx_s.assign(x_s, x_s + ns);
    pr->hx_c.assign(x_c, x_c + nc);
    pr->hx_a.assign(x_a, x_a + ma);
    int st = BQ_OK;
    auto H = [&](hipError_t e) {
        if (st == BQ_OK && e != hipSuccess)
            st = fail(c, e == hipErrorOutOfMemory ? BQ_ERR_NOMEM : BQ_ERR_HIP, "pair: %s",
                      hipGetErrorString(e));
    };
    const int M1 = (int)(nc + ma), nsc = pr->nsc;
    std::vector<double> xr((size_t)S * ns), yr((size_t)S * ns), xo((size_t)S * std::max(M1, 1)),
        one((size_t)S, 1.0), zero((size_t)S, 0.0), xsc((size_t)nsc), xr2((size_t)S * nsc),
        yr2((size_t)S * nsc, 0.0);
    for (int64_t b = 0; b < S; ++b) {
        std::memcpy(&xr[(size_t)b * ns], x_s, sizeof(double) * ns);
        std::memcpy(&yr[(size_t)b * ns], tl_s, sizeof(double) * ns);
        if (nc)
            std::memcpy(&xo[(size_t)b * M1], x_c, sizeof(double) * nc);
        if (ma)
            std::memcpy(&xo[(size_t)b * M1 + nc], x_a, sizeof(double) * ma);
    }
    std::memcpy(xsc.data(), x_s, sizeof(double) * ns);
    if (nc)
        std::memcpy(xsc.data() + ns, x_c, sizeof(double) * nc);
    for (int64_t b = 0; b < S; ++b)
        std::memcpy(&xr2[(size_t)b * nsc], xsc.data(), sizeof(double) * nsc);
    pr->merged = nc == 0 && ma == 0;
    if (pr->merged) {
        std::vector<double> x2((size_t)2 * S * ns), y2((size_t)2 * S * ns), one2((size_t)2 * S, 1.0),
            zero2((size_t)2 * S, 0.0);
        for (int64_t b = 0; b < 2 * S; ++b) {
            std::memcpy(&x2[(size_t)b * ns], x_s, sizeof(double) * ns);
            std::memcpy(&y2[(size_t)b * ns], b < S ? tl_s : l_s, sizeof(double) * ns);
        }
        st = bq_plan_create(c, 2 * S, 1, ns, 0, &pr->p1);
        if (st == BQ_OK)
            st = bq_plan_set_inputs(c, pr->p1, x2.data(), y2.data(), nullptr, one2.data(),
                                    one2.data(), zero2.data());
    } else {
        st = bq_plan_create(c, S, 1, ns, M1, &pr->p1);
        if (st == BQ_OK)
            st = bq_plan_set_inputs(c, pr->p1, xr.data(), yr.data(), M1 ? xo.data() : nullptr,
                                    one.data(), one.data(), zero.data());
    }
    if (st == BQ_OK && ma == 0 && !pr->merged) {
        st = bq_plan_create(c, S, 1, nsc, 0, &pr->p2);
        if (st == BQ_OK)
            st = bq_plan_set_inputs(c, pr->p2, xr2.data(), yr2.data(), nullptr, one.data(),
                                    one.data(), zero.data());
    }
    if (st == BQ_OK) {
        H(pr->l_s.alloc(sizeof(double) * ns));
        H(pr->x_sc.alloc(sizeof(double) * nsc));
        H(pr->x_a.alloc(sizeof(double) * std::max<int64_t>(ma, 1)));
        H(pr->flag.alloc(sizeof(int) * S));
        if (ma)
            H(pr->y2.alloc(sizeof(double) * (size_t)S * nsc));
        else {
            H(pr->dres.alloc(sizeof(double) * (size_t)S * (5 + nc)));
            H(hipHostMalloc(reinterpret_cast<void **>(&pr->hpar), sizeof(GaussParams) * 2 * S));
            H(hipHostMalloc(reinterpret_cast<void **>(&pr->hres),
                            sizeof(double) * (size_t)S * (5 + nc)));
        }
    }
    if (st == BQ_OK) {
        H(hipMemcpyAsync(pr->l_s.p, l_s, sizeof(double) * ns, hipMemcpyHostToDevice, c->stream));
        H(hipMemcpyAsync(pr->x_sc.p, xsc.data(), sizeof(double) * nsc, hipMemcpyHostToDevice,
                         c->stream));
        if (ma)
            H(hipMemcpyAsync(pr->x_a.p, x_a, sizeof(double) * ma, hipMemcpyHostToDevice, c->stream));
        H(hipStreamSynchronize(c->stream));
    }
    if (st != BQ_OK) {
        bq_pair_destroy(c, pr);
        return st;
    }
    *out = pr;
    return BQ_OK;
}

extern "C" void bq_pair_destroy(bq_ctx *c, bq_pair *pr)
{
    if (!pr)
        return;
    if (c) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
    }
    if (pr->p1)
        bq_plan_destroy(c, pr->p1);
    if (pr->p2)
        bq_plan_destroy(c, pr->p2);
    delete pr;
}

// the device side of one bq_pair_llh pass (the parameter sets are in pr->hpar)
static int pair_llh_enqueue(bq_ctx *c, bq_pair *pr)
{
    const int S = pr->S, nsc = pr->nsc, ns = pr->ns, nc = pr->nc;
    // (parameters in and the records out through kernels on the mapped pinned staging: a
    // copy-engine operation costs the stream -- or the captured graph -- 8-9 us, a kernel 2.9)
    GaussParams *hpar_d = nullptr;
    double *hres_d = nullptr;
    if (c->solve_kcopy) {
        HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&hpar_d), pr->hpar, 0));
        HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void **>(&hres_d), pr->hres, 0));
    }
    static_assert(sizeof(GaussParams) % 8 == 0, "GaussParams is copied in 8-byte words");
    constexpr size_t GW = sizeof(GaussParams) / 8;
    if (pr->merged) {
        // one plan of 2 S systems: [GP1 under the S sets | GP2 under the S sets]
        if (hpar_d)
            BQCHK(launch_copy_words2(c, pr->p1->gp.p, hpar_d, GW * 2 * S, nullptr, nullptr, 0));
        else
            HIPCHK(c, hipMemcpyAsync(pr->p1->gp.p, pr->hpar, sizeof(GaussParams) * 2 * S,
                                     hipMemcpyHostToDevice, c->stream));
        BQCHK(plan_enqueue(c, pr->p1));
        HIPCHK(c, hipMemsetAsync(pr->flag.p, 0, sizeof(int) * S, c->stream));
        hipLaunchKernelGGL(pair_collect_kernel, dim3(S), dim3(64), 0, c->stream, pr->p1->scal.d(),
                           pr->p1->scal.d() + 4 * S, pr->p1->info.i(), pr->p1->info.i() + S,
                           pr->flag.i(), pr->p1->y.d(), (long)pr->p1->L.npad, ns, 0,
                           pr->dres.d());
        HIPCHK(c, hipGetLastError());
    } else {
        if (hpar_d) {
            BQCHK(launch_copy_words2(c, pr->p1->gp.p, hpar_d, GW * S, pr->p2->gp.p, hpar_d + S,
                                     GW * S));
        } else {
            HIPCHK(c, hipMemcpyAsync(pr->p1->gp.p, pr->hpar, sizeof(GaussParams) * S,
                                     hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(pr->p2->gp.p, pr->hpar + S, sizeof(GaussParams) * S,
                                     hipMemcpyHostToDevice, c->stream));
        }
        BQCHK(plan_enqueue(c, pr->p1));
        HIPCHK(c, hipMemsetAsync(pr->flag.p, 0, sizeof(int) * S, c->stream));
        const long ms = std::max(nc, 1), ys = pr->p2->L.npad;
        hipLaunchKernelGGL(pair_targets_kernel, dim3((nsc + 255) / 256, S), dim3(256), 0,
                           c->stream, pr->l_s.d(), ns, nc, pr->p1->mean.d(), pr->p1->var.d(), ms,
                           max_log(), pr->p2->y.d(), ys, pr->flag.i());
        HIPCHK(c, hipGetLastError());
        BQCHK(plan_enqueue(c, pr->p2));
        hipLaunchKernelGGL(pair_collect_kernel, dim3(S), dim3(64), 0, c->stream,
                           pr->p1->scal.d(), pr->p2->scal.d(), pr->p1->info.i(),
                           pr->p2->info.i(), pr->flag.i(), pr->p2->y.d(), ys, ns, nc,
                           pr->dres.d());
        HIPCHK(c, hipGetLastError());
    }
    if (hres_d)
        BQCHK(launch_copy_words2(c, hres_d, pr->dres.p, (size_t)S * (5 + nc), nullptr, nullptr, 0));
    else
        HIPCHK(c, hipMemcpyAsync(pr->hres, pr->dres.p, sizeof(double) * (size_t)S * (5 + nc),
                                 hipMemcpyDeviceToHost, c->stream));
    return BQ_OK;
}

// llh[b] = log_lh(GP1) + log_lh(GP2) under parameter set b = (p_tl[3b..], p_l[3b..]) = (h, w, s)
// each; -inf where a factorisation fails or the overflow guard trips (status[b] = 1 / 2 / 3:
// GP1 not positive definite / GP mean too large / GP2 not positive definite).  l_c (S x nc,
// optional): the candidates' values under every set.
extern "C" int bq_pair_llh(bq_ctx *c, bq_pair *pr, const double *p_tl, const double *p_l,
                           double *llh, double *l_c, int32_t *status)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!pr || !p_tl || !p_l || !llh)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (pr->ma)
        return fail(c, BQ_ERR_BAD_ARG, "pair was created for the acquisition (ma > 0)");
    const int S = pr->S, nc = pr->nc;
    BQCHK(check_params(c, p_tl, S, "pair_llh (GP1)"));
    BQCHK(check_params(c, p_l, S, "pair_llh (GP2)"));
    HIPCHK(c, hipSetDevice(c->device));
    // both plans' kernel parameters from pinned staging
    for (int b = 0; b < S; ++b) {
        const double w1[1] = {p_tl[3 * b + 1]}, w2[1] = {p_l[3 * b + 1]};
        pr->hpar[b] = make_params(1, p_tl[3 * b], w1, p_tl[3 * b + 2]);
        pr->hpar[S + b] = make_params(1, p_l[3 * b], w2, p_l[3 * b + 2]);
    }
    // one graph launch per pass where graphs are in use (the pass of a small system is a dozen
    // stream operations of a few microseconds each)
    const unsigned long long key = launch_config_key(c);
    if (pr->lgraph_state == 1 && pr->lgraph_key != key) {
        (void)hipGraphExecDestroy(pr->lgexec);
        (void)hipGraphDestroy(pr->lgraph);
        pr->lgexec = nullptr;
        pr->lgraph = nullptr;
        pr->lgraph_state = 0;
    }
    if (!c->prof && c->use_graph && c->own_stream && pr->lgraph_state == 0) {
        pr->lgraph_key = key;
        pr->lgraph_state = -1;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
            const int st = pair_llh_enqueue(c, pr);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(c->stream, &g);
            if (st == BQ_OK && e == hipSuccess && g &&
                hipGraphInstantiate(&pr->lgexec, g, nullptr, nullptr, 0) == hipSuccess) {
                pr->lgraph = g;
                pr->lgraph_state = 1;
            } else {
                if (g)
                    (void)hipGraphDestroy(g);
                (void)hipGetLastError(); // clear; fall back to eager launches
            }
        } else {
            (void)hipGetLastError();
        }
    }
    if (!c->prof && c->use_graph && c->own_stream && pr->lgraph_state == 1)
        HIPCHK(c, hipGraphLaunch(pr->lgexec, c->stream));
    else
        BQCHK(pair_llh_enqueue(c, pr));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int b = 0; b < S; ++b) {
        const double *r = pr->hres + (size_t)b * (5 + nc);
        const int st = r[2] != 0.0 ? 1 : (r[4] != 0.0 ? 2 : (r[3] != 0.0 ? 3 : 0));
        if (status)
            status[b] = st;
        llh[b] = st ? -std::numeric_limits<double>::infinity() : r[0] + r[1];
        if (l_c)
            for (int i = 0; i < nc; ++i)
                l_c[(size_t)b * nc + i] = r[5 + i];
    }
    return BQ_OK;
}

// bq_pair_esm's stage 2 as S factorisations + border rows (pair.h has the picture): the sets'
// parameters, int K p, jitters and GP2's targets are on the device (pr->gpd, ik, ika, dj1, dj2, y2)
static int pair_esm_border(bq_ctx *c, bq_pair *pr, int p, double thresh, double *A_a,
                           double *A_sc_l, int32_t *status)
{
    const int S = pr->S, ns = pr->ns, nsc = pr->nsc, ma = pr->ma;
    const int nt0 = nsc - p;
    const int ntot = p + (int)roundup(nt0 + ma + 2, 64);
    const long lda = pick_ld(ntot), astride = lda * (long)ntot;
    EsmLayout Ls;
    Ls.ns = 0;
    Ls.nsc = nt0;
    Ls.npad = (int)roundup(nt0 + 1, 64);
    Ls.ntot = Ls.npad + 64;
    const long ldas = Ls.ntot, asstride = ldas * (long)Ls.ntot;
    if (pr->bchunk == 0) {
        size_t freeb = 0, totalb = 0;
        HIPCHK(c, hipMemGetInfo(&freeb, &totalb));
        const size_t per = sizeof(double) * ((size_t)astride + (size_t)ma * asstride) * 2;
        // (the small systems are a batch in blockIdx.z: at most 65535 of them per pass)
        const int64_t sch = std::min<int64_t>(
            std::min<int64_t>(S, std::max<int64_t>(1, 65535 / ma)),
            std::max<int64_t>(1, (int64_t)(std::min<size_t>(freeb / 2, (size_t)8 << 30) / per)));
        const int64_t ech = sch * ma;
        hipError_t e = pr->bA.alloc(sizeof(double) * (size_t)astride * sch);
        auto A = [&](DevBuf &b, size_t bytes) {
            if (e == hipSuccess)
                e = b.alloc(bytes);
        };
        A(pr->bdinv, sizeof(double) * BQ_DINV_STRIDE * (size_t)sch);
        A(pr->binfo, sizeof(int) * (size_t)sch);
        A(pr->bws, sizeof(double) * sweep_ws_doubles(c, ntot, (int)sch));
        A(pr->sA, sizeof(double) * (size_t)asstride * ech);
        A(pr->sdinv, sizeof(double) * BQ_DINV_STRIDE * (size_t)ech);
        A(pr->sinfo, sizeof(int) * (size_t)ech);
        A(pr->sws, sizeof(double) * sweep_ws_doubles(c, Ls.ntot, (int)ech));
        A(pr->sout, sizeof(double) * 2 * (size_t)ech);
        if (e != hipSuccess) {
            for (DevBuf *b : {&pr->bA, &pr->bdinv, &pr->binfo, &pr->bws, &pr->sA, &pr->sdinv,
                              &pr->sinfo, &pr->sws, &pr->sout})
                b->release();
            return fail(c, e == hipErrorOutOfMemory ? BQ_ERR_NOMEM : BQ_ERR_HIP,
                        "pair_esm: workspace allocation failed: %s", hipGetErrorString(e));
        }
        pr->bchunk = sch;
    }
    const int64_t sch = pr->bchunk;
    std::vector<double> hout((size_t)sch * ma * 2);
    std::vector<int> hsi((size_t)sch * ma), hbi((size_t)sch);
    for (int64_t s0 = 0; s0 < S; s0 += sch) {
        const int nb = (int)std::min<int64_t>(sch, S - s0), ne = nb * ma;
        HIPCHK(c, hipMemsetAsync(pr->binfo.p, 0, sizeof(int) * nb, c->stream));
        HIPCHK(c, hipMemsetAsync(pr->sinfo.p, 0, sizeof(int) * ne, c->stream));
        {
            Bracket br(c, BQ_K_GRAM, 8.0 * ntot * (ntot + 1.0) / 2.0 * nb);
            hipLaunchKernelGGL(assemble_esmb_kernel, dim3((ntot + 127) / 128, (ntot + 63) / 64, nb),
                               dim3(256), 0, c->stream, pr->x_sc.d(), pr->x_a.d(), ma, (long)s0,
                               pr->ik.d(), pr->ika.d(), pr->y2.d(), (long)nsc,
                               static_cast<const GaussParams *>(pr->gpd.p), pr->bA.d(), lda, astride,
                               nsc, ntot);
            HIPCHK(c, hipGetLastError());
        }
        // the first p columns of every set, the whole trailing block kept up to date
        BQCHK(enqueue_potrf_partial(c, pr->bA.d(), lda, astride, nb, ntot, p, pr->bdinv.d(),
                                    pr->binfo.i(), pr->bws.d(), pr->bws.bytes / sizeof(double)));
        {
            Bracket br(c, BQ_K_GRAM, 8.0 * Ls.ntot * Ls.ntot * ne);
            hipLaunchKernelGGL(esmb_gather_kernel, dim3((Ls.ntot + 63) / 64, Ls.ntot / 64, ne),
                               dim3(256), 0, c->stream, pr->bA.d(), lda, astride, p, nsc, ma, ns,
                               pr->x_sc.d(), pr->x_a.d(), pr->dj1.d(), pr->dj2.d(), thresh,
                               (long)s0, (long)s0 * ma, pr->sA.d(), ldas, asstride, Ls);
            HIPCHK(c, hipGetLastError());
        }
        BQCHK(enqueue_potrf_partial(c, pr->sA.d(), ldas, asstride, ne, Ls.ntot, Ls.npad,
                                    pr->sdinv.d(), pr->sinfo.i(), pr->sws.d(),
                                    pr->sws.bytes / sizeof(double)));
        hipLaunchKernelGGL(esm_multi_finalize_kernel, dim3((ne + 255) / 256), dim3(256), 0,
                           c->stream, pr->sA.d(), ldas, asstride, Ls, ne, pr->sout.d());
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(hout.data(), pr->sout.p, sizeof(double) * 2 * ne,
                                 hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hsi.data(), pr->sinfo.p, sizeof(int) * ne, hipMemcpyDeviceToHost,
                                 c->stream));
        HIPCHK(c, hipMemcpyAsync(hbi.data(), pr->binfo.p, sizeof(int) * nb, hipMemcpyDeviceToHost,
                                 c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int k = 0; k < ne; ++k) {
            const int64_t e = s0 * ma + k;
            A_a[e] = hout[(size_t)2 * k];
            A_sc_l[e] = hout[(size_t)2 * k + 1];
            // (a failure in the common leading block is every candidate's; the small system's
            // column counts on from the cut)
            const int lead = hbi[(size_t)(k / ma)];
            status[e] = lead ? lead : (hsi[(size_t)k] ? p + hsi[(size_t)k] : 0);
        }
    }
    return BQ_OK;
}

// The acquisition under S parameter sets: for set b and candidate a (element b * ma + a)
//   A_a, A_sc_l   the two bilinear forms of bq_c.pyx:425-490 from the bordered (nsc + 1)^2
//                 system with the reference's jitter (status = 1: singular system, the
//                 caller's fallback of bq.py:481-490)
//   tm_a, tC_a    GP1's posterior mean / variance at x_a (bq.py:493-496)
// and per set: l_c (S x nc) and sstatus (1: GP1 not positive definite, 2: GP mean too large;
// the set's elements are then not computed).  Only the noise-free Gram of GP2's kernel enters
// (gp.Kxoxo, bq.py:465): p_l's s is not used.
extern "C" int bq_pair_esm(bq_ctx *c, bq_pair *pr, const double *p_tl, const double *p_l,
                           double thresh, const double *mu, const double *cov, double *A_a,
                           double *A_sc_l, int32_t *status, double *tm_a, double *tC_a,
                           double *l_c, int32_t *sstatus)
{
    if (!c)
        return BQ_ERR_BAD_ARG;
    if (!pr || !p_tl || !p_l || !mu || !cov || !A_a || !A_sc_l || !status || !tm_a || !tC_a ||
        !sstatus)
        return fail(c, BQ_ERR_BAD_ARG, "illegal value");
    if (pr->ma == 0)
        return fail(c, BQ_ERR_BAD_ARG, "pair was created without acquisition points");
    const int S = pr->S, ns = pr->ns, nc = pr->nc, nsc = pr->nsc, ma = pr->ma;
    BQCHK(check_params(c, p_tl, S, "pair_esm (GP1)"));
    BQCHK(check_params(c, p_l, S, "pair_esm (GP2)"));
    if (!(cov[0] > 0.0))
        return fail(c, BQ_ERR_NOT_PD, "matrix is not positive definite");
    HIPCHK(c, hipSetDevice(c->device));
    BQCHK(run_stage1(c, pr, p_tl, pr->y2.d(), nsc));
    // GP1's posterior at x_a and the set flags come back while stage 2 is being set up
    const int M1 = nc + ma;
    std::vector<double> hm((size_t)S * M1), hv((size_t)S * M1);
    std::vector<int> i1((size_t)S), fl((size_t)S);
    HIPCHK(c, hipMemcpyAsync(hm.data(), pr->p1->mean.p, sizeof(double) * hm.size(),
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hv.data(), pr->p1->var.p, sizeof(double) * hv.size(),
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(i1.data(), pr->p1->info.p, sizeof(int) * S, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipMemcpyAsync(fl.data(), pr->flag.p, sizeof(int) * S, hipMemcpyDeviceToHost,
                             c->stream));
    if (l_c && nc)
        HIPCHK(c, hipMemcpy2DAsync(l_c, sizeof(double) * nc, pr->y2.d() + ns, sizeof(double) * nsc,
                                   sizeof(double) * nc, S, hipMemcpyDeviceToHost, c->stream));
    // ---- stage 2: S x ma bordered systems -------------------------------------------------
    EsmLayout L;
    L.ns = ns;
    L.nsc = nsc;
    L.npad = (int)roundup(nsc + 1, 64);
    L.ntot = L.npad + 64;
    const long lda = pick_ld(L.ntot);
    const int64_t E = (int64_t)S * ma;
    const size_t per = sizeof(double) * ((size_t)lda * L.ntot + panel_ws_doubles(L.ntot, 1) +
                                         BQ_DINV_STRIDE + 2) + sizeof(int);
    if (!pr->esm_small) {
        HIPCHK(c, pr->gpd.alloc(sizeof(GaussParams) * S));
        HIPCHK(c, pr->pard.alloc(sizeof(double) * 3 * S));
        HIPCHK(c, pr->ik.alloc(sizeof(double) * (size_t)S * nsc));
        HIPCHK(c, pr->ika.alloc(sizeof(double) * (size_t)E));
        HIPCHK(c, pr->dj1.alloc(sizeof(double) * (size_t)E));
        HIPCHK(c, pr->dj2.alloc(sizeof(double) * (size_t)E));
        pr->esm_small = true;
    }
    // the cut of the border route: the jitter sits on candidate points, rows >= ns >= pcut
    const int pcut = (ns / 64) * 64;
    const bool border = c->pair_border && pcut >= 64;
    if (!border && pr->chunk == 0) {
        // the S Ma full systems: a workspace of at most 8 GB (it stays with the pair between
        // calls -- choose_next calls once per step with the same shapes -- and a BQ object keeps
        // up to four pairs: an unbounded half of the free HBM starved later fits and plans)
        size_t freeb = 0, totalb = 0;
        HIPCHK(c, hipMemGetInfo(&freeb, &totalb));
        const size_t budget = std::min<size_t>(freeb / 2, (size_t)8 << 30);
        int64_t ch = std::max<int64_t>(1, (int64_t)(budget / per));
        ch = std::min<int64_t>(std::min<int64_t>(ch, E), 32768);
        hipError_t e = pr->Ad.alloc(sizeof(double) * (size_t)lda * L.ntot * (size_t)ch);
        if (e == hipSuccess)
            e = pr->dinv.alloc(sizeof(double) * BQ_DINV_STRIDE * (size_t)ch);
        if (e == hipSuccess)
            e = pr->panel.alloc(sizeof(double) * sweep_ws_doubles(c, L.ntot, (int)ch));
        if (e == hipSuccess)
            e = pr->info.alloc(sizeof(int) * (size_t)ch);
        if (e == hipSuccess)
            e = pr->outd.alloc(sizeof(double) * 2 * (size_t)ch);
        if (e != hipSuccess) {
            // nothing half-allocated stays behind
            pr->Ad.release(), pr->dinv.release(), pr->panel.release(), pr->info.release(),
                pr->outd.release();
            return fail(c, e == hipErrorOutOfMemory ? BQ_ERR_NOMEM : BQ_ERR_HIP,
                        "pair_esm: workspace allocation failed: %s", hipGetErrorString(e));
        }
        pr->chunk = ch;
    }
    const int64_t chunk = std::max<int64_t>(pr->chunk, 1);
    DevBuf &gpd = pr->gpd, &pard = pr->pard, &ik = pr->ik, &ika = pr->ika, &dj1 = pr->dj1,
           &dj2 = pr->dj2, &Ad = pr->Ad, &dinv = pr->dinv, &info = pr->info, &outd = pr->outd,
           &panel = pr->panel;
    // per set: kernel parameters of GP2 (no noise term), the closed form of int K p;
    // per element: the jitter exactly as two successive improve_covariance_conditioning calls
    // produce it (bq.py:470-476)
    const double eps = std::numeric_limits<double>::epsilon();
    std::vector<GaussParams> gps((size_t)S);
    std::vector<double> par((size_t)S * 3), j1((size_t)E), j2((size_t)E);
    std::vector<char> close((size_t)ma, 0);
    for (int a = 0; a < ma; ++a)
        for (int j = 0; j < nc && !close[(size_t)a]; ++j)
            close[(size_t)a] = std::fabs(pr->hx_c[(size_t)j] - pr->hx_a[(size_t)a]) < thresh;
    for (int b = 0; b < S; ++b) {
        const double h = p_l[3 * b], wv[1] = {p_l[3 * b + 1]};
        gps[(size_t)b] = make_params(1, h, wv, 0.0);
        const double Cc = 1.0 * cov[0] + wv[0] * wv[0], Lc = std::sqrt(Cc);
        par[(size_t)3 * b] = h * h;
        par[(size_t)3 * b + 1] = 1.0 / Lc;
        par[(size_t)3 * b + 2] = -0.5 * (1 * LOG_2PI + 2.0 * std::log(Lc));
        for (int a = 0; a < ma; ++a) {
            const double first = close[(size_t)a] ? std::max(eps, gps[(size_t)b].c) * 1e-4 : 0.0;
            j1[(size_t)b * ma + a] = first;
            j2[(size_t)b * ma + a] = std::max(eps, gps[(size_t)b].c + first) * 1e-4;
        }
    }
    HIPCHK(c, hipMemcpyAsync(gpd.p, gps.data(), sizeof(GaussParams) * S, hipMemcpyHostToDevice,
                             c->stream));
    HIPCHK(c, hipMemcpyAsync(pard.p, par.data(), sizeof(double) * 3 * S, hipMemcpyHostToDevice,
                             c->stream));
    HIPCHK(c, hipMemcpyAsync(dj1.p, j1.data(), sizeof(double) * E, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dj2.p, j2.data(), sizeof(double) * E, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(pair_int_K_kernel, dim3((nsc + 255) / 256, S), dim3(256), 0, c->stream,
                       pr->x_sc.d(), nsc, pard.d(), mu[0], ik.d());
    hipLaunchKernelGGL(pair_int_K_kernel, dim3((ma + 255) / 256, S), dim3(256), 0, c->stream,
                       pr->x_a.d(), ma, pard.d(), mu[0], ika.d());
    HIPCHK(c, hipGetLastError());
    if (border)
        BQCHK(pair_esm_border(c, pr, pcut, thresh, A_a, A_sc_l, status));
    std::vector<double> hout((size_t)chunk * 2);
    std::vector<int> hinfo((size_t)chunk);
    for (int64_t e0 = 0; e0 < E && !border; e0 += chunk) {
        const int nb = (int)std::min(chunk, E - e0);
        HIPCHK(c, hipMemsetAsync(info.p, 0, sizeof(int) * nb, c->stream));
        {
            Bracket br(c, BQ_K_GRAM, 8.0 * L.ntot * (L.ntot + 1.0) / 2.0 * nb);
            dim3 grid((L.ntot + 127) / 128, (L.ntot + 63) / 64, nb);
            hipLaunchKernelGGL(assemble_esm_multi_kernel, grid, dim3(256), 0, c->stream,
                               pr->x_sc.d(), pr->x_a.d(), ma, (long)e0, ik.d(), ika.d(),
                               pr->y2.d(), (long)nsc, dj1.d(), dj2.d(), thresh,
                               static_cast<const GaussParams *>(gpd.p), Ad.d(), lda,
                               lda * (long)L.ntot, L);
            HIPCHK(c, hipGetLastError());
        }
        BQCHK(enqueue_potrf_partial(c, Ad.d(), lda, lda * (long)L.ntot, nb, L.ntot, L.npad,
                                    dinv.d(), info.i(), panel.d(), panel.bytes / sizeof(double)));
        hipLaunchKernelGGL(esm_multi_finalize_kernel, dim3((nb + 255) / 256), dim3(256), 0,
                           c->stream, Ad.d(), lda, lda * (long)L.ntot, L, nb, outd.d());
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(hout.data(), outd.p, sizeof(double) * 2 * nb,
                                 hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hinfo.data(), info.p, sizeof(int) * nb, hipMemcpyDeviceToHost,
                                 c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int k = 0; k < nb; ++k) {
            A_a[e0 + k] = hout[(size_t)2 * k];
            A_sc_l[e0 + k] = hout[(size_t)2 * k + 1];
            status[e0 + k] = hinfo[(size_t)k];
        }
    }
    for (int b = 0; b < S; ++b) {
        sstatus[b] = i1[(size_t)b] ? 1 : (fl[(size_t)b] ? 2 : 0);
        for (int a = 0; a < ma; ++a) {
            tm_a[(size_t)b * ma + a] = hm[(size_t)b * M1 + nc + a];
            tC_a[(size_t)b * ma + a] = hv[(size_t)b * M1 + nc + a];
        }
    }
    return BQ_OK;
}
