// probe.hip -- hardware probes behind the C ABI (MFMA / FMA / HBM rates, launch latency,
// operand layouts, the diagonal factor's timeline) and their kernels (probe.h).
#include "host.h"
#include "probe.h"

using namespace bqh;

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

// `reps` launches of probe_read8_kernel over a `bytes`-sized buffer: a known byte count in the
// single-vector sweeps' access pattern (8 B per lane, 512 contiguous bytes per wave), for the
// calibration of rocprofv3's FETCH_SIZE on that pattern (tools/r06_profiles.sh, pass `calib`)
extern "C" int bq_probe_hbm_read8(bq_ctx *c, size_t bytes, int64_t reps, double *read_gbs)
{
    if (!c || bytes < 4096 || reps < 1)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf a, o;
    HIPCHK(c, a.alloc(bytes));
    HIPCHK(c, o.alloc(64));
    HIPCHK(c, hipMemsetAsync(a.p, 0, bytes, c->stream));
    const size_t n = bytes / 8;
    float ms = 0;
    hipLaunchKernelGGL(probe_read8_kernel, dim3(c->cus * 2), dim3(1024), 0, c->stream, a.d(), n,
                       o.d());
    BQCHK(bq_timer_start(c));
    for (int64_t i = 0; i < reps; ++i)
        hipLaunchKernelGGL(probe_read8_kernel, dim3(c->cus * 2), dim3(1024), 0, c->stream, a.d(),
                           n, o.d());
    BQCHK(bq_timer_stop_ms(c, &ms));
    HIPCHK(c, hipGetLastError());
    if (read_gbs)
        *read_gbs = (double)reps * bytes / (ms * 1e-3) / 1e9;
    return BQ_OK;
}

// C (m x n) -= P (m x k) Q (n x k)^T on scratch operands through launch_gemm (the engine's own
// kernel selection): average ms over `reps` back-to-back launches.  qt: Q given k-contiguous.
extern "C" int bq_probe_gemm(bq_ctx *c, int64_t m, int64_t n, int64_t k, int lower, int64_t batch,
                             int qt, int64_t reps, double *ms_out)
{
    if (!c || !ms_out || m < 16 || n < 16 || k < 8 || batch < 1 || reps < 1)
        return BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf C, P, Q;
    const long ldc = m, ldp = m, ldq = qt ? k : n;
    HIPCHK(c, C.alloc(sizeof(double) * (size_t)ldc * n * batch));
    HIPCHK(c, P.alloc(sizeof(double) * (size_t)ldp * k * batch));
    HIPCHK(c, Q.alloc(sizeof(double) * (size_t)n * k * batch));
    HIPCHK(c, hipMemsetAsync(C.p, 0, C.bytes, c->stream));
    hipLaunchKernelGGL(probe_fill_kernel, dim3(2048), dim3(256), 0, c->stream, P.d(),
                       P.bytes / sizeof(double), 1u);
    hipLaunchKernelGGL(probe_fill_kernel, dim3(2048), dim3(256), 0, c->stream, Q.d(),
                       Q.bytes / sizeof(double), 77u);
    HIPCHK(c, hipGetLastError());
    auto run = [&]() {
        return launch_gemm(c, BQ_K_GEMM, C.d(), ldc, ldc * n, P.d(), ldp, ldp * k, Q.d(),
                           qt ? ldq : 1, qt ? 1 : ldq, (long)n * k, (int)m, (int)n, (int)k, lower,
                           (int)batch);
    };
    BQCHK(run());
    float ms = 0;
    BQCHK(bq_timer_start(c));
    for (int64_t i = 0; i < reps; ++i)
        BQCHK(run());
    BQCHK(bq_timer_stop_ms(c, &ms));
    *ms_out = ms / (double)reps;
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
    if (kind == 4 || kind == 5) {
        // the GEMM inner step SUSTAINED: 300 launches of ~1 ms back to back, the last 200 timed;
        // kind 4: operands near 1.0 (few mantissa bits set), kind 5: random mantissas
        const int it = 1024, blocks = c->cus * blocks_per_cu;
        float ms = 0;
        for (int rep = 0; rep < 300; ++rep) {
            if (rep == 100)
                BQCHK(bq_timer_start(c));
            if (kind == 4)
                hipLaunchKernelGGL((probe_mfma_step_kernel<0, 0>), dim3(blocks), dim3(256), 0,
                                   c->stream, o.d(), it);
            else
                hipLaunchKernelGGL((probe_mfma_step_kernel<0, 1>), dim3(blocks), dim3(256), 0,
                                   c->stream, o.d(), it);
        }
        BQCHK(bq_timer_stop_ms(c, &ms));
        *tflops = 200.0 * (double)blocks * 4 * (double)it * 64 * 512.0 / (ms * 1e-3) / 1e12;
        return BQ_OK;
    }
    if (kind >= 2) { // the GEMM inner step, kind 2: no rotations, 3: with rotations
        const int it = 512, blocks = c->cus * blocks_per_cu;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            BQCHK(bq_timer_start(c));
            if (kind == 2)
                hipLaunchKernelGGL((probe_mfma_step_kernel<0, 0>), dim3(blocks), dim3(256), 0,
                                   c->stream, o.d(), it);
            else
                hipLaunchKernelGGL((probe_mfma_step_kernel<1, 0>), dim3(blocks), dim3(256), 0,
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
    // from_lds bit 2: per-wave barrier stamps as well (four-wave form; stamps[5] is the switch)
    {
        const long long flag = (from_lds & 4) ? 1 : 0;
        HIPCHK(c, hipMemsetAsync(st.p, 0, sizeof(long long) * 136, c->stream));
        HIPCHK(c, hipMemcpyAsync(static_cast<long long *>(st.p) + 5, &flag, sizeof flag,
                                 hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    auto launch = [&]() {
        // from_lds bit 1: the eight-wave form
        if (from_lds & 2)
            hipLaunchKernelGGL(potf2_probe_kernel<8>, dim3(1), dim3(512), 0, c->stream, ain.d(),
                               a.d(), 64L, dv.d(), inf.i(), static_cast<long long *>(st.p),
                               from_lds & 1);
        else
            hipLaunchKernelGGL(potf2_probe_kernel<4>, dim3(1), dim3(256), 0, c->stream, ain.d(),
                               a.d(), 64L, dv.d(), inf.i(), static_cast<long long *>(st.p),
                               from_lds & 1);
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

// ns per hand-off (one direction) of probe_hop_kernel's eight ping-pong pairs, the pairs' XCC ids
// and the payload words that arrived wrong; BQ_ERR_HIP if a partner never answered
extern "C" int bq_probe_xcd_hop(bq_ctx *c, int mode, int64_t iters, int64_t kib, double *ns_per_hop,
                                int32_t *xcc16, int64_t *bad_words)
{
    if (!c || !ns_per_hop || !xcc16 || !bad_words || mode < 0 || mode > 2 || iters < 1 ||
        iters > 100000 || kib < 1 || kib > 64)
        return c ? fail(c, BQ_ERR_BAD_ARG, "xcd_hop: mode 0..2, 1..100000 iterations, 1..64 KiB")
                 : BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    DevBuf flags, pay, out, xcc, bad, err;
    HIPCHK(c, flags.alloc(sizeof(unsigned) * 64 * 16));
    HIPCHK(c, pay.alloc(sizeof(double) * 128 * (size_t)kib * 16));
    HIPCHK(c, out.alloc(sizeof(long long) * 16));
    HIPCHK(c, xcc.alloc(sizeof(int) * 16));
    HIPCHK(c, bad.alloc(sizeof(int) * 16));
    HIPCHK(c, err.alloc(sizeof(int)));
    HIPCHK(c, hipMemsetAsync(flags.p, 0, flags.bytes, c->stream));
    HIPCHK(c, hipMemsetAsync(pay.p, 0, pay.bytes, c->stream));
    HIPCHK(c, hipMemsetAsync(err.p, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(probe_hop_kernel, dim3(16), dim3(64), 0, c->stream,
                       static_cast<unsigned *>(flags.p), pay.d(), (int)iters, mode, (int)kib,
                       static_cast<long long *>(out.p), xcc.i(), bad.i(), err.i());
    HIPCHK(c, hipGetLastError());
    long long ho[16];
    int hb[16], he = 0;
    HIPCHK(c, hipMemcpyAsync(ho, out.p, sizeof ho, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(xcc16, xcc.p, sizeof(int) * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hb, bad.p, sizeof hb, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&he, err.p, sizeof he, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (he)
        return fail(c, BQ_ERR_HIP, "xcd_hop: a partner never answered (bounded spin ran out)");
    long long worst = 0;
    *bad_words = 0;
    for (int b = 0; b < 16; ++b) {
        worst = std::max(worst, ho[b]);
        *bad_words += hb[b];
    }
    *ns_per_hop = (double)worst * 10.0 / (2.0 * (double)iters); // 100 MHz ticks
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

// The batched panel solve alone: X (m x kb per problem) <- X L^-T against `batch` lower-triangular
// kb x kb factors L (host, column-major, dense lower triangles), through the launches the batched
// factorisation issues for an outer block (mode 0: as the context is configured; 1: the recursive
// products + solves; 2: the one-launch sweep).  The factors' block-inverse records are built on
// the device (diag_winv_kernel).
extern "C" int bq_probe_panel_solve(bq_ctx *c, int64_t m, int64_t kb, int64_t batch, const double *L,
                                    double *X, int mode, int64_t reps, double *ms_per_call)
{
    if (!c || !L || !X || m < 64 || (m & 63) || kb < 64 || (kb & 63) || batch < 1)
        return c ? fail(c, BQ_ERR_BAD_ARG, "panel_solve: m, kb multiples of 64") : BQ_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const long lda = (long)(kb + m), astride = lda * (long)kb;
    const long rstride = (long)(kb / 64) * BQ_DINV_HALF;
    DevBuf A, rec;
    HIPCHK(c, A.alloc(sizeof(double) * (size_t)astride * batch));
    HIPCHK(c, rec.alloc(sizeof(double) * (size_t)rstride * batch));
    for (int64_t b = 0; b < batch; ++b) {
        HIPCHK(c, hipMemcpy2DAsync(A.d() + b * astride, sizeof(double) * lda, L + b * kb * kb,
                                   sizeof(double) * kb, sizeof(double) * kb, kb,
                                   hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(A.d() + b * astride + kb, sizeof(double) * lda, X + b * m * kb,
                                   sizeof(double) * m, sizeof(double) * m, kb,
                                   hipMemcpyHostToDevice, c->stream));
        BQCHK(launch_diag_winv(c, A.d() + b * astride, lda, (int)kb, rec.d() + b * rstride));
    }
    const int keep = c->df_sweep;
    if (mode == 1)
        c->df_sweep = 0;
    if (mode == 2)
        c->df_sweep = 1;
    const int st = enqueue_panel_solve(c, A.d(), lda, astride, (int)batch, (int)kb, (int)m, 0,
                                       (int)kb, rec.d(), rstride);
    if (st == BQ_OK && reps > 0 && ms_per_call) {
        // timing: the same call again and again on its own (now solved, still finite) output
        float ms = 0;
        BQCHK(bq_timer_start(c));
        for (int64_t r = 0; r < reps; ++r)
            (void)enqueue_panel_solve(c, A.d(), lda, astride, (int)batch, (int)kb, (int)m, 0,
                                      (int)kb, rec.d(), rstride);
        BQCHK(bq_timer_stop_ms(c, &ms));
        *ms_per_call = ms / (double)reps;
        c->df_sweep = keep;
        return BQ_OK; // (X is not downloaded: it has been solved reps + 1 times)
    }
    c->df_sweep = keep;
    BQCHK(st);
    for (int64_t b = 0; b < batch; ++b)
        HIPCHK(c, hipMemcpy2DAsync(X + b * m * kb, sizeof(double) * m, A.d() + b * astride + kb,
                                   sizeof(double) * lda, sizeof(double) * m, kb,
                                   hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return BQ_OK;
}
