"""One fixed, short workload per profiler pass (every `rooflines` fraction
must be reproducible from the tracked profiles alone, so passes are never blended):

    python3 tools/roofline_run.py <pass>

  c2            20 passes of the C2 problem (16 slab_step_kernel launches each)
  gram          gram_tri_kernel<2> at N=4096 d=2 (10 launches), gram_tri_kernel<1> at N=16384 (3)
  potrf256      ONE sequential (no look-ahead) potrf at N=16384 with the config's tile 256:
                the trailing updates on gemm_lds_kernel in isolation
  potrf256_dense  the same on DENSE operands (workloads.c4_dense: w = 200 dx, s = 1) -- C4's own
                Gram is banded and feeds the update > 97 % zeros
  potrf_engine  ONE potrf at N=16384 as shipped (engine block 512, two-stream look-ahead)
  trsv          resident N=16384 fit, 4 single-vector solves (one launch per sweep: trsv_fwd_flow_kernel, trsv_bwd_flow_kernel)
  solve256      resident N=4096 and N=16384 fits, 2 solves with 256 right-hand sides each
  predict       resident C2 fit, 5 predictions at M=256 and 5 at M=1000
  c5            one C5 shard (64 x N=2048, M=256), 4 plan runs
  c2x256        256 copies of the C2 problem as one plan, 4 runs
  c3            one C3 chunk (100 grid points at N=4096 d=2), 2 runs
  calib         4 read-only passes over 1 GiB with 8-byte-per-lane loads (FETCH_SIZE calibration)
  fitpost_n<N>  ONE problem at N (1024 / 2048 / 4096 / 16384), M = 256 through the bordered plan,
                4 passes: the kernels behind bench.py's fit_posterior_ms_at_n

Used under `rocprofv3 --kernel-trace --stats` and, in separate runs, `rocprofv3 --pmc ...`
(tools/r06_profiles.sh); tools/pmc_summary.py turns the outputs into profiles/r06_*."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine, _lib as L_  # noqa: E402
from bayesian_quadrature_amd import workloads as wl  # noqa: E402


def potrf_16384(e, nb, lookahead, dense=False):
    lib, ctx = e._lib, e._ctx
    n = 16384
    c4 = wl.c4_dense(n) if dense else wl.c4(n)
    w4 = np.ascontiguousarray(c4["w"])
    xd, Kd, info = e.alloc(8 * n), e.alloc(8 * n * n), e.alloc(64)
    e.upload(xd, np.ascontiguousarray(c4["x"]))
    e.set_lookahead(lookahead)
    e.set_block(nb)
    e._check(lib.bq_gram_gauss_dev(ctx, xd, 1, n, c4["h"], L_.dptr(w4), c4["s"], Kd, n))
    e._check(lib.bq_potrf_dev(ctx, Kd, n, n, info))
    e.sync()
    h = np.zeros(1, dtype=np.int32)
    e.download(h, info)
    print("potrf info", h[0])
    e.free(xd), e.free(Kd), e.free(info)


def main():
    what = sys.argv[1]
    e = Engine(0)
    lib, ctx = e._lib, e._ctx
    if what == "c2":
        c2 = wl.c2()
        plan = e.plan(1, 1, 1024, 256)
        plan.set_inputs(c2["x"][None], c2["y"][None], c2["xo"][None], c2["h"], c2["w"], c2["s"])
        for _ in range(20):
            plan.run()
        e.sync()
        plan.close()
    elif what == "gram":
        c3 = wl.c3()
        pts = np.asfortranarray(c3["x"])
        w3 = np.ascontiguousarray(c3["w"][200])
        xd, Kd = e.alloc(8 * 2 * 4096), e.alloc(8 * 4096 * 4096)
        e.upload(xd, pts)
        for _ in range(10):
            e._check(lib.bq_gram_gauss_dev(ctx, xd, 2, 4096, float(c3["h"][200]), L_.dptr(w3),
                                           c3["s"], Kd, 4096))
        e.sync()
        e.free(xd), e.free(Kd)
        n = 16384
        c4 = wl.c4(n)
        w4 = np.ascontiguousarray(c4["w"])
        xd, Kd = e.alloc(8 * n), e.alloc(8 * n * n)
        e.upload(xd, np.ascontiguousarray(c4["x"]))
        for _ in range(3):
            e._check(lib.bq_gram_gauss_dev(ctx, xd, 1, n, c4["h"], L_.dptr(w4), c4["s"], Kd, n))
        e.sync()
        e.free(xd), e.free(Kd)
    elif what == "potrf256":
        potrf_16384(e, 256, False)
    elif what == "potrf256_dense":
        potrf_16384(e, 256, False, dense=True)
    elif what == "potrf_engine":
        potrf_16384(e, 0, True)
    elif what in ("trsv", "solve256"):
        for n in ((16384,) if what == "trsv" else (4096, 16384)):
            c4 = wl.c4(n)
            fit = e.gp_fit(c4["x"], wl.norm_logpdf(c4["x"]), c4["h"], c4["w"], c4["s"])
            rs = np.random.RandomState(5)
            if what == "trsv":
                b = rs.randn(n)
                for _ in range(4):
                    fit.solve(b)
            else:
                B = np.asfortranarray(rs.randn(n, 256))
                for _ in range(2):
                    fit.solve(B)
            e.sync()
            fit.close()
    elif what == "predict":
        c2 = wl.c2()
        fit = e.gp_fit(c2["x"], c2["y"], c2["h"], c2["w"], c2["s"])
        for M in (256, 1000):
            xo = np.linspace(-5.0, 5.0, M) + 1e-3
            for _ in range(5):
                fit.predict(xo)
        fit.close()
    elif what == "c5":
        c5 = wl.c5(range(64))
        plan = e.plan(64, 1, 2048, 256)
        plan.set_inputs(c5["x"], c5["y"], c5["xo"], c5["h"], c5["w"], c5["s"])
        for rep in range(4):
            e.sync()
            e.timer_start()
            plan.run()
            print("c5 rep", rep, "%.3f ms" % e.timer_stop_ms(), flush=True)
        plan.close()
    elif what == "c2x256":
        c2 = wl.c2()
        B = 256
        plan = e.plan(B, 1, 1024, 256)
        plan.set_inputs(np.repeat(c2["x"][None], B, 0), np.repeat(c2["y"][None], B, 0),
                        np.repeat(c2["xo"][None], B, 0), c2["h"], c2["w"], c2["s"])
        for rep in range(4):
            e.sync()
            e.timer_start()
            plan.run()
            print("c2x256 rep", rep, "%.3f ms" % e.timer_stop_ms(), flush=True)
        plan.close()
    elif what == "c3":
        import time
        c3 = wl.c3()
        for rep in range(2):
            t0 = time.perf_counter()
            e.logml_grid(c3["x"], c3["y"], c3["h"][:100], c3["w"][:100], c3["s"], chunk=100)
            print("c3 chunk rep", rep, "%.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    elif what.startswith("fitpost_n"):
        n = int(what[len("fitpost_n"):])
        c = wl.c2(n, 256)
        plan = e.plan(1, 1, n, 256)
        plan.set_inputs(c["x"][None], c["y"][None], c["xo"][None], c["h"], c["w"], c["s"])
        for rep in range(4):
            e.sync()
            e.timer_start()
            plan.run()
            print(what, "rep", rep, "%.3f ms" % e.timer_stop_ms(), flush=True)
        plan.close()
    elif what == "calib":
        print("read8 GB/s", e.probe_hbm_read8(1 << 30, 4))
    else:
        raise SystemExit("unknown pass " + what)
    e.close()


if __name__ == "__main__":
    main()
