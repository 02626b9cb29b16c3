#!/usr/bin/env python3
"""bench.py -- BQ fit+posterior throughput on MI355X, with roofline and CPU baseline.

Contract (one JSON line on stdout from rank 0):
    python bench.py --gpus N --steps K --warmup W
A "step" is one pass of the hot path over one batch of synthetic problems that
are already resident in HBM: assemble the bordered Gaussian-kernel system,
blocked fp64 Cholesky (MFMA trailing update), posterior mean/variance at the
candidate points and the log marginal likelihood.  With one GPU the default
workload is BASELINE.json configs[1] (C2: d=1, N=1024, M=256); with N > 1 it is the
batched configs[4] (C5: 512 independent N=2048 problems over 8 GPUs = a block of 64
problems per rank), per-GPU work fixed as N grows (weak scaling, no collective on
the data path -- ranks only meet at the timing barrier).

Ranks: under ``python -m torch.distributed.run --nproc-per-node N`` every rank reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  Started bare with
``--gpus N`` (N > 1) the parent process starts N fresh child processes with that
environment itself -- before it has touched HIP in any way -- waits for them and
exits non-zero if any of them fails; it never creates an engine and never re-execs.

Extra objects on the same line:
    roofline      the dominant kernel class of the timed workload
    rooflines     the two north-star lines: N=16384 Cholesky trailing update
                  (fp64 MFMA) and N=4096 / N=16384 Gram build (HBM)
    cpu_baseline  the CPU oracle (a port, OpenMP) on a bounded sample of the
                  same workload, rank 0, N=1 only
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# peaks: HBM from /opt/skills/guides/MI355X_MICROARCH.md (8.0 TB/s spec); the guide
# has no fp64 row, so the fp64 matrix peak is AMD's public MI355X figure (78.6
# TFLOP/s, equal to the fp64 vector peak) and the on-box probe is reported beside it.
PEAK_HBM_GBS = 8000.0
PEAK_FP64_TFLOPS = 78.6


# the 128x128-tile LDS-staged trailing-update kernel as rocprofv3 names it
# (v_mfma_f64_4x4x4_4b_f64, four 64x64 wave tiles)
TRAILING_KERNEL = "gemm_lds_kernel"


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary
    (profiles/*_pmc_traffic.json, made by tools/pmc_summary.py from separate
    `rocprofv3 --pmc` passes); None when there is none.  WRITE_SIZE + FETCH_SIZE with the
    guide's gfx950 correction where it applies (16-B-per-lane reads tally at half): the
    corrected figure is an ESTIMATE and the raw counters stay beside it in the summary."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        ks = json.load(f)["kernels"]
    # (the headline kernel: its launches inside the C2 passes alone, if the summary has them)
    k = ks.get(kernel + " [C2 passes]") or ks.get(kernel)
    if not k:
        # template arguments change between rounds (slab_step_kernel<false> became <false, 8>):
        # fall back to the same kernel name with any arguments, the C2 passes first
        base = kernel.split("<")[0]
        cand = sorted((n for n in ks if n.split("<")[0].split(" ")[0] == base),
                      key=lambda n: ("[C2 passes]" not in n, n))
        k = ks.get(cand[0]) if cand else None
    if not k:
        return None, None
    fetch = k.get("FETCH_SIZE_estimate_bytes_avg",
                  k.get("FETCH_SIZE_corrected_bytes_avg", k.get("FETCH_SIZE_bytes_avg", 0.0)))
    return (fetch + k.get("WRITE_SIZE_bytes_avg", 0.0),
            "profiles/%s (separate rocprofv3 --pmc passes, not this run)"
            % os.path.basename(files[-1]))


def _trsv_traffic():
    """HBM bytes of one N=16384 single-vector solve from the committed PMC summary: the RAW
    FETCH_SIZE + WRITE_SIZE totals of the trsv step kernels over the pass's 4 solves, per solve
    (`bytes_per_solve_raw`), and beside it the estimate with the factor MEASURED on a known-bytes
    read of the same access pattern (8 B per lane: the counter is uncalibrated for it)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        doc = json.load(f)
    ks = doc["kernels"]
    raw, est, solves = 0.0, 0.0, None
    # (round 4: one launch per sweep; the summaries of rounds 2-3 hold the per-block kernels, 32
    # launches per sweep at N = 16384)
    names = ("trsv_fwd_flow_kernel<8>", "trsv_bwd_flow_kernel")
    per_solve = 1.0
    if not all(n in ks for n in names):
        names, per_solve = ("trsv_fwd_step_kernel<8>", "trsv_bwd_step_kernel"), 32.0
    for name in names:
        k = ks.get(name)
        if not k:
            return None
        raw += k.get("FETCH_SIZE_bytes_total", 0.0) + k.get("WRITE_SIZE_bytes_total", 0.0)
        est += (k.get("FETCH_SIZE_estimate_bytes_total",
                      k.get("FETCH_SIZE_corrected_bytes_total", k.get("FETCH_SIZE_bytes_total", 0.0)))
                + k.get("WRITE_SIZE_bytes_total", 0.0))
        solves = k["launches"] / per_solve
    cal = doc.get("read8_calibration")
    return {"bytes_per_solve_raw": raw / solves, "bytes_per_solve_estimate": est / solves,
            "calibration": cal,
            "from": "profiles/%s (separate rocprofv3 --pmc passes, not this run); the estimate "
                    "scales the raw FETCH_SIZE by the factor measured on probe_read8_kernel, a "
                    "known 1 GiB read with the same 8-B-per-lane pattern"
                    % os.path.basename(files[-1])}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=None, choices=["c2", "c5"],
                    help="default: c2 on one GPU, c5 (the batched config) on several")
    ap.add_argument("--batch", type=int, default=0, help="problems per GPU per step (0 = default)")
    ap.add_argument("--curve-batch", type=int, default=0,
                    help="problems per GPU of the scaling-curve workload (C5 shard) measured as "
                         "scale_point at N = 1 (0 = --batch when the workload is c5, else 64)")
    ap.add_argument("--no-extras", action="store_true", help="skip the C3/C4 roofline runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nb", type=int, default=0, help="outer Cholesky block override")
    ap.add_argument("--inproc", action="store_true",
                    help="N > 1 in ONE process: an engine per device, a host thread each "
                         "(EnginePool), nothing spawned")
    a = ap.parse_args()
    if a.workload is None:
        a.workload = "c2" if a.gpus == 1 else "c5"
    if not a.curve_batch:
        a.curve_batch = (a.batch if a.workload == CURVE_WORKLOAD else 0) or 64
    return a


def self_launch(a):
    """--gpus N > 1 without a launcher: start the N ranks as fresh child processes.
    Nothing in this (parent) process has touched HIP or torch yet, and nothing will."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the ranks re-run THIS file; a wrapper that drives main() itself (the CPU test double,
        # tests/bench_double_main.py) names itself in BQ_BENCH_LAUNCHER -- argv[0] is not trusted
        # (python -c, a harness with its own argv)
        script = os.environ.get("BQ_BENCH_LAUNCHER") or os.path.abspath(__file__)
        if not os.path.isfile(script):
            print("bench.py: launcher %r does not exist" % script, file=sys.stderr)
            sys.exit(2)
        procs.append(subprocess.Popen([sys.executable, script] + sys.argv[1:], env=env))
    rcs = [None] * len(procs)
    # a rank that dies leaves the others in the gloo barrier: stop them instead of waiting
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                try:
                    rcs[i] = p.wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    pass
        if any(rc not in (None, 0) for rc in rcs):
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.terminate()
    bad = [(i, rc) for i, rc in enumerate(rcs) if rc != 0]
    if bad:
        print("bench.py: rank(s) failed: %s" % bad, file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


class Dist(object):
    """Barrier + max over ranks.  torch.distributed (gloo, CPU tensors) when
    launched by torch.distributed.run with WORLD_SIZE > 1; trivial otherwise.
    The GPU work itself never goes through torch."""

    def __init__(self, gpus):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.td = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import torch
            import torch.distributed as td
            td.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self.td, self.torch = td, torch

    def gather(self, v):
        """list of every rank's value (floats)"""
        if self.td is None:
            return [v]
        out = [None] * self.world
        self.td.all_gather_object(out, v)
        return out

    def barrier(self):
        if self.td is not None:
            self.td.barrier()

    def max(self, v):
        if self.td is None:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        return float(t[0])

    def sum(self, v):
        if self.td is None:
            return v
        t = self.torch.tensor([v], dtype=self.torch.float64)
        self.td.all_reduce(t, op=self.td.ReduceOp.SUM)
        return float(t[0])

    def close(self):
        if self.td is not None:
            self.td.destroy_process_group()


def device_count():
    """HIP devices visible to libbqhip.so (0 without a GPU: there is no CPU fallback)."""
    import ctypes as C
    from bayesian_quadrature_amd import _lib as L_
    ndev = C.c_int(0)
    L_.load_library().bq_device_count(C.byref(ndev))
    return ndev.value


# the workload of the 1 -> N scaling curve: a C5 shard per GPU (BASELINE config 5)
CURVE_WORKLOAD = "c5"
NOISE_NOTE = ("s != 0 noise form unpinned by the reference: every reference-held number has "
              "s = 0; Kxx = K + s^2 I is the textbook form recalled for the absent gp package "
              "(BASELINE.md section 3, SURVEY 8c)")


def make_workload_desc(name, batch):
    """The description string make_workload(name, batch, .) puts in `desc` (no data built)."""
    if name == "c2":
        return "C2: d=1 N=1024 M=256 Gaussian integrand, w=dx, s=1e-3"
    return "C5 shard: %d x (d=1 N=2048 M=256), w=dx, s=1e-2" % (batch or 64)


def make_workload(name, batch, rank):
    from bayesian_quadrature_amd import workloads as wl
    if name == "c2":
        c = wl.c2()
        B = batch or 1
        x = np.repeat(c["x"][None], B, axis=0)
        y = np.repeat(c["y"][None], B, axis=0)
        xo = np.repeat(c["xo"][None], B, axis=0)
        desc = make_workload_desc("c2", B)
        return dict(x=x, y=y, xo=xo, h=c["h"], w=c["w"], s=c["s"], d=1, n=1024, M=256, B=B,
                    desc=desc)
    B = batch or 64
    probs = [rank * B + i for i in range(B)]  # each rank owns its own block of problems
    c = wl.c5(probs)
    desc = make_workload_desc("c5", B)
    return dict(x=c["x"], y=c["y"], xo=c["xo"], h=c["h"], w=c["w"], s=c["s"], d=1, n=2048, M=256,
                B=B, desc=desc)


def scale_point(eng, batch, passes=5, warm=3):
    """The N = 1 point of the scaling curve: rank 0's block of the curve workload (a C5 shard)
    on this one GPU, inputs resident, HIP events around `passes` plan passes -- the same
    measurement an N > 1 line reports as n1_same_workload and the same per-GPU work as its
    `value`."""
    wk = make_workload(CURVE_WORKLOAD, batch, 0)
    plan = eng.plan(wk["B"], wk["d"], wk["n"], wk["M"])
    plan.set_inputs(wk["x"], wk["y"], wk["xo"], wk["h"], wk["w"], wk["s"])
    for _ in range(warm):
        plan.run()
    eng.sync()
    eng.timer_start()
    for _ in range(passes):
        plan.run()
    ms = eng.timer_stop_ms() / passes
    status = plan.results()[3]
    plan.close()
    return {"workload": wk["desc"], "value": wk["B"] / ms * 1e3, "unit": "problems/s",
            "ms_per_step": ms, "problems_per_gpu_per_step": wk["B"], "n_gpus": 1,
            "failed": int((status != 0).sum()),
            "note": "the N = 1 point of the C5 curve: N > 1 lines report this workload as "
                    "`value` (whole job) and repeat this measurement as n1_same_workload"}


def host_buffer_step(eng, wk, reps=10):
    """One problem of the timed workload through the host-buffer entry point (bq_fit_predict:
    plan creation, H2D, run, D2H, synchronisation inside every call) -- what a BQ user's call
    costs; never the headline `value`."""
    x, y, xo = wk["x"][0], wk["y"][0], wk["xo"][0]
    eng.fit_predict(x, y, wk["h"], wk["w"], wk["s"], xo)
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.fit_predict(x, y, wk["h"], wk["w"], wk["s"], xo)
    return (time.perf_counter() - t0) / reps * 1e3


def run_main(eng, wk, steps, warmup, dist):
    plan = eng.plan(wk["B"], wk["d"], wk["n"], wk["M"])
    plan.set_inputs(wk["x"], wk["y"], wk["xo"], wk["h"], wk["w"], wk["s"])
    for _ in range(warmup):
        plan.run()
    eng.sync()
    solo_ms = None
    if dist.world > 1:
        # the N = 1 reference of the scaling line, in the same run: rank 0 passes over ITS
        # shard alone while the other ranks wait at the barrier (a few steps, untimed region)
        if dist.rank == 0:
            k = max(3, min(steps, 10))
            eng.timer_start()
            for _ in range(k):
                plan.run()
            solo_ms = eng.timer_stop_ms() / k
        dist.barrier()
    dist.barrier()
    eng.timer_start()
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run()
    eng.sync()
    t1 = time.perf_counter()
    ev_ms = eng.timer_stop_ms()
    dist.barrier()
    wall = dist.max(t1 - t0)
    ranks = dist.gather({"rank": dist.rank, "local_rank": dist.local_rank,
                         "device": eng.device,
                         "ms_per_step": (t1 - t0) / steps * 1e3,
                         "ms_per_step_hip_events": ev_ms / steps, "pid": os.getpid()})
    mean, var, logml, status = plan.results()
    # host issue time of a step: how long the calls take to RETURN (nothing waited for), against
    # the device time of the same steps -- with N contexts driven from one node's cores the
    # launch rate must stay below the device's pace
    ki = max(2, min(steps, 10))
    eng.sync()
    ti0 = time.perf_counter()
    for _ in range(ki):
        plan.run()
    ti1 = time.perf_counter()
    eng.sync()
    ti2 = time.perf_counter()
    issue = {"host_issue_ms_per_step": (ti1 - ti0) / ki * 1e3,
             "device_ms_per_step": (ti2 - ti0) / ki * 1e3}
    ranks_issue = dist.gather(dict(issue, rank=dist.rank))
    # the same K steps with the result read-back (sync + D2H) inside every step
    t2 = time.perf_counter()
    for _ in range(max(1, steps // 4)):
        plan.run()
        plan.results()
    t3 = time.perf_counter()
    rb_ms = (t3 - t2) / max(1, steps // 4) * 1e3
    # instrumented pass: HIP events around every launch, per kernel class
    eng.profile(True)
    eng.profile_reset()
    ninstr = 3
    for _ in range(ninstr):
        plan.run()
    prof = eng.profile_read()
    eng.profile(False)
    for k in prof:
        prof[k]["ms"] /= ninstr
        prof[k]["work"] /= ninstr
        prof[k]["launches"] //= ninstr
    nbytes = plan.nbytes()
    plan.close()
    return dict(wall=wall, ev_ms=ev_ms, rb_ms=rb_ms, prof=prof, mean=mean, var=var, logml=logml,
                status=status, plan_bytes=nbytes, ranks=ranks, solo_ms=solo_ms,
                ranks_issue=ranks_issue)


def parity_spotcheck(wk, res):
    """The bench is not the parity gate (tests/ is) but it refuses to print a
    number for wrong results: problem 0 against the CPU oracle."""
    from oracle import load
    o = load()
    o.set_threads(max(1, min(16, o.max_threads(), len(os.sched_getaffinity(0)))))
    x, y, xo = wk["x"][0], wk["y"][0], wk["xo"][0]
    L, a, lm = o.gp_fit(x, y, wk["h"], wk["w"], wk["s"])
    m, v = o.gp_predict(x, wk["h"], wk["w"], L, a, xo)
    k0 = o.kernel_scale(1, wk["h"], wk["w"])
    return {"mean_rel": float(np.max(np.abs(res["mean"][0] - m)) / np.max(np.abs(m))),
            "var_rel_prior": float(np.max(np.abs(res["var"][0] - v)) / k0),
            "logml_rel": float(abs(res["logml"][0] - lm) / abs(lm)),
            "tolerance": 1e-10, "against": "oracle/bq_oracle.c (CPU restatement)",
            "noise_form": NOISE_NOTE}


def cpu_baseline(wk, budget_s=12.0):
    from oracle import load
    o = load()
    # the box gives one GPU a 16-core share; more OpenMP threads than that only thrash
    cores = max(1, min(16, o.max_threads(), len(os.sched_getaffinity(0))))
    o.set_threads(cores)
    x, y, xo = wk["x"][0], wk["y"][0], wk["xo"][0]
    o.gp_fit(x, y, wk["h"], wk["w"], wk["s"])  # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        L, a, lm = o.gp_fit(x, y, wk["h"], wk["w"], wk["s"])
        o.gp_predict(x, wk["h"], wk["w"], L, a, xo)
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 200:
            break
    # the same problem on ONE thread, a few repetitions (1 thread and all cores)
    o.set_threads(1)
    r1, t1 = 0, time.perf_counter()
    while True:
        L, a, lm = o.gp_fit(x, y, wk["h"], wk["w"], wk["s"])
        o.gp_predict(x, wk["h"], wk["w"], L, a, xo)
        r1 += 1
        e1 = time.perf_counter() - t1
        if e1 > 3.0 or r1 >= 20:
            break
    o.set_threads(cores)
    return {"value": reps / el, "unit": "problems/s", "cores": cores, "kind": "port",
            "reference_timings_baseline_md": "the reference's own Cython + OpenBLAS path, "
            "measured in the survey container (8 vCPU Xeon 2.1 GHz, BASELINE.md section 2), "
            "N=1024: Gram 18.8 ms + cho_factor 11.9 ms + cho_solve_vec 0.97 ms = 31.7 ms; "
            "N=2048: 85.4 + 67.8 + 3.3 ms; N=4096: 373 + 477 + 15.4 ms.  It cannot run on "
            "the GPU box (Python 2 + the absent gp package)",
            "ms_per_problem": el / reps * 1e3,
            "one_thread": {"value": r1 / e1, "ms_per_problem": e1 / r1 * 1e3, "reps": r1},
            "sample": "%d x (gram + blocked potrf + potrs + predict mean/var + logML) of the "
                      "N=%d, M=%d problem, oracle/bq_oracle.c with OpenMP on %d threads, %.1f s"
                      % (reps, wk["n"], wk["M"], cores, el)}


def extras(eng, nb_override):
    """C4 (N=16384 Cholesky, MFMA roofline) and C3 (N=4096 d=2 Gram, HBM roofline)."""
    from bayesian_quadrature_amd import workloads as wl
    from bayesian_quadrature_amd import _lib as L_
    out = {}
    lib, ctx = eng._lib, eng._ctx
    # ---- Gram, N=4096 d=2 and N=16384 d=1 --------------------------------------
    for tag, n, d in (("gram_n4096_d2", 4096, 2), ("gram_n16384_d1", 16384, 1)):
        if d == 2:
            c = wl.c3()
            pts, h, w, s = np.asfortranarray(c["x"]), float(c["h"][200]), c["w"][200], c["s"]
        else:
            c = wl.c4(n)
            pts, h, w, s = np.asfortranarray(c["x"][None, :]), c["h"], c["w"], c["s"]
        w = np.ascontiguousarray(w, dtype=np.float64)
        xd = eng.alloc(8 * d * n)
        Kd = eng.alloc(8 * n * n)
        eng.upload(xd, pts)
        reps = 20 if n <= 4096 else 6
        for _ in range(3):
            eng._check(lib.bq_gram_gauss_dev(ctx, xd, d, n, h, L_.dptr(w), s, Kd, n))
        eng.sync()
        eng.timer_start()
        for _ in range(reps):
            eng._check(lib.bq_gram_gauss_dev(ctx, xd, d, n, h, L_.dptr(w), s, Kd, n))
        ms = eng.timer_stop_ms() / reps
        alg = 8.0 * n * n + 8.0 * d * n
        traffic, src = pmc_traffic("gram_tri_kernel<%d>" % d)
        out[tag] = {"kernel": "gram_tri_kernel<%d>" % d, "bound": "hbm",
                    "achieved": alg / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": alg / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": traffic,
                    "traffic_source": src, "ms_per_launch": ms, "algorithmic_bytes": alg,
                    "note": "HIP events around %d back-to-back launches" % reps}
        if n == 16384:
            # ---- C4: potrf on the matrix just built ------------------------------
            info = eng.alloc(64)
            # BASELINE config 4 names the blocking: tile = 256.  The engine's own choice at
            # this size (512) is timed below as potrf_n16384_engine_block.
            nb = nb_override or 256
            eng.set_block(nb)
            tfl = wl.trailing_flops(n, nb)

            def potrf_runs(reps):
                ts = []
                for rep in range(reps):
                    eng._check(lib.bq_gram_gauss_dev(ctx, xd, d, n, h, L_.dptr(w), s, Kd, n))
                    eng.sync()
                    eng.timer_start()
                    eng._check(lib.bq_potrf_dev(ctx, Kd, n, n, info))
                    ts.append(eng.timer_stop_ms())
                return ts

            def potrf_profile():
                eng._check(lib.bq_gram_gauss_dev(ctx, xd, d, n, h, L_.dptr(w), s, Kd, n))
                eng.profile(True)
                eng.profile_reset()
                eng._check(lib.bq_potrf_dev(ctx, Kd, n, n, info))
                pr = eng.profile_read()
                eng.profile(False)
                return pr

            hinfo = np.zeros(1, dtype=np.int32)
            # (1) the pipeline as shipped: look-ahead on two streams
            eng.set_lookahead(True)
            t_la = potrf_runs(3)
            eng.download(hinfo, info)
            prof_la = potrf_profile()
            # (2) strictly sequential launches: the kernels timed in isolation
            eng.set_lookahead(False)
            t_seq = potrf_runs(3)
            prof = potrf_profile()
            eng.set_lookahead(True)
            best = min(t_la[1:])
            sy, sy_la = prof["syrk_trailing"], prof_la["syrk_trailing"]
            sm = prof["syrk_trailing_small"]
            out["potrf_n16384"] = {
                "ms": best, "gflops": wl.potrf_flops(n) / (best * 1e-3) / 1e9, "nb": nb,
                "info": int(hinfo[0]), "lookahead_ms": t_la, "sequential_ms": t_seq,
                "sequential_gflops": wl.potrf_flops(n) / (min(t_seq[1:]) * 1e-3) / 1e9,
                "class_ms_sequential": {k: v["ms"] for k, v in prof.items()},
                "class_ms_lookahead": {k: v["ms"] for k, v in prof_la.items()},
                "class_launches_sequential": {k: v["launches"] for k, v in prof.items()}}
            traffic, src = pmc_traffic(TRAILING_KERNEL)
            ach = sy["work"] / (sy["ms"] * 1e-3) / 1e12
            out["trailing_update_n16384"] = {
                "kernel": TRAILING_KERNEL, "bound": "mfma", "achieved": ach,
                "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP64_TFLOPS,
                "traffic": traffic, "traffic_source": src,
                "algorithmic_flops_per_launch": sy["work"] / max(1, sy["launches"]),
                "launches": sy["launches"], "ms_total": sy["ms"],
                "ms_per_launch": sy["ms"] / max(1, sy["launches"]),
                "whole_trailing_update": {
                    "algorithmic_flops": tfl, "ms_total": sy["ms"] + sm["ms"],
                    "launches": sy["launches"] + sm["launches"],
                    "achieved": tfl / ((sy["ms"] + sm["ms"]) * 1e-3) / 1e12,
                    "frac": tfl / ((sy["ms"] + sm["ms"]) * 1e-3) / 1e12 / PEAK_FP64_TFLOPS},
                "in_lookahead_pipeline": {
                    "ms_total": sy_la["ms"] + prof_la["syrk_trailing_small"]["ms"],
                    "achieved": tfl / ((sy_la["ms"] + prof_la["syrk_trailing_small"]["ms"])
                                       * 1e-3) / 1e12,
                    "note": "same flops while the next block's panel kernels share the GPU"},
                "note": "sequential launches (look-ahead off), HIP events per launch; the "
                        "algorithmic flops of a trailing launch are m^2 nb (lower half)"}
            # the same factorisation on DENSE operands (workloads.c4_dense): the kernel's figure
            cd = wl.c4_dense(n)
            wd = np.ascontiguousarray(cd["w"])

            def dense_profile():
                eng._check(lib.bq_gram_gauss_dev(ctx, xd, d, n, cd["h"], L_.dptr(wd), cd["s"], Kd, n))
                eng.profile(True)
                eng.profile_reset()
                eng._check(lib.bq_potrf_dev(ctx, Kd, n, n, info))
                pr = eng.profile_read()
                eng.profile(False)
                return pr

            eng.set_lookahead(False)
            dense_profile()
            prd = dense_profile()
            eng.download(hinfo, info)
            eng.set_lookahead(True)
            syd, smd = prd["syrk_trailing"], prd["syrk_trailing_small"]
            achd = syd["work"] / (syd["ms"] * 1e-3) / 1e12
            whole_d = tfl / ((syd["ms"] + smd["ms"]) * 1e-3) / 1e12
            out["trailing_update_n16384_dense"] = {
                "whole_trailing_update": {
                    "algorithmic_flops": tfl, "ms_total": syd["ms"] + smd["ms"],
                    "launches": syd["launches"] + smd["launches"], "achieved": whole_d,
                    "frac": whole_d / PEAK_FP64_TFLOPS},
                "kernel": TRAILING_KERNEL, "bound": "mfma", "achieved": achd,
                "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": achd / PEAK_FP64_TFLOPS,
                "traffic": traffic, "traffic_source": src, "launches": syd["launches"],
                "ms_total": syd["ms"], "ms_per_launch": syd["ms"] / max(1, syd["launches"]),
                "info": int(hinfo[0]),
                "note": "the same N, tile and launches as trailing_update_n16384 on a Gram that is "
                        "not banded (w = 200 dx, s = 1: workloads.c4_dense); C4's own data feed "
                        "the update > 97 % exact zeros, which clock higher"}
            eng.set_block(nb_override)
            t_auto = potrf_runs(3)
            out["potrf_n16384_engine_block"] = {
                "ms": min(t_auto[1:]), "gflops": wl.potrf_flops(n) / (min(t_auto[1:]) * 1e-3) / 1e9,
                "note": "outer block chosen by the engine (512 at this size) instead of the "
                        "config's 256"}
            eng.free(info)
        eng.free(xd)
        eng.free(Kd)
    out.update(solve_predict_rooflines(eng))
    out.update(hyper_loop_body(eng))
    out.update(batched_configs(eng))
    try:
        out.update(sustained_clock(eng))
    except Exception as exc:  # context only: never fail the bench line for it
        out["sustained_clock"] = None
        out["sustained_clock_note"] = "not measured: %r" % (exc,)
    # the same C2 problem through the host-buffer entry point (allocation, PCIe both
    # ways, synchronisation inside every call): never the headline `value`
    c2 = wl.c2()
    eng.fit_predict(c2["x"], c2["y"], c2["h"], c2["w"], c2["s"], c2["xo"])
    t0 = time.perf_counter()
    for _ in range(10):
        eng.fit_predict(c2["x"], c2["y"], c2["h"], c2["w"], c2["s"], c2["xo"])
    out["c2_host_buffer_call"] = {"ms_per_call": (time.perf_counter() - t0) / 10 * 1e3,
                                  "note": "bq_fit_predict: plan creation + H2D + run + D2H per call"}
    return out


def _prof_call(eng, fn, reps=3):
    """Time of fn(): the host wall time of a call (median of five groups of `reps` calls,
    launch profiler off), then HIP events around every kernel launch (per class), averaged
    over `reps` calls.  The event brackets cost a few microseconds per launch: for a chain of
    many short launches their sum exceeds the unprofiled wall time."""
    fn()
    eng.sync()
    walls = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        eng.sync()
        walls.append((time.perf_counter() - t0) / reps)
    wall = sorted(walls)[2]
    eng.profile(True)
    eng.profile_reset()
    for _ in range(reps):
        fn()
    pr = eng.profile_read()
    eng.profile(False)
    ms = {k: v["ms"] / reps for k, v in pr.items() if v["launches"]}
    return sum(ms.values()), ms, wall * 1e3


def solve_predict_rooflines(eng):
    """the triangular solves (cho_solve: 8 N^2 bytes for one right-hand side --
    two passes over the lower triangle --, 2 N^2 flop per right-hand side) and the posterior
    over M points (mean M N kernel evaluations, variance M N^2 flop through the forward
    sweep) on resident fits."""
    from bayesian_quadrature_amd import workloads as wl
    out = {}
    for n in (4096, 16384):
        c = wl.c4(n)
        y = wl.norm_logpdf(c["x"])
        fit = eng.gp_fit(c["x"], y, c["h"], c["w"], c["s"])
        rs = np.random.RandomState(n)
        b1 = rs.randn(n)
        dev, cls, wall = _prof_call(eng, lambda: fit.solve(b1), reps=10)
        byt = 8.0 * n * n
        # the headline is the unprofiled wall time of the whole call (hipGraph replay, host
        # vector in and out); the sum of the per-launch event brackets (eager launches, a
        # different execution mode) is informational only
        t1 = wall
        out["cho_solve_n%d_rhs1" % n] = {
            "bound": "hbm", "achieved": byt / (t1 * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": byt / (t1 * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "ms": t1, "ms_event_brackets": dev, "ms_call_host_buffers": wall, "class_ms": cls,
            "algorithmic_bytes": byt,
            "traffic": _trsv_traffic() if n == 16384 else None,
            "note": "bq_gp_solve on a resident factor, one right-hand side: the GEMV sweeps "
                    "of trsv.h as ONE launch per sweep (trsvflow.h: all steps' workgroups in one "
                    "grid, hand-offs through sentinel-filled slots); ms = unprofiled wall time of "
                    "the whole call (median of five groups), host vector in and out; "
                    "ms_event_brackets = the two launches under the launch profiler"}
        B = np.asfortranarray(rs.randn(n, 256))
        dev, cls, wall = _prof_call(eng, lambda: fit.solve(B), reps=2)
        fl = 2.0 * n * n * 256
        out["cho_solve_n%d_rhs256" % n] = {
            "bound": "mfma", "achieved": fl / (dev * 1e-3) / 1e12, "peak": PEAK_FP64_TFLOPS,
            "unit": "TFLOP/s", "frac": fl / (dev * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
            "traffic": None, "ms_kernels": dev, "ms_call_host_buffers": wall, "class_ms": cls,
            "algorithmic_flops": fl}
        fit.close()
    c2 = wl.c2()
    fit = eng.gp_fit(c2["x"], c2["y"], c2["h"], c2["w"], c2["s"])
    for M in (256, 1000):
        xo = np.linspace(-5.0, 5.0, M) + 1e-3
        dev, cls, wall = _prof_call(eng, lambda: fit.predict(xo), reps=5)
        fl = float(M) * 1024 * 1024
        out["predict_mean_var_n1024_m%d" % M] = {
            "bound": "mfma", "achieved": fl / (dev * 1e-3) / 1e12, "peak": PEAK_FP64_TFLOPS,
            "unit": "TFLOP/s", "frac": fl / (dev * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
            "traffic": None, "ms_kernels": dev, "ms_call_host_buffers": wall, "class_ms": cls,
            "algorithmic_flops": fl,
            "note": "bq_gp_predict mean + variance: cross Gram, forward sweep V = K(xo,x) L^-T "
                    "(M N^2 flop), row reductions; a chain of 16 x 2 dependent launches, "
                    "latency bound at this N"}
        dev, cls, wall = _prof_call(eng, lambda: fit.predict(xo, want_var=False), reps=5)
        out["predict_mean_n1024_m%d" % M] = {
            "bound": "hbm", "achieved": 8.0 * (2 * 1024 + M) / (dev * 1e-3) / 1e9,
            "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": 8.0 * (2 * 1024 + M) / (dev * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None,
            "ms_kernels": dev, "ms_call_host_buffers": wall,
            "kernel_evaluations": float(M) * 1024,
            "note": "fused cross-Gram x alpha (predict_mean_kernel): M N exp evaluations, "
                    "8 (N + M + N) algorithmic bytes -- a single short launch"}
    fit.close()
    return out


def hyper_loop_body(eng):
    """the body of the hyper-parameter loop (bq.py:536-550, 933-965) on
    resident fits -- GP1: new parameters and the posterior of the candidates (one sweep,
    bq_gp_refit_predict); GP2: refit on samples + candidates; the two log-MLs.  Wall time per
    iteration, host buffers and read-backs included (a latency chain: report only)."""
    from bayesian_quadrature_amd import workloads as wl
    out = {}
    for n, nc in ((19, 10), (1024, 10)):
        dx = 10.0 / (n - 1)
        x = np.linspace(-5.0, 5.0, n)
        xc = np.linspace(-5.5, 5.5, nc) + 0.37 * dx
        g1 = eng.gp_fit(x, wl.norm_logpdf(x), 15.0, 1.3 * dx, 1e-3)
        xsc = np.concatenate([x, xc])
        g2 = eng.gp_fit(xsc, np.exp(wl.norm_logpdf(xsc)), 0.2, 1.3 * dx, 1e-3)
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for it in range(20):
                wi = (1.3 + 0.001 * it) * dx
                m, v = g1.refit_predict(15.0, wi, 1e-3, xc)
                g2.refit(0.2, wi, 1e-3)
                _ = g1.logml + g2.logml
            ts.append((time.perf_counter() - t0) / 20 * 1e3)
        out["hyper_loop_body_n%d_nc%d" % (n, nc)] = {
            "ms_per_iteration": sorted(ts)[2], "bound": "latency",
            "note": "GP1 refit + candidates' mean/variance in one sweep, GP2 refit, both "
                    "log-MLs; wall time with host buffers"}
        g1.close()
        g2.close()
    return out


def _hwmon_sample():
    """(watts, sclk MHz) of the busiest card that exposes a hwmon node: plain sysfs reads
    (power1_input in microwatts, freq1_input in Hz), no child process."""
    import glob
    best = None
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        try:
            w = int(open(h + "/power1_input").read()) / 1e6
            f = int(open(h + "/freq1_input").read()) / 1e6
        except (OSError, ValueError):
            continue
        if best is None or w > best[0]:
            best = (w, f)
    return best


def sustained_clock(eng):
    """What the chip sustains while the product runs: socket power and shader clock sampled from
    sysfs during ~1.5 s of (a) the fp64 MFMA loop without memory and (b) the trailing-update
    kernel on random operands.  The MFMA peak of MICROARCH.md (78.6 TFLOP/s) is quoted at
    2.4 GHz; under the power cap the product runs below that clock, and `peak_at_clock` is what
    the matrix cores could deliver at the clock it actually had.  Context for the fractions
    above, never a replacement for them."""
    import threading
    if _hwmon_sample() is None:
        return {"sustained_clock": None, "sustained_clock_note": "no hwmon node readable"}

    def watch(fn):
        res, seen = {}, []
        th = threading.Thread(target=lambda: res.setdefault("v", fn()))
        th.start()
        time.sleep(0.5)
        while th.is_alive():
            seen.append(_hwmon_sample())
            time.sleep(0.2)
        th.join()
        seen = seen[:-1] or seen
        return res["v"], sum(x[0] for x in seen) / len(seen), sum(x[1] for x in seen) / len(seen)

    out = {"at_entry": dict(zip(("watts", "sclk_mhz"), _hwmon_sample()))}
    v, w, f = watch(lambda: [eng.probe_mfma_variant(5, 8, 2) for _ in range(6)])
    out["mfma_loop_no_memory"] = {"tflops": sum(v) / len(v), "watts": w, "sclk_mhz": f}
    for tag, m, k, b, reps in (("trailing_update_m16064_k256", 16064, 256, 1, 1400),
                               ("trailing_update_m2048_k320_batch32", 2048, 320, 32, 1800)):
        ms, w, f = watch(lambda: eng.probe_gemm(m, m, k, 1, b, False, reps))
        tf = float(m) * m * k * b / ms / 1e9
        peak = PEAK_FP64_TFLOPS * f / 2400.0
        out[tag] = {"tflops": tf, "watts": w, "sclk_mhz": f, "peak_at_clock": peak,
                    "frac_of_peak_at_clock": tf / peak, "frac": tf / PEAK_FP64_TFLOPS,
                    "note": "random operands, %d launches back to back" % reps}
    out["note"] = ("sysfs power1_input / freq1_input sampled every 0.2 s while the kernel runs; "
                   "the product is power-bound: it holds the socket at its cap and the clock "
                   "below 2.4 GHz, the MFMA loop without memory does not")
    return {"sustained_clock": out}


def batched_configs(eng):
    """The two batched BASELINE configs on this one GPU, for the record (not the
    headline): a C5 shard (64 of the 512 problems, N=2048, M=256) and the C3
    hyper-grid (20 x 20 log-MLs at N=4096, d=2)."""
    from bayesian_quadrature_amd import workloads as wl
    out = {}
    B = 64
    c = wl.c5(list(range(B)))
    plan = eng.plan(B, 1, 2048, 256)
    plan.set_inputs(c["x"], c["y"], c["xo"], c["h"], c["w"], c["s"])
    # (three untimed passes: the first passes of a batch run 2-4 % slower than the steady state
    # the scaling curve is about -- tools/c5_time.py prints every pass)
    for _ in range(3):
        plan.run()
    eng.sync()
    eng.timer_start()
    for _ in range(5):
        plan.run()
    ms = eng.timer_stop_ms() / 5
    status = plan.results()[3]
    plan.close()
    flops = B * (2048 ** 3 / 3.0)   # the factorisation alone, per problem N^3/3
    # the posterior's M N^2 (marginal variance through the forward sweep).  Round 1
    # and the first half of round 2 also executed and counted the M^2 N of the full posterior
    # covariance; the sweep no longer computes that block (plan_readout_kernel)
    post = B * (256.0 * 2048 * 2048)
    out["c5_shard_64x2048"] = {
        "ms_per_batch": ms, "problems_per_s": B / ms * 1e3, "failed": int((status != 0).sum()),
        "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_FP64_TFLOPS,
        "achieved": (flops + post) / (ms * 1e-3) / 1e12,
        "frac": (flops + post) / (ms * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
        "algorithmic_flops": flops + post,
        "potrf_tflops_lower_bound": flops / (ms * 1e-3) / 1e12,
        "note": "wall-derived: (N^3/3 + M N^2) per problem over the HIP-event time of a "
                "plan pass; the kernel classes of this shard and the MFMA utilisation of its "
                "gemm_lds_kernel launches are in profiles/r06_c5_kernel_stats.csv and "
                "profiles/r06_mfma_util.json; the pass's launch timeline in "
                "profiles/r06_plan_timeline_c5.txt"}
    # the headline problem, 256 independent copies per step: what batching buys over the
    # latency-bound single problem of `value`
    c2 = wl.c2()
    B2 = 256
    plan = eng.plan(B2, 1, 1024, 256)
    plan.set_inputs(np.repeat(c2["x"][None], B2, axis=0), np.repeat(c2["y"][None], B2, axis=0),
                    np.repeat(c2["xo"][None], B2, axis=0), c2["h"], c2["w"], c2["s"])
    for _ in range(3):
        plan.run()
    eng.sync()
    eng.timer_start()
    for _ in range(5):
        plan.run()
    ms2 = eng.timer_stop_ms() / 5
    st2 = plan.results()[3]
    plan.close()
    fl2 = B2 * (1024 ** 3 / 3.0 + 256.0 * 1024 * 1024)
    out["c2_batch_256x1024"] = {"ms_per_batch": ms2, "problems_per_s": B2 / ms2 * 1e3,
                                "failed": int((st2 != 0).sum()), "bound": "mfma",
                                "unit": "TFLOP/s", "peak": PEAK_FP64_TFLOPS,
                                "achieved": fl2 / (ms2 * 1e-3) / 1e12,
                                "frac": fl2 / (ms2 * 1e-3) / 1e12 / PEAK_FP64_TFLOPS,
                                "algorithmic_flops": fl2}
    c3 = wl.c3()
    # one untimed chunk first: the first call pays the 100 x 128 MiB workspace allocation
    eng.logml_grid(c3["x"], c3["y"], c3["h"][:100], c3["w"][:100], c3["s"], chunk=100)
    walls = []
    for _ in range(3):
        t0 = time.perf_counter()
        lm = eng.logml_grid(c3["x"], c3["y"], c3["h"], c3["w"], c3["s"], chunk=100)
        walls.append(time.perf_counter() - t0)
    wall = min(walls)
    fl3 = len(lm) * (4096 ** 3 / 3.0)
    out["c3_grid_400x4096"] = {"wall_ms": wall * 1e3, "wall_ms_all": [w * 1e3 for w in walls],
                               "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_FP64_TFLOPS,
                               "achieved": fl3 / wall / 1e12,
                               "frac": fl3 / wall / 1e12 / PEAK_FP64_TFLOPS,
                               "algorithmic_flops": fl3,
                               "ms_per_point": wall * 1e3 / len(lm),
                               "n_minus_inf": int(np.isinf(lm).sum()),
                               "note": "wall-derived: N^3/3 per grid point over the host wall "
                                       "clock incl. the upload of the 400 parameter sets and the "
                                       "read-back of the 400 results; kernel classes of one chunk "
                                       "in profiles/r06_c3_kernel_stats.csv"}
    return out


def fit_posterior_at_n(eng, sizes=(1024, 2048, 4096, 16384), M=256):
    """BASELINE.json's metric read literally -- "BQ fit+posterior ms at N": ONE problem (Gram,
    Cholesky, mean + variance at M = 256 points, log-ML) through the bordered plan with the inputs
    resident, and through bq_fit_predict with host buffers, at N = 1024 ... 16384 (C2's data
    shape: 1-D linspace, w = dx, s = 1e-3).  Work: N^3/3 + M N^2 flop."""
    from bayesian_quadrature_amd import workloads as wl
    out = {}
    for n in sizes:
        c = wl.c2(n, M)
        reps = 20 if n <= 2048 else (8 if n <= 4096 else 3)
        plan = eng.plan(1, 1, n, M)
        plan.set_inputs(c["x"][None], c["y"][None], c["xo"][None], c["h"], c["w"], c["s"])
        for _ in range(2):
            plan.run()
        eng.sync()
        eng.timer_start()
        for _ in range(reps):
            plan.run()
        ms = eng.timer_stop_ms() / reps
        status = plan.results()[3]
        plan.close()
        eng.fit_predict(c["x"], c["y"], c["h"], c["w"], c["s"], c["xo"])
        t0 = time.perf_counter()
        for _ in range(max(2, reps // 2)):
            eng.fit_predict(c["x"], c["y"], c["h"], c["w"], c["s"], c["xo"])
        ms_host = (time.perf_counter() - t0) / max(2, reps // 2) * 1e3
        fl = n ** 3 / 3.0 + float(M) * n * n
        out["n%d" % n] = {"n": n, "m": M, "ms": ms, "ms_host_buffers": ms_host,
                          "failed": int((status != 0).sum()), "algorithmic_flops": fl,
                          "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_FP64_TFLOPS,
                          "achieved": fl / (ms * 1e-3) / 1e12,
                          "frac": fl / (ms * 1e-3) / 1e12 / PEAK_FP64_TFLOPS}
    out["note"] = ("one problem per step, HIP events around back-to-back plan passes (inputs "
                   "resident) / host wall per bq_fit_predict call (plan creation, H2D, run, D2H, "
                   "synchronisation); kernel classes per size: profiles/r06_fitpost_n*_kernel_stats.csv")
    return out


NORTHSTAR_KEYS = ("trail16k_frac", "trail16k_dense_frac", "gram4096_frac", "gram16k_frac",
                  "c5_frac", "c3_frac")


def northstar_first(roof, rl):
    """`roofline` with the north-star scalars FIRST, under short names (<= 24 characters): the
    driver's record keeps about the first twenty keys of the object and cuts names at 40
    characters (BENCH_r05.json lost the Gram fractions that way).  Each scalar equals the object
    of `rooflines` named in `northstar_source`; the long definitions live there."""
    head = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")
    out = {k: roof[k] for k in head if k in roof}
    # 1.4318e12 flop / sum of ALL trailing-update launch time at N = 16384, nb = 256
    out["trail16k_frac"] = rl["trailing_update_n16384"]["whole_trailing_update"]["frac"]
    out["trail16k_dense_frac"] = rl["trailing_update_n16384_dense"]["whole_trailing_update"]["frac"]
    out["gram4096_frac"] = rl["gram_n4096_d2"]["frac"]
    out["gram16k_frac"] = rl["gram_n16384_d1"]["frac"]
    out["c5_frac"] = rl["c5_shard_64x2048"]["frac"]
    out["c3_frac"] = rl["c3_grid_400x4096"]["frac"]
    fp = rl.get("fit_posterior_ms_at_n", {})
    for n in (1024, 2048, 4096, 16384):
        if "n%d" % n in fp:
            out["fitpost_n%d_ms" % n] = fp["n%d" % n]["ms"]
    out["c2x256_frac"] = rl["c2_batch_256x1024"]["frac"]
    out["c5_ms"] = rl["c5_shard_64x2048"]["ms_per_batch"]
    out["c3_ms"] = rl["c3_grid_400x4096"]["wall_ms"]
    out["potrf16k_ms"] = rl["potrf_n16384"]["ms"]
    out["c2x256_ms"] = rl["c2_batch_256x1024"]["ms_per_batch"]
    for k, v in roof.items():
        if k not in out:
            out[k] = v
    rl["northstar_source"] = {
        "trail16k_frac": "rooflines.trailing_update_n16384.whole_trailing_update (C4's banded data): "
                         "sum_k m_k^2 nb = 1.4318e12 flop / the time of every trailing-update "
                         "launch, sequential launches, HIP events per launch",
        "trail16k_dense_frac": "rooflines.trailing_update_n16384_dense.whole_trailing_update",
        "trail16k_bulk_launches": "rooflines.trailing_update_n16384[_dense].frac: the bulk "
                                  "gemm_lds_kernel launches alone",
        "gram4096_frac": "rooflines.gram_n4096_d2 (134 MB: Infinity-Cache resident, not an HBM "
                         "figure)",
        "gram16k_frac": "rooflines.gram_n16384_d1", "c5_frac": "rooflines.c5_shard_64x2048",
        "c3_frac": "rooflines.c3_grid_400x4096", "c2x256_frac": "rooflines.c2_batch_256x1024",
        "potrf16k_ms": "rooflines.potrf_n16384 (tile 256; engine block: "
                       "rooflines.potrf_n16384_engine_block)",
        "fitpost_n*_ms": "rooflines.fit_posterior_ms_at_n"}
    return out


def curve_fields(line, wk, solo_ms, world):
    """Top-level fields of an N > 1 line from which a reader builds the 1 -> N curve without
    any other file: the same workload's N = 1 throughput measured in this run (rank 0 passes
    over its shard alone while the others wait) and the efficiency against it."""
    if solo_ms is None or world <= 1:
        return
    solo = wk["B"] / solo_ms * 1e3
    line["n1_same_workload"] = {
        "workload": wk["desc"], "value": solo, "unit": "problems/s", "ms_per_step": solo_ms,
        "n_gpus": 1, "problems_per_gpu_per_step": wk["B"],
        "note": "rank 0's shard of the same workload run alone (other ranks idle at the "
                "barrier) just before the timed region; equals scale_point.value of the "
                "--gpus 1 line"}
    line["scaling_efficiency"] = line["value"] / (solo * world) if line["value"] else None
    line["scaling_efficiency_note"] = ("value / (n_gpus x n1_same_workload.value): weak scaling, "
                                       "per-GPU work fixed (a block of %d problems per rank)"
                                       % wk["B"])


def inproc_main(a):
    """--gpus N --inproc: the second launch mode ("one Python thread per device").
    This process creates the N contexts itself -- an EnginePool, one host thread per engine --
    and every engine passes over its own block of problems; the threads meet at a barrier
    before and after the timed region, the time is first start to last end."""
    import threading
    from bayesian_quadrature_amd import EnginePool
    ndev = device_count()
    share = os.environ.get("BQ_BENCH_SHARE_DEVICE", "0") == "1"
    if ndev <= 0 or (ndev < a.gpus and not share):
        print("bench.py: --gpus %d --inproc needs %d HIP devices, this box has %d"
              % (a.gpus, a.gpus, ndev), file=sys.stderr)
        sys.exit(2)
    pool = EnginePool([r % ndev for r in range(a.gpus)])
    barrier = threading.Barrier(a.gpus)

    def job(rank):
        def body(eng):
            wk = make_workload(a.workload, a.batch, rank)
            plan = eng.plan(wk["B"], wk["d"], wk["n"], wk["M"])
            plan.set_inputs(wk["x"], wk["y"], wk["xo"], wk["h"], wk["w"], wk["s"])
            for _ in range(a.warmup):
                plan.run()
            eng.sync()
            # the N = 1 point of the curve in the same run: rank 0 alone, the others wait
            solo_ms = None
            barrier.wait(timeout=600)
            if rank == 0:
                k = max(3, min(a.steps, 10))
                eng.timer_start()
                for _ in range(k):
                    plan.run()
                solo_ms = eng.timer_stop_ms() / k
            barrier.wait(timeout=600)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                plan.run()
            eng.sync()
            t1 = time.perf_counter()
            barrier.wait(timeout=600)
            mean, var, logml, status = plan.results()
            # host issue time of a step beside its device time, all ranks' threads at once
            ki = max(2, min(a.steps, 10))
            barrier.wait(timeout=600)
            ti0 = time.perf_counter()
            for _ in range(ki):
                plan.run()
            ti1 = time.perf_counter()
            eng.sync()
            ti2 = time.perf_counter()
            plan.close()
            return dict(rank=rank, device=eng.device, t0=t0, t1=t1, wk=wk, mean=mean, var=var,
                        logml=logml, status=status, info=eng.info() if rank == 0 else None,
                        solo_ms=solo_ms, host_issue_ms_per_step=(ti1 - ti0) / ki * 1e3,
                        device_ms_per_step=(ti2 - ti0) / ki * 1e3)

        def run(eng):
            # a rank that fails before a barrier (allocation, inputs, a HIP error on its device)
            # breaks the barrier for the others: they raise BrokenBarrierError instead of
            # waiting for ever, pool.run re-raises the first failure and the process exits 1
            try:
                return body(eng)
            except BaseException:
                barrier.abort()
                raise
        return run

    try:
        res = pool.run([job(r) for r in range(a.gpus)])
    except BaseException as exc:
        # (the first failure in rank order; ranks that only saw the broken barrier come after it)
        print("bench.py --inproc: a rank failed: %r" % (exc,), file=sys.stderr)
        pool.close()
        sys.exit(1)
    pool.close()
    wall = max(r["t1"] for r in res) - min(r["t0"] for r in res)
    wk = res[0]["wk"]
    nfail = int(sum((r["status"] != 0).sum() for r in res))
    units = a.steps * wk["B"] * a.gpus
    line = {
        "metric": "bq_fit_posterior_throughput", "value": units / wall, "unit": "problems/s",
        "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": wk["desc"], "value_is": "problems/s of the whole job on: " + wk["desc"],
                   "curve_workload": wk["desc"], "problems_per_gpu_per_step": wk["B"],
                   "sharding": "independent problems per device, no data-path collective",
                   "launch_mode": "inproc: one process, an engine and a host thread per device",
                   "ranks": [{"rank": r["rank"], "device": r["device"],
                              "ms_per_step": (r["t1"] - r["t0"]) / a.steps * 1e3} for r in res]},
        "host_issue": [{"rank": r["rank"], "host_issue_ms_per_step": r["host_issue_ms_per_step"],
                        "device_ms_per_step": r["device_ms_per_step"]} for r in res],
        "failed_problems": nfail, "device": res[0]["info"],
        "parity": parity_spotcheck(wk, res[0]),
    }
    curve_fields(line, wk, res[0]["solo_ms"], a.gpus)
    p = line["parity"]
    if not (p["mean_rel"] < 1e-10 and p["var_rel_prior"] < 1e-10 and p["logml_rel"] < 1e-10) \
            or nfail:
        line["value"] = None
        line["invalid"] = "parity check failed"
    print(json.dumps(line))
    sys.exit(0 if line["value"] is not None else 1)


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world == 1 and a.inproc:
        inproc_main(a)  # does not return
    if a.gpus > 1 and world == 1:
        self_launch(a)  # does not return
    if a.gpus != world:
        print("bench.py: --gpus %d but WORLD_SIZE %d" % (a.gpus, world), file=sys.stderr)
        sys.exit(2)
    # The engine (HIP context on LOCAL_RANK) exists before torch is imported; torch is used
    # for the gloo barrier and reductions on CPU tensors only and never initialises HIP.
    from bayesian_quadrature_amd import Engine
    ndev = device_count()
    if ndev <= 0:
        print("bench.py needs a HIP device (no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one device per rank; BQ_BENCH_SHARE_DEVICE=1 lets several ranks share a device (the
    # two-rank rehearsal on a one-GPU box in tests/test_sharding.py), never a measurement
    share = os.environ.get("BQ_BENCH_SHARE_DEVICE", "0") == "1"
    if ndev < world and not share:
        print("bench.py: --gpus %d needs %d HIP devices, this box has %d"
              % (a.gpus, world, ndev), file=sys.stderr)
        sys.exit(2)
    eng = Engine(local_rank % ndev)
    dist = Dist(a.gpus)
    if a.nb:
        eng.set_block(a.nb)
    wk = make_workload(a.workload, a.batch, dist.rank)
    res = run_main(eng, wk, a.steps, a.warmup, dist)
    nfail = dist.sum(float((res["status"] != 0).sum()))
    units = a.steps * wk["B"] * dist.world  # problems processed by the whole job
    value = units / res["wall"]
    line = None
    if dist.rank == 0:
        info = eng.info()
        npad = -(-wk["n"] // 64) * 64
        ntot = -(-(npad + wk["M"] + 1) // 64) * 64
        prof = res["prof"]
        dom = max(prof, key=lambda k: prof[k]["ms"])
        dom_ms = prof[dom]["ms"]
        per_launch_ms = dom_ms / max(1, prof[dom]["launches"])
        if dom in ("gram", "reduce"):
            ach = prof[dom]["work"] / (dom_ms * 1e-3) / 1e9
            roof = {"kernel": dom, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None}
        else:
            ach = prof[dom]["work"] / (dom_ms * 1e-3) / 1e12
            roof = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_FP64_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / PEAK_FP64_TFLOPS, "traffic": None}
        dom_kernel = {"syrk_trailing_small": "slab_step_kernel<false, 8>",
                      "syrk_trailing": TRAILING_KERNEL}.get(dom)
        if dom_kernel:
            roof["traffic"], roof["traffic_source"] = pmc_traffic(dom_kernel)
            roof["kernel_symbol"] = dom_kernel
        # The instrumented pass brackets every launch with HIP events, which stretches a chain of
        # short launches beyond the step it belongs to.  The class time behind `achieved` is
        # therefore the class's SHARE of the instrumented pass applied to the un-instrumented
        # step (HIP events around the whole timed region): never more than ms_per_step.
        ev_step = res["ev_ms"] / a.steps
        tot_instr = sum(v["ms"] for v in prof.values())
        dom_ms_step = dom_ms * min(1.0, ev_step / tot_instr) if tot_instr > 0 else dom_ms
        scale = dom_ms / dom_ms_step if dom_ms_step > 0 else 1.0
        roof["achieved"] *= scale
        roof["frac"] *= scale
        roof.update({"launches_per_step": prof[dom]["launches"],
                     "ms_per_launch": dom_ms_step / max(1, prof[dom]["launches"]),
                     "ms_per_launch_instrumented": per_launch_ms,
                     "ms_per_step_in_class": dom_ms_step,
                     "ms_per_step_in_class_instrumented": dom_ms,
                     "time_basis": "class share of the instrumented pass x the un-instrumented "
                                   "step (HIP events over the timed region)",
                     "algorithmic_work_per_step": prof[dom]["work"],
                     "class_ms_per_step": {k: v["ms"] for k, v in prof.items()},
                     "class_launches_per_step": {k: v["launches"] for k, v in prof.items()},
                     "note": "dominant kernel class of the timed workload (HIP events on the "
                             "engine stream, instrumented pass; the events' own overhead makes "
                             "the class sum exceed ms_per_step).  A single C2 problem is a "
                             "chain of 16 dependent one-launch steps, each bound by ONE "
                             "workgroup's critical path (panel-row solve, tile update, the "
                             "64-pivot diagonal factor: tools/c2_timeline.py), not by the "
                             "chip's MFMA rate; the kernels with a roofline target are under "
                             "'rooflines'"})
        line = {
            "metric": "bq_fit_posterior_throughput",
            "value": value,
            "unit": "problems/s",
            "n_gpus": dist.world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": res["wall"] / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wk["desc"],
                       "value_is": "problems/s of the whole job on: " + wk["desc"],
                       # the workload the 1 -> N curve is on: at N = 1 the curve's point rides
                       # beside the headline (scale_point); an N > 1 line IS a curve point only
                       # when it runs that workload (`--gpus N --workload c2` used to
                       # claim the C5 description for a C2 measurement)
                       "curve_workload": (make_workload_desc(CURVE_WORKLOAD, a.curve_batch)
                                          if dist.world == 1 else wk["desc"]),
                       "on_curve": dist.world == 1 or a.workload == CURVE_WORKLOAD,
                       "curve_note": "the 1 -> N scaling curve is on curve_workload: N = 1 reads "
                                     "scale_point.value of the --gpus 1 line (whose `value` is "
                                     "config.workload), N > 1 reads `value`; every N > 1 line "
                                     "repeats the N = 1 point as n1_same_workload; a line with "
                                     "on_curve false ran another workload and is not a curve point",
                       "problems_per_gpu_per_step": wk["B"],
                       "bordered_system": ntot, "sharding": "independent problems per rank, "
                       "no data-path collective",
                       "ranks": res["ranks"]},
            "host_issue": res["ranks_issue"],
            "ms_per_problem": res["wall"] / a.steps / wk["B"] * 1e3,
            "ms_per_step_hip_events": res["ev_ms"] / a.steps,
            "ms_per_step_with_readback": res["rb_ms"],
            "failed_problems": int(nfail),
            "device": info,
            "plan_bytes": res["plan_bytes"],
            "roofline": roof,
        }
        curve_fields(line, wk, res["solo_ms"], dist.world)
        line["parity"] = parity_spotcheck(wk, res)
        if dist.world == 1:
            # the N = 1 point of the C5 curve, whatever the headline workload is
            if a.workload == CURVE_WORKLOAD:
                line["scale_point"] = {
                    "workload": wk["desc"], "value": value, "unit": "problems/s",
                    "ms_per_step": res["wall"] / a.steps * 1e3, "n_gpus": 1,
                    "problems_per_gpu_per_step": wk["B"], "failed": int(nfail),
                    "note": "this line's own value: the headline workload is the curve workload"}
            else:
                line["scale_point"] = scale_point(eng, a.curve_batch)
            line["ms_per_step_host_buffers"] = host_buffer_step(eng, wk)
            line["ms_per_step_host_buffers_note"] = (
                "one problem of config.workload through bq_fit_predict with host buffers: plan "
                "creation + H2D + run + D2H + synchronisation per call (what a BQ user's call "
                "costs); `value` / ms_per_step have the inputs resident")
            try:
                line["probes"] = {"mfma_f64_tflops": eng.probe_mfma_f64(),
                                  "mfma_f64_4x4x4_4b_tflops": eng.probe_mfma_variant(1, 8, 2),
                                  "fma_f64_tflops": eng.probe_fma_f64()}
                wgb, cgb = eng.probe_hbm(1 << 30)
                line["probes"].update({"hbm_write_gbs": wgb, "hbm_copy_gbs": cgb})
                # the size of the N = 4096 Gram (134 MB, inside the 256 MiB Infinity Cache):
                # what a kernel that only stores reaches on this box
                wgb2, cgb2 = eng.probe_hbm(1 << 27)
                line["probes"].update({"hbm_write_gbs_128MiB": wgb2, "hbm_copy_gbs_128MiB": cgb2})
            except Exception as e:  # probes are informational
                line["probes"] = {"error": str(e)}
            if not a.no_extras:
                line["rooflines"] = extras(eng, a.nb)
                line["rooflines"]["fit_posterior_ms_at_n"] = fit_posterior_at_n(eng)
                line["roofline"] = northstar_first(line["roofline"], line["rooflines"])
            if not a.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(wk)
                line["cpu_baseline"]["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        p = line["parity"]
        if not (p["mean_rel"] < 1e-10 and p["var_rel_prior"] < 1e-10 and p["logml_rel"] < 1e-10) \
                or nfail:
            line["value"] = None
            line["invalid"] = "parity check failed"
    dist.barrier()
    if dist.rank == 0:
        print(json.dumps(line))
    eng.close()
    dist.close()


if __name__ == "__main__":
    main()
