"""The N > 1 path: independent problems block-partitioned over ranks, no data-path
collective, results merged in problem order.  Exercised with world_size 2 on the CPU
(gloo) against the engine double; the device work per rank is what
tests/test_gpu_parity.py::test_batch_fit_predict checks on the GPU."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_shard_partition():
    from bayesian_quadrature_amd.workloads import shard
    for n in (0, 1, 7, 8, 512, 513):
        for world in (1, 2, 3, 8):
            blocks = [shard(n, r, world) for r in range(world)]
            assert sorted(sum(blocks, [])) == list(range(n))
            sizes = [len(b) for b in blocks]
            assert max(sizes) - min(sizes) <= 1
            assert all(b == list(range(b[0], b[0] + len(b))) for b in blocks if b)


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "LOCAL_RANK": str(rank)})
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import torch.distributed as td
    td.init_process_group(backend="gloo", rank=rank, world_size=world)
    from oracle import load
    from engine_double import EngineDouble
    from bayesian_quadrature_amd import shard as sh
    from bayesian_quadrature_amd import workloads as wl
    eng = EngineDouble(load())
    c = wl.c5(list(range(5)), n=60, m=7)
    idx, mean, var, logml, status = sh.batch_fit_predict_sharded(
        eng, c["x"], c["y"], c["xo"], c["h"], c["w"] * 10, c["s"])
    all_idx, (gmean, gvar, glogml) = sh.gather(idx, [mean, var, logml])
    h = np.array([0.5, 1.0, 2.0])
    gidx, lm = sh.logml_grid_sharded(eng, c["x"][0], c["y"][0], h, np.full(3, 1.7), 0.01)
    gall, (glm,) = sh.gather(gidx, [lm])
    td.barrier()
    q.put((rank, idx, all_idx, gmean, gvar, glogml, gall, glm))
    td.destroy_process_group()


def test_two_rank_gloo_matches_single_process(oracle):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # blocks are disjoint, contiguous, and cover 0..4
    assert outs[0][1] == [0, 1, 2] and outs[1][1] == [3, 4]
    from engine_double import EngineDouble
    from bayesian_quadrature_amd import workloads as wl
    c = wl.c5(list(range(5)), n=60, m=7)
    mean, var, logml, _ = EngineDouble(oracle).batch_fit_predict(
        c["x"], c["y"], c["h"], c["w"] * 10, c["s"], c["xo"])
    ref_lm = EngineDouble(oracle).logml_grid(c["x"][0], c["y"][0], np.array([0.5, 1.0, 2.0]),
                                             np.full(3, 1.7), 0.01)
    for rank, idx, all_idx, gmean, gvar, glogml, gall, glm in outs:
        assert all_idx == [0, 1, 2, 3, 4] and gall == [0, 1, 2]
        assert (gmean == mean).all() and (gvar == var).all() and (glogml == logml).all()
        assert (glm == ref_lm).all()


# ---- the same path on the real engine: two processes, two HIP contexts ------------------
def _gpu_worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "LOCAL_RANK": str(rank)})
    sys.path.insert(0, ROOT)
    from bayesian_quadrature_amd import Engine      # the HIP context first, torch after
    eng = Engine(0)                                  # a one-GPU box: both ranks on device 0
    import torch.distributed as td
    td.init_process_group(backend="gloo", rank=rank, world_size=world)
    from bayesian_quadrature_amd import shard as sh
    from bayesian_quadrature_amd import workloads as wl
    c = wl.c5(list(range(7)), n=300, m=40)
    idx, mean, var, logml, status = sh.batch_fit_predict_sharded(
        eng, c["x"], c["y"], c["xo"], c["h"], c["w"] * 10, c["s"])
    all_idx, (gmean, gvar, glogml, gstatus) = sh.gather(idx, [mean, var, logml, status])
    td.barrier()
    q.put((rank, idx, all_idx, gmean, gvar, glogml, gstatus))
    td.destroy_process_group()
    eng.close()


@pytest.mark.gpu
def test_two_ranks_real_engines_match_single_process(engine, oracle):
    """world_size 2, one process and one HIP context per rank (both on device 0 of the
    one-GPU test box), block partition, gloo gather: the merged result equals what one
    process computes for all problems, and the oracle's."""
    import torch.multiprocessing as mp
    from bayesian_quadrature_amd import workloads as wl
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert outs[0][1] == [0, 1, 2, 3] and outs[1][1] == [4, 5, 6]
    c = wl.c5(list(range(7)), n=300, m=40)
    mean, var, logml, status = engine.batch_fit_predict(c["x"], c["y"], c["h"], c["w"] * 10,
                                                        c["s"], c["xo"])
    k0 = oracle.kernel_scale(1, c["h"], c["w"] * 10)
    for rank, idx, all_idx, gmean, gvar, glogml, gstatus in outs:
        assert all_idx == list(range(7)) and (gstatus == 0).all()
        assert np.max(np.abs(gmean - mean)) <= 1e-10 * np.max(np.abs(mean))
        assert np.max(np.abs(gvar - var)) <= 1e-10 * k0
        assert np.max(np.abs(glogml - logml) / np.abs(logml)) <= 1e-10
    for i in (0, 6):
        Lo, ao, lmo = oracle.gp_fit(c["x"][i], c["y"][i], c["h"], c["w"] * 10, c["s"])
        assert abs(outs[0][5][i] - lmo) <= 1e-10 * abs(lmo)


def _run_bench(args, env_extra):
    import subprocess
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env,
                          capture_output=True, text=True, timeout=900)


def _line_of(r):
    import json
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_lines_carry_the_same_curve_workload():
    """The 1 -> N scaling curve can be computed from bench.py's own lines:
    ``--gpus 1`` (headline C2) carries ``scale_point`` on the C5 shard, ``--gpus 2`` (two real
    ranks over gloo) carries the same workload as ``value`` and repeats the N = 1 point as
    ``n1_same_workload``; both name the workload under the same key.  On the CPU through
    tests/bench_double_main.py (an oracle-backed double; the numbers mean nothing here)."""
    import subprocess
    launcher = os.path.join(ROOT, "tests", "bench_double_main.py")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)

    def run(args):
        return subprocess.run([sys.executable, launcher] + args, env=env, capture_output=True,
                              text=True, timeout=600)

    common = ["--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    one = _line_of(run(["--gpus", "1", "--curve-batch", "2"] + common))
    two = _line_of(run(["--gpus", "2", "--batch", "2"] + common))
    # N = 1: the headline stays C2 (BASELINE configs[1]); the curve's point is beside it
    assert one["n_gpus"] == 1 and one["config"]["workload"].startswith("C2")
    sp = one["scale_point"]
    assert sp["workload"].startswith("C5 shard: 2 x") and sp["value"] > 0 and sp["n_gpus"] == 1
    assert one["config"]["curve_workload"] == sp["workload"]
    assert one["ms_per_step_host_buffers"] > 0
    assert "n1_same_workload" not in one and "scaling_efficiency" not in one
    # N = 2: value is the curve workload, the N = 1 point measured in the same run
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["config"]["workload"] == two["config"]["curve_workload"] == sp["workload"]
    n1 = two["n1_same_workload"]
    assert n1["workload"] == sp["workload"] and n1["value"] > 0 and n1["n_gpus"] == 1
    assert abs(two["scaling_efficiency"] - two["value"] / (2 * n1["value"])) < 1e-12
    assert sorted(r_["rank"] for r_ in two["config"]["ranks"]) == [0, 1]
    assert len({r_["pid"] for r_ in two["config"]["ranks"]}) == 2
    for line in (one, two):
        assert line["value"] is not None and line["failed_problems"] == 0
        assert line["parity"]["logml_rel"] < 1e-10 and "unpinned" in line["parity"]["noise_form"]
        assert line["config"]["value_is"].endswith(line["config"]["workload"])


def test_north_star_scalars_lead_the_roofline_object():
    """The driver's record keeps about the first twenty keys of `roofline` and cuts names at 40
    characters (BENCH_r05.json lost the Gram fractions behind three long keys): the six north-star
    scalars are among the first 16 keys, at most 32 characters each, and equal their sources."""
    import bench
    rl = {"trailing_update_n16384": {"frac": 0.74, "whole_trailing_update": {"frac": 0.69}},
          "trailing_update_n16384_dense": {"frac": 0.70, "whole_trailing_update": {"frac": 0.66}},
          "gram_n4096_d2": {"frac": 0.84}, "gram_n16384_d1": {"frac": 0.67},
          "c5_shard_64x2048": {"frac": 0.57, "ms_per_batch": 5.6},
          "c3_grid_400x4096": {"frac": 0.64, "wall_ms": 181.0},
          "c2_batch_256x1024": {"frac": 0.48, "ms_per_batch": 4.2},
          "potrf_n16384": {"ms": 27.2},
          "fit_posterior_ms_at_n": {"n1024": {"ms": 0.25}, "n16384": {"ms": 28.0}}}
    roof = {"kernel": "syrk_trailing_small", "bound": "mfma", "achieved": 3.4, "peak": 78.6,
            "unit": "TFLOP/s", "frac": 0.043, "traffic": None, "note": "x" * 300,
            "time_basis": "y" * 100, "class_ms_per_step": {"a": 1.0}}
    out = bench.northstar_first(roof, rl)
    keys = list(out)
    for k in bench.NORTHSTAR_KEYS:
        assert k in keys[:16] and len(k) <= 32, k
    assert keys[:7] == ["kernel", "bound", "achieved", "peak", "unit", "frac", "traffic"]
    assert out["trail16k_frac"] == 0.69 and out["trail16k_dense_frac"] == 0.66
    assert out["gram4096_frac"] == 0.84 and out["gram16k_frac"] == 0.67
    assert out["c5_frac"] == 0.57 and out["c3_frac"] == 0.64
    assert out["fitpost_n1024_ms"] == 0.25 and out["fitpost_n16384_ms"] == 28.0
    assert keys.index("note") > keys.index("c2x256_ms") and out["note"] == roof["note"]
    assert all(len(k) <= 32 for k in keys[:24])
    assert set(bench.NORTHSTAR_KEYS) <= set(rl["northstar_source"])


@pytest.mark.gpu
def test_bench_launches_its_own_ranks(engine):
    """``bench.py --gpus 2`` started bare spawns two ranks itself.  On a one-GPU box that
    must fail loudly; with BQ_BENCH_SHARE_DEVICE=1 (rehearsal only) the two ranks share the
    device and the line reports n_gpus 2 with two distinct ranks / processes."""
    import json
    ndev = engine.device_count()
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-extras",
            "--no-cpu-baseline"]
    if ndev < 2:
        r = _run_bench(args, {})
        assert r.returncode != 0
        assert "needs 2 HIP devices" in r.stderr
    r = _run_bench(args, {"BQ_BENCH_SHARE_DEVICE": "1"} if ndev < 2 else {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] is not None and line["failed_problems"] == 0
    assert line["config"]["workload"].startswith("C5")
    ranks = line["config"]["ranks"]
    assert sorted(r_["rank"] for r_ in ranks) == [0, 1]
    assert len({r_["pid"] for r_ in ranks}) == 2
    assert line["n1_same_workload"]["ms_per_step"] > 0
    assert line["n1_same_workload"]["workload"] == line["config"]["workload"]
    assert 0 < line["scaling_efficiency"] < 1.5
    assert line["parity"]["logml_rel"] < 1e-10


def test_engine_pool_partitions_and_propagates_errors(oracle):
    """EnginePool's host logic on the CPU: contiguous blocks per worker, results in item
    order, exceptions re-raised -- with oracle-backed doubles in place of device engines."""
    from bayesian_quadrature_amd import pool as pool_mod
    from bayesian_quadrature_amd import workloads as wl
    from engine_double import EngineDouble

    class FakeEngine(EngineDouble):
        def __init__(self, device):
            EngineDouble.__init__(self, oracle)
            self.device = device
            self.seen = []

        def logml_grid(self, x, y, h, w, s=0.0, chunk=0):
            self.seen.append(len(h))
            return EngineDouble.logml_grid(self, x, y, h, w, s, chunk)

        def close(self):
            pass

    real = pool_mod.Engine
    pool_mod.Engine = FakeEngine
    try:
        with pool_mod.EnginePool([0, 0, 0]) as pool:
            assert len(pool) == 3
            c = wl.c5([0], n=40, m=5)
            h = np.linspace(0.5, 2.0, 7)
            w = np.full(7, float(c["w"][0]) * 10)
            lm = pool.logml_grid(c["x"][0], c["y"][0], h, w, 0.01)
            ref = EngineDouble(oracle).logml_grid(c["x"][0], c["y"][0], h, w[:, None], 0.01)
            assert np.array_equal(lm, ref)
            assert sorted(n for wk in pool._workers for n in wk.engine.seen) == [2, 2, 3]
            c5 = wl.c5(list(range(4)), n=40, m=5)
            got = pool.batch_fit_predict(c5["x"], c5["y"], c5["h"], c5["w"] * 10, c5["s"], c5["xo"])
            one = EngineDouble(oracle).batch_fit_predict(c5["x"], c5["y"], c5["h"], c5["w"] * 10,
                                                         c5["s"], c5["xo"])
            assert all(np.array_equal(a, b) for a, b in zip(got, one))
            with pytest.raises(ZeroDivisionError):
                pool.run([lambda e: 1, lambda e: 1 // 0, lambda e: 3])
            assert pool.run([lambda e: e.device] * 3) == [0, 0, 0]   # still alive afterwards
            # a rank that fails before a barrier breaks it for the others instead of leaving them
            # waiting (bench.py --inproc): everybody returns, the REAL failure is
            # the one that is re-raised
            import threading
            bar = threading.Barrier(3)

            def rank(i):
                def run(e):
                    try:
                        if i == 1:
                            raise MemoryError("plan allocation failed on this device")
                        bar.wait(timeout=30)
                        return i
                    except BaseException:
                        bar.abort()
                        raise
                return run
            with pytest.raises(MemoryError):
                pool.run([rank(0), rank(1), rank(2)])
            assert pool.run([lambda e: e.device] * 3) == [0, 0, 0]
        # a constructor that fails half way closes the workers it has started
        started = []

        class HalfEngine(FakeEngine):
            def __init__(self, device):
                if device == 7:
                    raise RuntimeError("no such device")
                super().__init__(device)
                started.append(self)

            def close(self):
                started.remove(self)
        pool_mod.Engine = HalfEngine
        with pytest.raises(RuntimeError):
            pool_mod.EnginePool([0, 0, 7])
        assert started == []
    finally:
        pool_mod.Engine = real


@pytest.mark.gpu
def test_engine_pool_two_contexts_one_process(engine, oracle):
    """One process, an engine + host thread per device (here: two contexts on device 0, the
    one-GPU box's rehearsal of SURVEY 8e's "one Python thread per device"): a log-ML grid, a
    batch of problems and an acquisition sweep partitioned over the pool equal what a single
    context computes."""
    from bayesian_quadrature_amd import EnginePool
    from bayesian_quadrature_amd import workloads as wl
    ndev = engine.device_count()
    with EnginePool([0, 1 % ndev]) as pool:
        c = wl.c5(list(range(7)), n=300, m=40)
        got = pool.batch_fit_predict(c["x"], c["y"], c["h"], c["w"] * 10, c["s"], c["xo"])
        one = engine.batch_fit_predict(c["x"], c["y"], c["h"], c["w"] * 10, c["s"], c["xo"])
        assert (got[3] == 0).all()
        for a, b in zip(got[:3], one[:3]):
            assert np.max(np.abs(a - b)) <= 1e-12 * np.max(np.abs(b))
        c3 = wl.c3(side=16, gh=4, gw=3)
        lm = pool.logml_grid(c3["x"], c3["y"], c3["h"], c3["w"], c3["s"])
        lm1 = engine.logml_grid(c3["x"], c3["y"], c3["h"], c3["w"], c3["s"])
        assert np.max(np.abs(lm - lm1) / np.abs(lm1)) <= 1e-12
        xs = np.linspace(-5, 5, 40)
        x_sc = np.concatenate([xs, [-5.6, 5.7]])
        l_sc = np.exp(wl.norm_logpdf(x_sc))
        x_a = np.linspace(-7, 7, 11) + 0.013
        mu, cov = np.array([0.0]), np.array([[10.0]])
        a = pool.esm_batch(x_sc, l_sc, 40, x_a, 0.2, 0.3, 0.5, mu, cov)
        b = engine.esm_batch(x_sc, l_sc, 40, x_a, 0.2, 0.3, 0.5, mu, cov)
        assert (a[2] == b[2]).all()
        assert np.allclose(a[0], b[0], rtol=1e-12, atol=0) and np.allclose(a[1], b[1], rtol=1e-12, atol=0)
        # the two workers really ran concurrently-capable: distinct contexts, distinct threads
        import threading
        names = pool.run([lambda e: threading.current_thread().name] * 2)
        assert len(set(names)) == 2


@pytest.mark.gpu
def test_bench_inproc_mode(engine):
    """``bench.py --gpus 2 --inproc``: the parent creates both contexts itself and spawns
    nothing; loud failure without a second device, shared-device rehearsal otherwise."""
    import json
    ndev = engine.device_count()
    args = ["--gpus", "2", "--inproc", "--steps", "2", "--warmup", "1", "--batch", "2"]
    if ndev < 2:
        r = _run_bench(args, {})
        assert r.returncode != 0 and "needs 2 HIP devices" in r.stderr
    r = _run_bench(args, {"BQ_BENCH_SHARE_DEVICE": "1"} if ndev < 2 else {})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["value"] is not None and line["failed_problems"] == 0
    assert line["config"]["launch_mode"].startswith("inproc")
    assert line["n1_same_workload"]["workload"] == line["config"]["workload"]
    assert sorted(r_["rank"] for r_ in line["config"]["ranks"]) == [0, 1]
    assert line["parity"]["logml_rel"] < 1e-10


@pytest.mark.gpu
def test_eight_rank_host_side_rehearsal(engine):
    """The host side of an eight-GPU node on the one-GPU box (no curve can be measured here):
    ``bench.py --gpus 8 --inproc`` -- ONE process, eight contexts, eight host threads, 8 problems per
    rank -- and five ranks as five processes (the box admits six processes on its card, this test
    session is one of them), all on device 0 under BQ_BENCH_SHARE_DEVICE=1.  Every rank's host issue time per step stays below
    the device time of the same steps: with the plan's launches replayed from a hipGraph a step
    costs the host one call, so the launch rate is not what limits eight contexts."""
    import json
    for args in (["--gpus", "8", "--inproc"], ["--gpus", "5", "--no-extras", "--no-cpu-baseline"]):
        r = _run_bench(args + ["--steps", "4", "--warmup", "2", "--batch", "8"],
                       {"BQ_BENCH_SHARE_DEVICE": "1"})
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
        assert line["n_gpus"] == int(args[1]) and line["value"] is not None
        assert line["failed_problems"] == 0
        hi = line["host_issue"]
        assert len(hi) == int(args[1])
        for rk in hi:
            assert rk["host_issue_ms_per_step"] < rk["device_ms_per_step"], (args, rk)
