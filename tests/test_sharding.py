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
