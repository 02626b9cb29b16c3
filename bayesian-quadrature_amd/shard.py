"""Sharding of independent BQ problems / hyper-parameter points over GPUs.

The path has no exchange step: rank r owns a contiguous block of
the problem list, runs it on its own device context and the host concatenates a few
doubles per problem.  No RCCL call exists on the data path; ``gather`` below moves only
the tiny result vectors, through whatever ``torch.distributed`` backend the launcher
initialised (gloo in the tests and in bench.py), or does nothing in a single process.
"""
import os

import numpy as np

from .workloads import shard


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def my_block(nitems, rank=None, world=None):
    """Indices of the items this rank owns."""
    if rank is None or world is None:
        rank, world = rank_world()
    return shard(nitems, rank, world)


def batch_fit_predict_sharded(eng, x, y, xo, h, w, s, rank=None, world=None):
    """Run this rank's block of the P problems (x: (P, n) or (P, d, n), ...).
    Returns (indices, mean, var, logml, status) for the block."""
    idx = my_block(len(x), rank, world)
    if not idx:
        M = np.asarray(xo).shape[-1]
        return idx, np.empty((0, M)), np.empty((0, M)), np.empty(0), np.empty(0, dtype=np.int32)
    sel = np.asarray(idx)
    mean, var, logml, status = eng.batch_fit_predict(
        np.asarray(x)[sel], np.asarray(y)[sel], h, w, s, np.asarray(xo)[sel])
    return idx, mean, var, logml, status


def logml_grid_sharded(eng, x, y, h, w, s, rank=None, world=None):
    """This rank's block of the G hyper-parameter points of a log-ML grid."""
    h = np.asarray(h, dtype=np.float64).ravel()
    w = np.asarray(w, dtype=np.float64).reshape(h.shape[0], -1)
    idx = my_block(h.shape[0], rank, world)
    if not idx:
        return idx, np.empty(0)
    sel = np.asarray(idx)
    return idx, eng.logml_grid(x, y, h[sel], w[sel], s)


def gather(idx, arrays):
    """All ranks' (idx, arrays) merged in problem order on every rank."""
    try:
        import torch.distributed as td
        live = td.is_available() and td.is_initialized()
    except Exception:
        live = False
    if not live:
        return list(idx), [np.asarray(a) for a in arrays]
    parts = [None] * td.get_world_size()
    td.all_gather_object(parts, (list(idx), [np.asarray(a) for a in arrays]))
    order = np.argsort(np.concatenate([np.asarray(p[0], dtype=np.int64) for p in parts]))
    merged = []
    for k in range(len(arrays)):
        merged.append(np.concatenate([p[1][k] for p in parts], axis=0)[order])
    return sorted(i for p in parts for i in p[0]), merged
