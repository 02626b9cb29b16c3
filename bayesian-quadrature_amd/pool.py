"""One process, all devices: an engine per GPU, each driven by a host thread of its own.

The path shards with no exchange ("one Python thread per device or a
single thread issuing async launches to 8 contexts"): the hyper-parameter points of a log-ML
grid, the problems of a batch, the candidates of an acquisition sweep are cut into contiguous
blocks (``workloads.shard``), every block runs on its own device context from its own thread --
ctypes releases the GIL for the duration of a library call, and a context is only ever touched
by the one thread that owns it --, and the host concatenates a few doubles per item.  No
``torch.distributed``, no collective.  ``shard.py`` is the same partition for the
process-per-GPU launch mode.
"""
import threading

import numpy as np

from . import _lib as L
from .engine import Engine
from .workloads import shard


class _Worker(threading.Thread):
    """The one thread that ever calls into its engine's context."""

    def __init__(self, device):
        super().__init__(daemon=True)
        self.device = device
        self.engine = None
        self._jobs = []
        self._cv = threading.Condition()
        self._ready = threading.Event()
        self._error = None
        self.start()
        self._ready.wait()
        if self._error is not None:
            raise self._error

    def run(self):
        try:
            self.engine = Engine(self.device)
        except BaseException as e:  # reported to the creating thread
            self._error = e
            self._ready.set()
            return
        self._ready.set()
        while True:
            with self._cv:
                while not self._jobs:
                    self._cv.wait()
                fn, box, done = self._jobs.pop(0)
            if fn is None:
                self.engine.close()
                done.set()
                return
            try:
                box.append((True, fn(self.engine)))
            except BaseException as e:
                box.append((False, e))
            done.set()

    def submit(self, fn):
        box, done = [], threading.Event()
        with self._cv:
            self._jobs.append((fn, box, done))
            self._cv.notify()
        return box, done


class EnginePool(object):
    """Engines on ``devices`` (default: every visible device; an index may repeat -- two
    contexts on one device -- which is how the one-GPU test box exercises the pool)."""

    def __init__(self, devices=None):
        if devices is None:
            import ctypes as C
            n = C.c_int(0)
            L.load_library().bq_device_count(C.byref(n))
            if n.value <= 0:
                raise RuntimeError("no HIP device visible: the MI355X engine cannot run "
                                   "(there is no CPU fallback)")
            devices = list(range(n.value))
        self.devices = [int(d) for d in devices]
        self._workers = []
        try:
            for d in self.devices:
                self._workers.append(_Worker(d))
        except BaseException:
            self.close()  # the threads and contexts already started
            raise

    def __len__(self):
        return len(self._workers)

    def close(self):
        for w in self._workers:
            _, done = w.submit(None)
            done.wait()
        self._workers = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- plumbing ------------------------------------------------------------------------
    def run(self, fns):
        """fns[i](engine_i) on worker i, concurrently; the results in order.  The first
        exception (in worker order) is re-raised after every worker has finished."""
        if len(fns) != len(self._workers):
            raise ValueError("one function per engine")
        pend = [w.submit(fn) for w, fn in zip(self._workers, fns)]
        out = []
        for box, done in pend:
            done.wait()
            out.append(box[0])
        # the first REAL failure in worker order: a worker that only saw a barrier broken by
        # another one's failure (bench.py --inproc) is reported only if nothing else failed
        bad = [val for ok, val in out if not ok]
        if bad:
            real = [e for e in bad if not isinstance(e, threading.BrokenBarrierError)]
            raise (real or bad)[0]
        return [val for _, val in out]

    def map_blocks(self, nitems, fn):
        """fn(engine, indices) for every engine's contiguous block of range(nitems); returns
        [(indices, result)] for the non-empty blocks, in item order."""
        world = len(self._workers)
        blocks = [shard(nitems, r, world) for r in range(world)]

        def job(idx):
            return (lambda eng: fn(eng, idx)) if idx else (lambda eng: None)

        res = self.run([job(idx) for idx in blocks])
        return [(idx, r) for idx, r in zip(blocks, res) if idx]

    # -- the sharded entry points ----------------------------------------------------------
    def logml_grid(self, x, y, h, w, s=0.0, chunk=0):
        """Engine.logml_grid with the G hyper-parameter points block-partitioned over the
        devices (BASELINE config 3 on a whole node)."""
        h = np.ascontiguousarray(h, dtype=np.float64).ravel()
        G = h.shape[0]
        w = np.ascontiguousarray(np.asarray(w, dtype=np.float64).reshape(G, -1))
        parts = self.map_blocks(G, lambda eng, idx: eng.logml_grid(x, y, h[idx], w[idx], s, chunk))
        return np.concatenate([r for _, r in parts]) if parts else np.empty(0)

    def batch_fit_predict(self, x, y, h, w, s, xo):
        """Engine.batch_fit_predict with the P problems block-partitioned over the devices
        (BASELINE config 5): (mean, var, logml, status)."""
        x, y, xo = np.asarray(x), np.asarray(y), np.asarray(xo)
        parts = self.map_blocks(
            len(x), lambda eng, idx: eng.batch_fit_predict(x[idx], y[idx], h, w, s, xo[idx]))
        return tuple(np.concatenate([r[k] for _, r in parts], axis=0) for k in range(4))

    def esm_batch(self, x_sc, l_sc, ns, x_a, h, w, thresh, mu, cov):
        """Engine.esm_batch with the candidates x_a block-partitioned over the devices
        (bq.py:399-402: the acquisition loop is embarrassingly parallel over x_a)."""
        x_a = np.ascontiguousarray(x_a, dtype=np.float64)
        parts = self.map_blocks(
            x_a.shape[0],
            lambda eng, idx: eng.esm_batch(x_sc, l_sc, ns, x_a[idx], h, w, thresh, mu, cov))
        return tuple(np.concatenate([r[k] for _, r in parts]) for k in range(3))
