"""Launcher of ``bench.py`` on a machine WITHOUT a GPU -- test infrastructure only.

Runs bench.main() with the device engine replaced by an oracle-backed double that has the
shape bench.py drives (plan / timers / launch profiler), so that the CPU suite can check the
JSON lines bench.py prints at --gpus 1 and --gpus 2 (tests/test_sharding.py).  The numbers in
those lines are meaningless (a "pass" is a sleep); the keys, the rank plumbing over gloo and
the parity block are what the test reads.  Nothing in the product or in bench.py imports this
file; bench.py's own children re-run it because it names itself in BQ_BENCH_LAUNCHER.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402
import bayesian_quadrature_amd as pkg  # noqa: E402
from engine_double import EngineDouble  # noqa: E402
from oracle import load  # noqa: E402

PASS_S = 0.002


class PlanDouble(object):
    def __init__(self, eng, nprob, d, n, M):
        self.eng, self.shape = eng, (nprob, d, n, M)
        self.inputs = self.out = None

    def set_inputs(self, x, y, xo, h, w, s):
        self.inputs, self.out = (x, y, h, w, s, xo), None

    def run(self):
        time.sleep(PASS_S)
        self.eng.passes += 1

    def results(self):
        if self.out is None:   # one evaluation per input set: the passes themselves are sleeps
            self.out = self.eng.double.batch_fit_predict(*self.inputs)
        return self.out

    def nbytes(self):
        P, d, n, M = self.shape
        return 8 * P * (n + M + 64) ** 2

    def close(self):
        pass


class BenchEngineDouble(object):
    def __init__(self, device=0):
        self.device, self.passes, self._t = int(device), 0, 0.0
        o = load()
        o.set_threads(2)
        self.double = EngineDouble(o)

    def plan(self, nprob, d, n, M):
        return PlanDouble(self, nprob, d, n, M)

    def fit_predict(self, x, y, h, w, s, xo):
        m, v, lm, _ = self.double.batch_fit_predict([x], [y], h, w, s, [xo])
        return m[0], v[0], lm[0]

    def sync(self):
        pass

    def timer_start(self):
        self._t = time.perf_counter()

    def timer_stop_ms(self):
        return (time.perf_counter() - self._t) * 1e3

    def profile(self, on):
        self._p0 = self.passes

    def profile_reset(self):
        self._p0 = self.passes

    def profile_read(self):
        k = max(1, self.passes - self._p0)
        return {"syrk_trailing": {"ms": PASS_S * 1e3 * k, "work": 1e9 * k, "launches": 4 * k},
                "gram": {"ms": 0.1 * k, "work": 1e6 * k, "launches": k}}

    def info(self):
        return {"name": "engine double (no device)", "device": self.device}

    def set_block(self, nb):
        pass

    def close(self):
        pass


if __name__ == "__main__":
    os.environ["BQ_BENCH_LAUNCHER"] = os.path.abspath(__file__)
    pkg.Engine = BenchEngineDouble
    bench.device_count = lambda: int(os.environ.get("BQ_TEST_FAKE_DEVICES", "8"))
    bench.main()
