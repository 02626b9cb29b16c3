"""Socket power and shader clock (hwmon) WHILE a kernel runs for seconds: the fp64 MFMA loop
without memory, and the LDS-staged trailing-update kernel on random operands -- the evidence for
DESIGN.md's "the product is power-bound" paragraph.  BQ_GEMM_TILE=64|128 python tools/power_probe.py"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402

e = Engine(0)


def smi():
    """(sclk MHz, socket watts) from the card's hwmon node (bench.py's sampler: plain sysfs reads)"""
    import bench
    v = bench._hwmon_sample()
    return (int(v[1]), v[0]) if v else (-1, -1.0)


def watch(name, fn, rate):
    res = {}
    t = threading.Thread(target=lambda: res.setdefault("v", fn()))
    t.start()
    time.sleep(1.0)
    seen = []
    while t.is_alive():
        seen.append(smi())
        time.sleep(0.4)
    t.join()
    seen = seen[:-1] or seen
    clk = sum(s[0] for s in seen) / max(len(seen), 1)
    pw = sum(s[1] for s in seen) / max(len(seen), 1)
    r = rate(res["v"])
    print("%-46s %6.1f TFLOP/s  sclk %4.0f MHz  %5.0f W  -> %.1f%% of the MFMA rate at that clock"
          % (name, r, clk, pw, 100.0 * r / (78.6 * clk / 2400.0)), flush=True)


print("tile forced to", os.environ.get("BQ_GEMM_TILE", "(gemm_lds_tile's choice)"), " idle:", smi())
watch("MFMA loop, random mantissas, 2 waves/SIMD", lambda: [e.probe_mfma_variant(5, 8, 2) for _ in range(10)],
      lambda v: sum(v) / len(v))
for m, k, b, reps in ((16064, 640, 1, 1200), (8192, 640, 1, 4000), (2816, 320, 100, 600), (2048, 320, 32, 4000)):
    watch("trailing update m=%d k=%d batch=%d" % (m, k, b),
          lambda: e.probe_gemm(m, m, k, 1, b, False, reps),
          lambda ms: float(m) * m * k * b / ms / 1e9)
e.close()
