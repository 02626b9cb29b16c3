"""The evidence behind gemm_lds_tile (k_gemm.hip): TFLOP/s of C -= P Q^T on RANDOM operands,
each shape launched back to back for >= 100 ms, through the engine's own tile choice and with
the LDS kernel's tile forced to 64 / 128 (BQ_GEMM_TILE, read when a context is created).
Lower = 1: the lower triangle of a square trailing update (flops m^2 k); else the full product.
python tools/gemm_probe.py [m n k lower batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402

shapes = [(16064, 16064, 640, 1, 1), (8192, 8192, 640, 1, 1), (6144, 6144, 640, 1, 1),
          (5120, 5120, 640, 1, 1), (4096, 4096, 640, 1, 1), (3072, 3072, 320, 1, 1),
          (2816, 2816, 320, 1, 100), (1856, 1856, 320, 1, 100), (2048, 2048, 320, 1, 32),
          (1728, 1728, 320, 1, 32), (1408, 1408, 320, 1, 32), (1088, 1088, 320, 1, 32),
          (768, 768, 320, 1, 32), (448, 448, 320, 1, 32), (1024, 1024, 320, 1, 128),
          (704, 704, 320, 1, 128), (384, 384, 320, 1, 128), (8192, 640, 640, 0, 1),
          (256, 3584, 512, 0, 1), (256, 15872, 512, 0, 1), (4096, 320, 320, 0, 32)]
if len(sys.argv) > 3:
    a = [int(v) for v in sys.argv[1:]]
    shapes = [(a[0], a[1], a[2], a[3] if len(a) > 3 else 0, a[4] if len(a) > 4 else 1)]
res = {}
for tile in ("", "64", "128"):
    if tile:
        os.environ["BQ_GEMM_TILE"] = tile
    else:
        os.environ.pop("BQ_GEMM_TILE", None)
    e = Engine(0)
    for m, n, k, lo, b in shapes:
        fl = (float(m) * m * k if lo else 2.0 * m * n * k) * b
        reps = max(20, int(100e-3 / (fl / 55e12)))
        ms = e.probe_gemm(m, n, k, lo, b, False, reps)
        res.setdefault((m, n, k, lo, b), {})[tile] = fl / ms / 1e9
    e.close()
print("%-44s %8s %8s %8s" % ("shape", "rule", "tile 64", "tile 128"))
for (m, n, k, lo, b), v in res.items():
    best = max(v["64"], v["128"])
    print("lower=%d m=%5d n=%5d k=%4d batch=%3d        %8.1f %8.1f %8.1f  %s"
          % (lo, m, n, k, b, v[""], v["64"], v["128"], "" if v[""] >= 0.98 * best else "<<< rule off the best"))
