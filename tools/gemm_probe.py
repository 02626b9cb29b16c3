"""ms and TFLOP/s of C (m x n) -= P Q^T through the engine's kernel selection for the shapes the
sweeps and the batched factorisations produce: python tools/gemm_probe.py [m n k [batch]] ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402

e = Engine(0)
shapes = [(256, 3584, 512, 1), (256, 1024, 512, 1), (256, 15872, 512, 1), (256, 8192, 512, 1),
          (1024, 4096, 256, 1), (2048, 2048, 320, 1), (448, 448, 320, 32), (768, 768, 320, 32),
          (2176, 128, 192, 32)]
if len(sys.argv) > 3:
    a = [int(v) for v in sys.argv[1:]]
    shapes = [(a[0], a[1], a[2], a[3] if len(a) > 3 else 1)]
for m, n, k, b in shapes:
    for qt in (False, True):
        ms = e.probe_gemm(m, n, k, 0, b, qt)
        print("m=%5d n=%5d k=%4d batch=%3d qt=%d: %8.2f us  %6.1f TFLOP/s  (tiles64 %d)"
              % (m, n, k, b, qt, ms * 1e3, 2.0 * m * n * k * b / ms / 1e9, (m // 64) * (n // 64) * b),
              flush=True)
e.close()
