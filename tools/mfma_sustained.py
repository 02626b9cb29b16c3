"""What the fp64 matrix cores SUSTAIN on this chip: the GEMM inner step (64 v_mfma_f64_4x4x4_4b
per wave and step, no memory) for 0.2-0.3 s, with operands near 1.0 and with random mantissas, at
1, 2 and 4 waves per SIMD; and the single short launch the round-1 peak probe times."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine  # noqa: E402

e = Engine(0)
print("short launch (kind 2), 2 waves/SIMD: %.1f TFLOP/s" % e.probe_mfma_variant(2, 8, 2))
for kind, name in ((4, "operands near 1.0"), (5, "random mantissas")):
    for w in (1, 2, 4):
        print("sustained, %-18s %d waves/SIMD: %.1f TFLOP/s" % (name, w, e.probe_mfma_variant(kind, 8, w)),
              flush=True)
e.close()
