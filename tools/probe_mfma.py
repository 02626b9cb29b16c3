"""fp64 MFMA issue study: sustained TFLOP/s by instruction shape, independent
accumulators per wave and waves per SIMD."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_quadrature_amd import Engine, _lib as L
e = Engine(0)
for kind, name in ((0, "16x16x4"), (1, "4x4x4_4b")):
    for nacc in (1, 2, 4, 8):
        row = []
        for bpc in (1, 2, 4, 8):
            v = C.c_double()
            e._check(e._lib.bq_probe_mfma_variant(e._ctx, kind, nacc, bpc, C.cast(C.byref(v), L._dp)))
            row.append(round(v.value, 1))
        print(name, "nacc", nacc, "waves/SIMD 1,2,4,8:", row)
for kind, name in ((2, "gemm step, no rotation"), (3, "gemm step, 3 DPP rotations per Q fragment")):
    row = []
    for bpc in (1, 2):
        v = C.c_double()
        e._check(e._lib.bq_probe_mfma_variant(e._ctx, kind, 4, bpc, C.cast(C.byref(v), L._dp)))
        row.append(round(v.value, 1))
    print(name, "waves/SIMD 1,2:", row)
