"""The C-ABI library loads without a GPU and exports every symbol that
include/bqhip.h declares; the ctypes table covers all of them; the product
fails loudly (no fallback) when no device is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "bqhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bq_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from bayesian_quadrature_amd import _lib
    lib = _lib.load_library()
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), "libbqhip.so does not export %s" % n


def test_ctypes_table_matches_header():
    from bayesian_quadrature_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_device_count_and_loud_failure():
    from bayesian_quadrature_amd import _lib, Engine
    n = C.c_int(-1)
    assert _lib.load_library().bq_device_count(C.byref(n)) == 0
    assert n.value >= 0
    if n.value == 0:
        with pytest.raises(RuntimeError):
            Engine(0)


def test_product_does_not_import_oracle():
    # the oracle is test infrastructure: nothing under the product package may
    # mention it
    pkg = os.path.join(ROOT, "bayesian-quadrature_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower(), f


def test_null_context_is_rejected():
    from bayesian_quadrature_amd import _lib
    lib = _lib.load_library()
    assert lib.bq_ctx_sync(None) == _lib.BQ_ERR_BAD_ARG
    assert lib.bq_set_block(None, 64) == _lib.BQ_ERR_BAD_ARG
