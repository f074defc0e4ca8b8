"""The C-ABI boundary: libdpp_hip.so (built for gfx950 by __graft_entry__.build()) loads without a GPU and exports every
symbol include/dpp_hip.h declares, the ctypes binding covers exactly that set, and the product runtime refuses to
start without a GPU instead of falling back to the CPU."""
import ctypes
import os
import re

import pytest

from hipdp import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'dpp_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(dpp_[a-z0-9_]+)\s*\(', hdr)))


def test_header_and_binding_agree():
    decl = declared_symbols()
    assert len(decl) >= 30
    assert sorted(lib.SIGNATURES) == decl


def test_product_library_exports_every_declared_symbol():
    import __graft_entry__
    if not os.path.exists(lib.DEFAULT_LIB):
        __graft_entry__.build()
    so = ctypes.CDLL(lib.DEFAULT_LIB)
    for name in declared_symbols():
        assert hasattr(so, name), name
    assert so.dpp_abi_version() == lib.ABI_VERSION
    # argument validation happens before any launch: callable without a GPU
    assert so.dpp_gemm(None, None) == 10001


def test_no_cpu_fallback_in_product_runtime():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hipdp.runtime import TorchHipRuntime
    with pytest.raises(lib.DppError):
        TorchHipRuntime()
    with pytest.raises(lib.DppError):
        lib.load('/nonexistent/libdpp_hip.so')


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'deep-prior-pp_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(dirpath, f)
