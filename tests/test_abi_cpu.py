"""CPU-side checks of the C-ABI boundary: the shared library loads and exports every
symbol that include/zero_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "zero_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(zk_\w+)\s*\(", text)))


def test_header_symbols_exported():
    syms = declared_symbols()
    assert len(syms) >= 20
    lib = ctypes.CDLL(os.path.join(ROOT, "zero_amd", "libzero_hip.so"))
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/zero_hip.h but not exported"
    assert lib.zk_abi_version() == 6


def test_binding_covers_header():
    from zero_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from zero_amd import ops
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        ops.gemm(a, a)
