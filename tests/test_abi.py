"""The C-ABI library loads on a GPU-less host and exports every symbol include/nerf_amd.h declares
(no compute call is made here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'nerf_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nerf_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported():
    from nerf_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'run `python -m nerf_amd.build` first'
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), f'{n} declared in nerf_amd.h but not exported'


def test_bindings_cover_header():
    from nerf_amd import _lib
    import nerf_amd.ops  # noqa: F401  (registers every binding)
    import nerf_amd.mlp  # noqa: F401
    import nerf_amd.mlp_bf16  # noqa: F401
    bound = set(_lib.exported_symbols())
    assert set(_declared()) <= bound, sorted(set(_declared()) - bound)
    L = _lib.lib()
    assert L.nerf_abi_version() >= 1
    assert L.nerf_error_string(-4).decode().startswith('value out of range')


def test_cpu_tensors_are_refused():
    import pytest
    import torch
    from nerf_amd import ops
    with pytest.raises(RuntimeError):
        ops.stratified_t(torch.zeros(3, 1), torch.ones(3, 1), 4)
