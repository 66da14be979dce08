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


def _prototypes():
    """name -> number of parameters, for every function prototype in the header."""
    text = open(os.path.join(ROOT, 'include', 'nerf_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    out = {}
    for name, args in re.findall(r'\b(nerf_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;', text):
        args = args.strip()
        out[name] = 0 if args in ('', 'void') else args.count(',') + 1
    return out


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
    assert L.nerf_abi_version() >= 2
    assert L.nerf_error_string(-4).decode().startswith('value out of range')


def test_binding_argument_counts_match_header():
    """Every ctypes signature declares exactly as many arguments as the header's prototype (ctypes lets a cdecl
    call over- or under-supply silently, so drift between the two would otherwise go unseen)."""
    from nerf_amd import _lib
    import nerf_amd.ops  # noqa: F401
    import nerf_amd.mlp  # noqa: F401
    import nerf_amd.mlp_bf16  # noqa: F401
    protos = _prototypes()
    assert set(protos) == set(_declared())
    for name, n_params in protos.items():
        restype, argtypes = _lib._SIGS[name]
        assert len(argtypes) == n_params, (name, len(argtypes), n_params)


def test_cpu_tensors_are_refused():
    import pytest
    import torch
    from nerf_amd import ops
    with pytest.raises(RuntimeError):
        ops.stratified_t(torch.zeros(3, 1), torch.ones(3, 1), 4)


def test_python_surface_has_the_reference_names():
    """Every name the reference's model.py takes from `utils` / `util_modules` (model.py:8-11), and the model
    classes its scripts take from `model` (train.py:7, train_sar.py:14), exists here under the same name."""
    import nerf_amd.model as M
    import nerf_amd.util_modules as UM
    import nerf_amd.utils as U
    for n in ('sample_along_rays', 'resample_along_rays', 'volumetric_rendering', 'namedtuple_map', 'to8b',
              'volumetric_scattering', 'cast_rays', 'sorted_piecewise_constant_pdf', 'convert_to_ndc', 'Rays', 'to_float'):
        assert callable(getattr(U, n)), n
    for n in ('PositionalEncoding', 'MipLRDecay', 'NeRFLoss', 'SARNeRFLoss', 'mse_to_psnr'):
        assert callable(getattr(UM, n)), n
    for cls, methods in ((M.MipNeRF, ('forward', 'render_image', 'training_step', 'validation_step', 'train_val_get',
                                      'configure_optimizers')),
                         (M.SARNeRF, ('forward', 'training_step', 'validation_step', 'train_val_get',
                                      'configure_optimizers'))):
        for name in methods:
            assert callable(getattr(cls, name)), (cls.__name__, name)
