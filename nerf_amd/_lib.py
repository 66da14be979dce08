"""ctypes binding of libnerf_amd.so (the C ABI declared in include/nerf_amd.h).

The library is built in-tree by `python -m nerf_amd.build`.  There is no fallback: if the
shared object is missing, or a tensor is not a contiguous fp32 tensor on a HIP device, the
call raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (NERF_AMD_LIB: another build of the same library, for A/B timing of kernel variants)
LIB_PATH = os.environ.get('NERF_AMD_LIB') or os.path.join(_HERE, 'lib', 'libnerf_amd.so')


class NerfActivation(Structure):
    _fields_ = [('apply_activation', c_int), ('rgb_padding', c_float), ('density_bias', c_float)]


class MlpDims(Structure):
    _fields_ = [('enc_dim', c_int), ('view_dim', c_int), ('hidden', c_int), ('width', c_int), ('sar', c_int)]


_P = c_void_p
_SIGS = {
    'nerf_abi_version': (c_int, []),
    'nerf_error_string': (c_char_p, [c_int]),
    'nerf_stratified_t': (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P, _P]),
    'nerf_cast_rays': (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, _P, _P, _P]),
    'nerf_ipe_encode': (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P, _P]),
    'nerf_pe_encode': (c_int, [_P, c_int64, c_int, c_int, c_int, _P, _P]),
    'nerf_sar_raygen': (c_int, [_P, _P, c_int64, _P, _P, _P]),
    'nerf_pinhole_rays': (c_int, [_P, c_int, c_int, c_int, c_float, c_float, c_float, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    'nerf_convert_to_ndc': (c_int, [_P, _P, c_int64, c_float, c_int, c_int, c_float, _P, _P, _P]),
    'nerf_ndc_rays': (c_int, [_P, c_int, c_int, c_int, c_float, c_float, c_float, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P]),
    'nerf_resample': (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_float, c_int, _P, _P, _P, _P]),
    'nerf_invert_cdf': (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P, _P, _P]),
    'nerf_composite_fwd': (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, POINTER(NerfActivation),
                                   _P, _P, _P, _P, _P, _P]),
    'nerf_composite_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, POINTER(NerfActivation),
                                   _P, _P, _P]),
    'nerf_composite_train': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_float, c_int64, c_int, c_int,
                                     POINTER(NerfActivation), _P, _P, _P, _P, _P, _P, _P, _P]),
    'nerf_scatter_fwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_float,
                                 POINTER(NerfActivation), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'nerf_scatter_fwd_static': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_float,
                                        POINTER(NerfActivation), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'nerf_scatter_bwd': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_float,
                                 POINTER(NerfActivation), _P, _P, _P, _P]),
    'nerf_gemm_f32': (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, _P, c_int64, c_int64, c_int, c_int64,
                              c_int, c_int, _P]),
    'nerf_act_bwd_f32': (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int, c_int, _P]),
    'nerf_matmul_small_f32': (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, _P, c_int64, c_int, c_int, c_int,
                                      c_int, _P]),
    'nerf_colsum_f32': (c_int, [_P, c_int64, c_int64, c_int, _P, _P]),
    'nerf_layernorm_fwd': (c_int, [_P, _P, _P, c_int64, c_int, c_float, _P, _P, _P, _P]),
    'nerf_layernorm_bwd': (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, _P, _P, _P, _P]),
    'nerf_sar_pulse_loss': (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
}

_lib = None


def lib():
    """The loaded library; raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f'{LIB_PATH} not found: build it with `python -m nerf_amd.build` '
                              '(there is no CPU or eager fallback)')
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)           # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def register(name, restype, argtypes):
    """Declare one more entry point (used by the MLP bindings)."""
    _SIGS[name] = (restype, argtypes)
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.restype = restype
        fn.argtypes = argtypes


def exported_symbols():
    return list(_SIGS)


def check(code):
    if code != 0:
        raise RuntimeError(f'libnerf_amd: {lib().nerf_error_string(code).decode()} (code {code})')


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous tensor of the given dtype (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('nerf_amd kernels need tensors on a HIP device; got a CPU tensor')
    if t.dtype != dtype:
        raise RuntimeError(f'expected dtype {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise RuntimeError('expected a contiguous tensor')
    return c_void_p(t.data_ptr())


def f32c(t):
    """Plumbing: make a tensor fp32 + contiguous (no copy when it already is)."""
    return t.detach().to(torch.float32).contiguous()
