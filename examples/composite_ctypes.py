"""Binding one entry point of libnerf_amd.so by hand: alpha compositing (the reference's utils.py:283-318 with the
activation glue of model.py:139-140 fused in front).  The integration guide quotes the block between the markers."""
# --- begin snippet
import ctypes, os, torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'nerf_amd', 'lib', 'libnerf_amd.so'))
lib.nerf_composite_fwd.restype = ctypes.c_int
lib.nerf_error_string.restype = ctypes.c_char_p
P = ctypes.c_void_p


class nerf_activation(ctypes.Structure):        # include/nerf_amd.h
    _fields_ = [('apply_activation', ctypes.c_int), ('rgb_padding', ctypes.c_float), ('density_bias', ctypes.c_float)]


def volumetric_rendering_fwd(raw_rgb, raw_density, t_vals, dirs, white_bkgd, rgb_padding=0.001, density_bias=-1.0):
    """raw_rgb [N,S,3], raw_density [N,S], t_vals [N,S+1], dirs [N,3]: contiguous fp32 tensors on one HIP device."""
    n, s = raw_density.shape
    dev = raw_rgb.device
    comp, dist, acc = torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
    w, alpha = torch.empty(n, s, device=dev), torch.empty(n, s, device=dev)
    act = nerf_activation(1, rgb_padding, density_bias)
    rc = lib.nerf_composite_fwd(                     # the 14 arguments of the prototype, in its order
        P(raw_rgb.data_ptr()), P(raw_density.data_ptr()), P(t_vals.data_ptr()), P(dirs.data_ptr()),
        ctypes.c_int64(n), ctypes.c_int(s), ctypes.c_int(int(white_bkgd)), ctypes.byref(act),
        P(comp.data_ptr()), P(dist.data_ptr()), P(acc.data_ptr()), P(w.data_ptr()), P(alpha.data_ptr()),
        P(torch.cuda.current_stream().cuda_stream))
    if rc:
        raise RuntimeError(lib.nerf_error_string(rc).decode())
    return comp, dist, acc, w, alpha
# --- end snippet
