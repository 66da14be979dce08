"""Tensor-level wrappers over the C ABI: allocate outputs with torch (plumbing), pass raw
device pointers + the current HIP stream, and hook the hand-written backward kernels into
autograd.  All tensors are flattened to `[N, ...]` rays here; shape handling for the
reference-compatible surface lives in `nerf_amd.utils`.
"""
import ctypes

import torch

from . import _lib
from ._lib import NerfActivation, check, f32c, lib, ptr, stream

RAY_SHAPES = {'cone': 0, 'cylinder': 1}


def _act(apply=False, rgb_padding=0.0, density_bias=0.0):
    return ctypes.byref(NerfActivation(int(bool(apply)), float(rgb_padding), float(density_bias)))


def stratified_t(near, far, num_samples, t_rand=None, lindisp=False):
    near, far = f32c(near).reshape(-1), f32c(far).reshape(-1)
    n = near.shape[0]
    if t_rand is not None:
        t_rand = f32c(t_rand).reshape(n, num_samples + 1)
    t = torch.empty(n, num_samples + 1, device=near.device, dtype=torch.float32)
    check(lib().nerf_stratified_t(ptr(near), ptr(far), ptr(t_rand), n, num_samples, int(bool(lindisp)), ptr(t),
                                  stream()))
    return t


def cast_rays(t_vals, origins, directions, radii, ray_shape):
    if ray_shape not in RAY_SHAPES:
        raise AssertionError(f'unknown ray_shape {ray_shape!r}')
    t_vals = f32c(t_vals)
    n, s1 = t_vals.shape
    origins, directions = f32c(origins).reshape(n, 3), f32c(directions).reshape(n, 3)
    radii = f32c(radii).reshape(n)
    mean = torch.empty(n, s1 - 1, 3, device=t_vals.device, dtype=torch.float32)
    cov = torch.empty_like(mean)
    check(lib().nerf_cast_rays(ptr(t_vals), ptr(origins), ptr(directions), ptr(radii), n, s1 - 1,
                               RAY_SHAPES[ray_shape], ptr(mean), ptr(cov), stream()))
    return mean, cov


def ipe_encode(mean, cov, min_deg, max_deg, with_variance=False):
    """Integrated positional encoding; with `with_variance` also the per-feature variance (util_modules.py:45)."""
    lead = mean.shape[:-1]
    m, c = f32c(mean).reshape(-1, 3), f32c(cov).reshape(-1, 3)
    out = torch.empty(m.shape[0], 6 * (max_deg - min_deg), device=m.device, dtype=torch.float32)
    var = torch.empty_like(out) if with_variance else None
    check(lib().nerf_ipe_encode(ptr(m), ptr(c), m.shape[0], min_deg, max_deg, ptr(out), ptr(var), stream()))
    if with_variance:
        return out.reshape(*lead, -1), var.reshape(*lead, -1)
    return out.reshape(*lead, -1)


def pe_encode(x, min_deg, max_deg, append_input=False):
    lead = x.shape[:-1]
    xx = f32c(x).reshape(-1, 3)
    width = 6 * (max_deg - min_deg) + (3 if append_input else 0)
    out = torch.empty(xx.shape[0], width, device=xx.device, dtype=torch.float32)
    check(lib().nerf_pe_encode(ptr(xx), xx.shape[0], min_deg, max_deg, int(bool(append_input)), ptr(out), stream()))
    return out.reshape(*lead, -1)


def sar_raygen(az, el):
    az, el = f32c(az).reshape(-1), f32c(el).reshape(-1)
    n = az.shape[0]
    ray_d = torch.empty(n, 3, device=az.device, dtype=torch.float32)
    ray_p = torch.empty(n, device=az.device, dtype=torch.float32)
    check(lib().nerf_sar_raygen(ptr(az), ptr(el), n, ptr(ray_d), ptr(ray_p), stream()))
    return ray_d, ray_p


def pinhole_rays(cam_to_world, h, w, focal, near, far, ids=None, ndc=False):
    """Rays of a pinhole camera model generated on the device (dataloader.py:103-138) -> utils.Rays of `[n, .]`
    tensors.  cam_to_world: [C, 3 or 4, 4]; ids: flat (camera, row, col) indices (int64) or None for every pixel.
    `ndc=True` gives the forward-facing variant in normalised device coordinates (dataloader.py:437-460)."""
    from .utils import Rays
    entry = lib().nerf_ndc_rays if ndc else lib().nerf_pinhole_rays
    c2w = f32c(cam_to_world[:, :3, :4])
    n_cams = c2w.shape[0]
    if ids is None:
        n, idp = n_cams * h * w, None
    else:
        ids = ids.to(torch.int64).contiguous()
        n, idp = ids.numel(), ptr(ids, torch.int64)
    dev = c2w.device
    o, d, v = (torch.empty(n, 3, device=dev, dtype=torch.float32) for _ in range(3))
    r, lm, ne, fa = (torch.empty(n, 1, device=dev, dtype=torch.float32) for _ in range(4))
    check(entry(ptr(c2w), n_cams, h, w, float(focal), float(near), float(far), idp, n, ptr(o), ptr(d), ptr(v), ptr(r),
                ptr(lm), ptr(ne), ptr(fa), stream()))
    return Rays(o, d, v, r, lm, ne, fa)


def ndc_rays(cam_to_world, h, w, focal, near, far, ids=None):
    return pinhole_rays(cam_to_world, h, w, focal, near, far, ids, ndc=True)


def convert_to_ndc(origins, directions, focal, w, h, near=1.):
    """utils.py:66-86 on device tensors `[..., 3]` -> (origins, directions) in normalised device coordinates."""
    o, d = f32c(origins), f32c(directions)
    assert o.shape == d.shape and o.shape[-1] == 3
    n = o.numel() // 3
    oo, dd = torch.empty_like(o), torch.empty_like(d)
    check(lib().nerf_convert_to_ndc(ptr(o), ptr(d), n, float(focal), int(w), int(h), float(near), ptr(oo), ptr(dd),
                                    stream()))
    return oo, dd


def resample(t_vals, weights, n_out, resample_padding, u_rand=None, blur=True, return_idx=False, return_cdf=False):
    t_vals, weights = f32c(t_vals), f32c(weights)
    n, s1 = t_vals.shape
    assert weights.shape == (n, s1 - 1), (weights.shape, t_vals.shape)
    if u_rand is not None:
        u_rand = f32c(u_rand).reshape(n, n_out)
    new_t = torch.empty(n, n_out, device=t_vals.device, dtype=torch.float32)
    idx = torch.empty(n, n_out, device=t_vals.device, dtype=torch.int32) if return_idx else None
    cdf = torch.empty(n, s1, device=t_vals.device, dtype=torch.float32) if return_cdf else None
    check(lib().nerf_resample(ptr(t_vals), ptr(weights), ptr(u_rand), n, s1 - 1, n_out, float(resample_padding),
                              int(bool(blur)), ptr(new_t), ptr(idx, torch.int32), ptr(cdf), stream()))
    out = (new_t,)
    if return_idx:
        out += (idx,)
    if return_cdf:
        out += (cdf,)
    return out if len(out) > 1 else new_t


def invert_cdf(bins, cdf, u):
    bins, cdf, u = f32c(bins), f32c(cdf), f32c(u)
    n, nf = bins.shape
    n_out = u.shape[1]
    samples = torch.empty(n, n_out, device=bins.device, dtype=torch.float32)
    idx = torch.empty(n, n_out, device=bins.device, dtype=torch.int32)
    check(lib().nerf_invert_cdf(ptr(bins), ptr(cdf), ptr(u), n, nf, n_out, ptr(samples), ptr(idx, torch.int32),
                                stream()))
    return samples, idx


class _Composite(torch.autograd.Function):
    """comp_rgb, distance, acc, weights, alpha = composite(rgb, density, t_vals, dirs).

    With `act` set, `rgb`/`density` are the raw MLP outputs and the padding affine and the
    softplus are applied inside the kernels (forward and backward).  `weights`/`alpha` are
    returned detached: the reference only consumes them under no_grad."""

    @staticmethod
    def forward(ctx, rgb, density, t_vals, dirs, white_bkgd, act):
        rgb, density, t_vals, dirs = f32c(rgb), f32c(density), f32c(t_vals), f32c(dirs)
        n, s = density.shape[0], density.shape[1]
        dev = rgb.device
        comp = torch.empty(n, 3, device=dev)
        dist = torch.empty(n, device=dev)
        acc = torch.empty(n, device=dev)
        w = torch.empty(n, s, device=dev)
        alpha = torch.empty(n, s, device=dev)
        check(lib().nerf_composite_fwd(ptr(rgb), ptr(density), ptr(t_vals), ptr(dirs), n, s, int(white_bkgd),
                                       _act(*act), ptr(comp), ptr(dist), ptr(acc), ptr(w), ptr(alpha), stream()))
        ctx.save_for_backward(rgb, density, t_vals, dirs)
        ctx.meta = (n, s, int(white_bkgd), act)
        ctx.mark_non_differentiable(w, alpha)
        return comp, dist, acc, w, alpha

    @staticmethod
    def backward(ctx, g_comp, g_dist, g_acc, _gw, _ga):
        rgb, density, t_vals, dirs = ctx.saved_tensors
        n, s, wb, act = ctx.meta
        g_comp = f32c(g_comp) if g_comp is not None else torch.zeros(n, 3, device=rgb.device)
        g_dist = f32c(g_dist) if g_dist is not None else None
        g_acc = f32c(g_acc) if g_acc is not None else None
        g_rgb = torch.empty_like(rgb)
        g_den = torch.empty_like(density)
        check(lib().nerf_composite_bwd(ptr(rgb), ptr(density), ptr(t_vals), ptr(dirs), ptr(g_comp), ptr(g_dist),
                                       ptr(g_acc), n, s, wb, _act(*act), ptr(g_rgb), ptr(g_den), stream()))
        return g_rgb, g_den, None, None, None, None


def composite(rgb, density, t_vals, dirs, white_bkgd=False, act=(False, 0.0, 0.0)):
    """rgb [N,S,3], density [N,S] or [N,S,1], t_vals [N,S+1], dirs [N,3]."""
    n, s = t_vals.shape[0], t_vals.shape[1] - 1
    return _Composite.apply(rgb.reshape(n, s, 3), density.reshape(n, s), t_vals, dirs, bool(white_bkgd), tuple(act))


class _Scatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, angle, density, power, t_vals, dirs, pulse_bins, wavelength, act):
        rgb, angle, density = f32c(rgb), f32c(angle), f32c(density)
        power, t_vals, dirs, pulse_bins = f32c(power), f32c(t_vals), f32c(dirs), f32c(pulse_bins)
        r, s = density.shape
        nsam = pulse_bins.shape[0]
        dev = rgb.device
        pulse = torch.empty(nsam, 2, device=dev)
        dist = torch.empty(r, device=dev)
        acc = torch.empty(r, device=dev)
        w = torch.empty(r, s, device=dev)
        alpha = torch.empty(r, s, device=dev)
        ray_ret = torch.empty(r, 2, device=dev)
        ray_bin = torch.empty(r, device=dev, dtype=torch.int32)
        comp = torch.empty(r, 5, device=dev)
        winner = torch.empty(nsam, device=dev, dtype=torch.int32)
        status = torch.empty(1, device=dev, dtype=torch.int32)
        check(lib().nerf_scatter_fwd(ptr(rgb), ptr(angle), ptr(density), ptr(power), ptr(t_vals), ptr(dirs),
                                     ptr(pulse_bins), r, s, nsam, float(wavelength), _act(*act), ptr(pulse),
                                     ptr(dist), ptr(acc), ptr(w), ptr(alpha), ptr(ray_ret),
                                     ptr(ray_bin, torch.int32), ptr(comp), ptr(winner, torch.int32),
                                     ptr(status, torch.int32), stream()))
        ctx.save_for_backward(rgb, angle, density, power, t_vals, dirs, ray_bin, comp, dist)
        ctx.meta = (r, s, nsam, float(wavelength), act)
        ctx.status = status
        ctx.mark_non_differentiable(w, alpha, ray_bin, status)
        return pulse, dist, acc, w, alpha, ray_bin, status

    @staticmethod
    def backward(ctx, g_pulse, g_dist, g_acc, *_):
        rgb, angle, density, power, t_vals, dirs, ray_bin, comp, dist = ctx.saved_tensors
        r, s, nsam, wl, act = ctx.meta
        g_pulse = f32c(g_pulse) if g_pulse is not None else torch.zeros(nsam, 2, device=rgb.device)
        g_dist = f32c(g_dist).reshape(-1) if g_dist is not None else None
        g_acc = f32c(g_acc).reshape(-1) if g_acc is not None else None
        g_rgb, g_ang, g_den = torch.empty_like(rgb), torch.empty_like(angle), torch.empty_like(density)
        check(lib().nerf_scatter_bwd(ptr(rgb), ptr(angle), ptr(density), ptr(power), ptr(t_vals), ptr(dirs),
                                     ptr(g_pulse), ptr(ray_bin, torch.int32), ptr(comp), ptr(dist), ptr(g_dist),
                                     ptr(g_acc), r, s, nsam, wl, _act(*act), ptr(g_rgb), ptr(g_ang), ptr(g_den),
                                     stream()))
        return g_rgb, g_ang, g_den, None, None, None, None, None, None


def scatter(rgb, angle, density, power, t_vals, dirs, pulse_bins, wavelength, act=(False, 0.0, 0.0)):
    """One pulse.  rgb [R,S,3], angle [R,S,2], density [R,S], power [R], t_vals [R,S+1], dirs [R,3],
    pulse_bins [nsam] -> pulse [nsam,2], distance [R], acc [R], weights, alpha, ray_bin, status."""
    r, s = t_vals.shape[0], t_vals.shape[1] - 1
    return _Scatter.apply(rgb.reshape(r, s, 3), angle.reshape(r, s, 2), density.reshape(r, s), power.reshape(r),
                          t_vals, dirs.reshape(r, 3), pulse_bins.reshape(-1), wavelength, tuple(act))


SAR_PULSE_MAX_FFT = 8192


def sar_pulse_supported(nsam, nfft):
    """Sizes nerf_sar_pulse_loss takes: the filter's length a power of two up to 8192, 2 <= nsam <= nfft."""
    return 2 <= nsam <= nfft <= SAR_PULSE_MAX_FFT and nfft & (nfft - 1) == 0


def sar_pulse_loss(pulse, mfilt, target, coarse_weight, need_grad=True, need_profile=False):
    """Matched filter, normalisation and loss of one pulse (model.py:377-380, util_modules.py:74-89) in one launch.
    pulse [L, nsam, 2] range profiles, mfilt [nfft] complex64 (or [1, nfft]), target [nsam, 2] (or [1, nsam, 2])
    -> (loss, level_mse [L], profile [L, nsam, 2] or None, g_pulse [L, nsam, 2] = d loss / d pulse or None)."""
    levels, nsam = pulse.shape[0], pulse.shape[1]
    mf = torch.view_as_real(mfilt.reshape(-1).to(torch.complex64)).contiguous()
    nfft = mf.shape[0]
    dev = pulse.device
    pulse, tg = f32c(pulse), f32c(target.reshape(nsam, 2))
    mse = torch.empty(levels, device=dev)
    profile = torch.empty(levels, nsam, 2, device=dev) if need_profile else None
    g = torch.empty(levels, nsam, 2, device=dev) if need_grad else None
    check(lib().nerf_sar_pulse_loss(ptr(pulse), ptr(mf), ptr(tg), levels, nsam, nfft, float(coarse_weight),
                                    ptr(profile) if need_profile else None, ptr(mse), ptr(g) if need_grad else None,
                                    stream()))
    loss = coarse_weight * mse[:-1].sum() + mse[-1]
    return loss, mse, profile, g
