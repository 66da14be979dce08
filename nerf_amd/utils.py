"""Drop-in counterpart of the reference's `utils.py` render functions on HIP kernels.

Same names, argument order and return tuples as the reference (utils.py:8-380).  The
reference's functions only run for 3-D `[B, R, .]` ray tensors; these accept `[N, .]` and
`[B, R, .]` alike (leading dims are flattened for the kernels and restored on return).
Tensors must live on a HIP device: there is no CPU path.
"""
import collections

import torch

from . import ops

Rays = collections.namedtuple('Rays', ('origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'near', 'far'))

_EPS32 = float(torch.finfo(torch.float32).eps)


def namedtuple_map(fn, tup):
    """Apply `fn` to each element of `tup` and cast to `tup`'s namedtuple (utils.py:11-13)."""
    return type(tup)(*map(fn, tup))


def convert_to_ndc(origins, directions, focal, w, h, near=1.):
    """Convert a set of rays to NDC coordinates (utils.py:66-86); device tensors instead of NumPy arrays."""
    return ops.convert_to_ndc(origins, directions, focal, w, h, near)


def to8b(img):
    """utils.py:619-623 on a NumPy image (any leading dims)."""
    import numpy as np
    return (255 * np.clip(np.nan_to_num(np.asarray(img)), 0, 1)).astype(np.uint8)


def to_float(img):
    """utils.py:626-630."""
    import numpy as np
    return (np.asarray(img) / 255.).astype(np.float32)


def _flat(x, last):
    return x.reshape(-1, last)


def cast_rays(t_vals, origins, directions, radii, ray_shape, diag=True):
    """utils.py:168-191 (diagonal covariances only, as everywhere in the reference)."""
    if not diag:
        raise NotImplementedError('full covariances are never used on the render path')
    lead = t_vals.shape[:-1]
    s1 = t_vals.shape[-1]
    mean, cov = ops.cast_rays(_flat(t_vals, s1), _flat(origins, 3), _flat(directions, 3), radii.reshape(-1), ray_shape)
    return mean.reshape(*lead, s1 - 1, 3), cov.reshape(*lead, s1 - 1, 3)


def sample_along_rays(origins, directions, radii, num_samples, near, far, randomized, lindisp, ray_shape,
                      t_rand=None):
    """Stratified sampling + ray casting (utils.py:194-230) -> (t_vals, (means, covs)).

    `t_rand` optionally supplies the U[0,1) jitter; otherwise, when `randomized`, it is drawn
    with `torch.rand` on the rays' device exactly where the reference draws it."""
    lead = origins.shape[:-1]
    n = origins.numel() // 3
    if randomized and t_rand is None:
        t_rand = torch.rand(*lead, num_samples + 1, device=origins.device)
    t = ops.stratified_t(near.reshape(-1), far.reshape(-1), num_samples,
                         t_rand.reshape(n, num_samples + 1) if randomized else None, lindisp)
    mean, cov = ops.cast_rays(t, _flat(origins, 3), _flat(directions, 3), radii.reshape(-1), ray_shape)
    return (t.reshape(*lead, num_samples + 1),
            (mean.reshape(*lead, num_samples, 3), cov.reshape(*lead, num_samples, 3)))


def sorted_piecewise_constant_pdf(bins, weights, num_samples, randomized, u_rand=None):
    """Inverse-CDF sampling of a piecewise-constant PDF (utils.py:16-63).  Unlike the reference
    this does not modify `weights` in place."""
    lead = bins.shape[:-1]
    nb = bins.shape[-1]
    if randomized and u_rand is None:
        u_rand = torch.empty(*lead, num_samples, device=bins.device).uniform_(0, 1 / num_samples - _EPS32)
    out = ops.resample(_flat(bins, nb), _flat(weights, nb - 1), num_samples, 0.0,
                       u_rand.reshape(-1, num_samples) if randomized else None, blur=False)
    return out.reshape(*lead, num_samples)


def resample_along_rays(origins, directions, radii, t_vals, weights, randomized, stop_grad, resample_padding,
                        ray_shape, u_rand=None, num_fenceposts=None):
    """Hierarchical resampling + ray casting (utils.py:233-280) -> (new_t_vals, (means, covs)).

    The reference always calls this with `stop_grad=True`; the kernels implement that case only.
    `num_fenceposts` defaults to `t_vals.shape[-1]` as upstream; another count goes through the
    same body (64 coarse + 128 fine)."""
    if not stop_grad:
        raise NotImplementedError('resampling is only used under no_grad (stop_grad=True) on the render path')
    lead = t_vals.shape[:-1]
    s1 = t_vals.shape[-1]
    n_out = s1 if num_fenceposts is None else int(num_fenceposts)
    with torch.no_grad():
        if randomized and u_rand is None:
            u_rand = torch.empty(*lead, n_out, device=t_vals.device).uniform_(0, 1 / n_out - _EPS32)
        new_t = ops.resample(_flat(t_vals, s1), _flat(weights, s1 - 1), n_out, resample_padding,
                             u_rand.reshape(-1, n_out) if randomized else None, blur=True)
        mean, cov = ops.cast_rays(new_t, _flat(origins, 3), _flat(directions, 3), radii.reshape(-1), ray_shape)
    return (new_t.reshape(*lead, n_out),
            (mean.reshape(*lead, n_out - 1, 3), cov.reshape(*lead, n_out - 1, 3)))


def volumetric_rendering(rgb, density, t_vals, dirs, white_bkgd):
    """utils.py:283-318 -> comp_rgb [..,3], distance [..], acc [..], weights [..,S], alpha [..,S]."""
    lead = t_vals.shape[:-1]
    s = t_vals.shape[-1] - 1
    comp, dist, acc, w, alpha = ops.composite(rgb.reshape(-1, s, 3), density.reshape(-1, s),
                                              _flat(t_vals, s + 1), _flat(dirs, 3), white_bkgd)
    return (comp.reshape(*lead, 3), dist.reshape(lead), acc.reshape(lead), w.reshape(*lead, s),
            alpha.reshape(*lead, s))


def volumetric_scattering(rgb, angle, density, power, t_vals, dirs, pulse_bins, wavelength):
    """utils.py:321-380.  rgb [B,R,S,3], angle [B,R,S,2], density [B,R,S,1], power [1,R] or [B,R],
    t_vals [B,R,S+1], dirs [1,R,3] or [B,R,3], pulse_bins [B,nsam]
    -> pulse [B,nsam,2], distance [B,R], acc [B,R], weights [B,R,S], alpha [B,R,S].

    A distance beyond the last range bin raises IndexError as the reference does (this check
    synchronises the stream)."""
    B, R, S1 = t_vals.shape
    pulses, dists, accs, ws, alphas = [], [], [], [], []
    for b in range(B):
        pw = power[b] if power.shape[0] == B else power[0]
        dr = dirs[b] if dirs.shape[0] == B else dirs[0]
        pulse, dist, acc, w, alpha, _, status = ops.scatter(rgb[b], angle[b], density[b], pw, t_vals[b], dr,
                                                            pulse_bins[b], wavelength)
        if int(status) != 0:
            raise IndexError('range bin index out of bounds: a ray distance lies beyond the last pulse bin')
        pulses.append(pulse); dists.append(dist); accs.append(acc); ws.append(w); alphas.append(alpha)
    return torch.stack(pulses), torch.stack(dists), torch.stack(accs), torch.stack(ws), torch.stack(alphas)
