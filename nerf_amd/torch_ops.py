"""The render kernels as `torch.library` custom ops (namespace `nerf_amd`): thin, traceable wrappers over the C ABI.

`nerf_amd.ops` binds the same entry points through `torch.autograd.Function`s, which is what the models use in eager
mode; the ops here carry a schema, a fake (meta) implementation and a registered backward, so `torch.compile`,
`torch.export` and `make_fx` see them as single opaque nodes (no graph break, no tracing into ctypes) and HIP-graph
capture works unchanged.  All take contiguous fp32 tensors on one HIP device, flattened to `[N, ...]` rays.

    torch.ops.nerf_amd.stratified_t(near, far, num_samples, t_rand, lindisp)             utils.py:194-214
    torch.ops.nerf_amd.cast_ipe(t_vals, origins, directions, radii, ray_shape, lo, hi)    utils.py:168-191, util_modules.py:36-50
    torch.ops.nerf_amd.resample(t_vals, weights, n_out, padding, u_rand, blur)            utils.py:233-280
    torch.ops.nerf_amd.composite(rgb, density, t_vals, dirs, white_bkgd, act, pad, bias)  utils.py:283-318 (differentiable)
    torch.ops.nerf_amd.composite_bwd(...)                                                its hand-written backward
    torch.ops.nerf_amd.scatter(rgb, angle, density, power, t_vals, dirs, pulse_bins, ...) utils.py:321-380 (differentiable)
    torch.ops.nerf_amd.scatter_bwd(...)                                                  its hand-written backward
    torch.ops.nerf_amd.encode_frag(t_vals, origins, directions, radii, ray_shape, lo, hi) cast + IPE as operand fragments
    torch.ops.nerf_amd.view_frag(dirs, lo, hi)                                            util_modules.py:49, model.py:125-126
    torch.ops.nerf_amd.mlp_fwd(weights, enc, view, geom, M, ray_div, n_rays, save)        model.py:117-132, 343-359 (differentiable
                                                                                          w.r.t. the weights)
    torch.ops.nerf_amd.mlp_bwd(weights, geom, M, g_density, g_rgb, g_angle, outs, acts)   the input-gradient chain
    torch.ops.nerf_amd.mlp_wgrad(weights, geom, M, acts, dys)                             the weight gradients
"""
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import ops
from ._lib import check, f32c, lib, ptr, stream

_RAY_SHAPES = ('cone', 'cylinder')


@torch.library.custom_op('nerf_amd::stratified_t', mutates_args=())
def stratified_t(near: Tensor, far: Tensor, num_samples: int, t_rand: Optional[Tensor], lindisp: bool) -> Tensor:
    return ops.stratified_t(near, far, num_samples, t_rand, lindisp)


@stratified_t.register_fake
def _(near, far, num_samples, t_rand, lindisp):
    return near.new_empty(near.numel(), num_samples + 1, dtype=torch.float32)


@torch.library.custom_op('nerf_amd::cast_ipe', mutates_args=())
def cast_ipe(t_vals: Tensor, origins: Tensor, directions: Tensor, radii: Tensor, ray_shape: int, min_deg: int,
             max_deg: int) -> Tensor:
    """Fenceposts -> conical-frustum (0) / cylinder (1) Gaussians -> integrated positional encoding [N, S, 6 L]."""
    mean, cov = ops.cast_rays(t_vals, origins, directions, radii, _RAY_SHAPES[ray_shape])
    return ops.ipe_encode(mean, cov, min_deg, max_deg)


@cast_ipe.register_fake
def _(t_vals, origins, directions, radii, ray_shape, min_deg, max_deg):
    return t_vals.new_empty(t_vals.shape[0], t_vals.shape[1] - 1, 6 * (max_deg - min_deg), dtype=torch.float32)


@torch.library.custom_op('nerf_amd::resample', mutates_args=())
def resample(t_vals: Tensor, weights: Tensor, n_out: int, resample_padding: float, u_rand: Optional[Tensor],
             blur: bool) -> Tensor:
    return ops.resample(t_vals, weights, n_out, resample_padding, u_rand, blur)


@resample.register_fake
def _(t_vals, weights, n_out, resample_padding, u_rand, blur):
    return t_vals.new_empty(t_vals.shape[0], n_out, dtype=torch.float32)


@torch.library.custom_op('nerf_amd::composite', mutates_args=())
def composite(rgb: Tensor, density: Tensor, t_vals: Tensor, dirs: Tensor, white_bkgd: bool, apply_act: bool,
              rgb_padding: float, density_bias: float) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """rgb [N,S,3], density [N,S], t_vals [N,S+1], dirs [N,3] -> comp_rgb [N,3], distance [N], acc [N], weights [N,S]
    (weights carry no gradient: the reference consumes them under no_grad, model.py:104-111)."""
    rgb, density, t_vals, dirs = f32c(rgb), f32c(density), f32c(t_vals), f32c(dirs)
    n, s = density.shape
    dev = rgb.device
    comp, dist, acc = torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
    w, alpha = torch.empty(n, s, device=dev), torch.empty(n, s, device=dev)
    check(lib().nerf_composite_fwd(ptr(rgb), ptr(density), ptr(t_vals), ptr(dirs), n, s, int(white_bkgd),
                                   ops._act(apply_act, rgb_padding, density_bias), ptr(comp), ptr(dist), ptr(acc), ptr(w),
                                   ptr(alpha), stream()))
    return comp, dist, acc, w


@composite.register_fake
def _(rgb, density, t_vals, dirs, white_bkgd, apply_act, rgb_padding, density_bias):
    n, s = density.shape
    return (rgb.new_empty(n, 3), rgb.new_empty(n), rgb.new_empty(n), rgb.new_empty(n, s))


@torch.library.custom_op('nerf_amd::composite_bwd', mutates_args=())
def composite_bwd(rgb: Tensor, density: Tensor, t_vals: Tensor, dirs: Tensor, g_comp: Tensor, g_dist: Tensor,
                  g_acc: Tensor, white_bkgd: bool, apply_act: bool, rgb_padding: float,
                  density_bias: float) -> Tuple[Tensor, Tensor]:
    rgb, density, t_vals, dirs = f32c(rgb), f32c(density), f32c(t_vals), f32c(dirs)
    n, s = density.shape
    g_rgb, g_den = torch.empty_like(rgb), torch.empty_like(density)
    check(lib().nerf_composite_bwd(ptr(rgb), ptr(density), ptr(t_vals), ptr(dirs), ptr(f32c(g_comp)), ptr(f32c(g_dist)),
                                   ptr(f32c(g_acc)), n, s, int(white_bkgd), ops._act(apply_act, rgb_padding, density_bias),
                                   ptr(g_rgb), ptr(g_den), stream()))
    return g_rgb, g_den


@composite_bwd.register_fake
def _(rgb, density, t_vals, dirs, g_comp, g_dist, g_acc, white_bkgd, apply_act, rgb_padding, density_bias):
    return torch.empty_like(rgb), torch.empty_like(density)


def _composite_setup(ctx, inputs, output):
    rgb, density, t_vals, dirs, white_bkgd, apply_act, rgb_padding, density_bias = inputs
    ctx.save_for_backward(rgb, density, t_vals, dirs)
    ctx.meta = (white_bkgd, apply_act, rgb_padding, density_bias)
    ctx.mark_non_differentiable(output[3])      # weights: consumed under no_grad upstream; no gradient is implemented


def _composite_backward(ctx, g_comp, g_dist, g_acc, _g_w):
    rgb, density, t_vals, dirs = ctx.saved_tensors
    n = density.shape[0]
    zeros = lambda *shape: torch.zeros(*shape, device=rgb.device)
    g_rgb, g_den = composite_bwd(rgb, density, t_vals, dirs,
                                 g_comp if g_comp is not None else zeros(n, 3),
                                 g_dist if g_dist is not None else zeros(n),
                                 g_acc if g_acc is not None else zeros(n), *ctx.meta)
    return g_rgb, g_den, None, None, None, None, None, None


composite.register_autograd(_composite_backward, setup_context=_composite_setup)


def volumetric_rendering(rgb, density, t_vals, dirs, white_bkgd):
    """utils.py:283-318 through the custom op, `[..., S, .]` inputs as in the reference -> comp_rgb, distance, acc,
    weights (the reference's fifth output, alpha, is `nerf_amd.utils.volumetric_rendering`'s)."""
    lead = t_vals.shape[:-1]
    s = t_vals.shape[-1] - 1
    comp, dist, acc, w = composite(rgb.reshape(-1, s, 3), density.reshape(-1, s), t_vals.reshape(-1, s + 1),
                                   dirs.reshape(-1, 3), bool(white_bkgd), False, 0.0, 0.0)
    return comp.reshape(*lead, 3), dist.reshape(lead), acc.reshape(lead), w.reshape(*lead, s)


@torch.library.custom_op('nerf_amd::scatter', mutates_args=())
def scatter(rgb: Tensor, angle: Tensor, density: Tensor, power: Tensor, t_vals: Tensor, dirs: Tensor, pulse_bins: Tensor,
            wavelength: float, apply_act: bool, rgb_padding: float,
            density_bias: float) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """One SAR pulse: rgb [R,S,3], angle [R,S,2], density [R,S], power [R], t_vals [R,S+1], dirs [R,3], pulse_bins [nsam]
    -> pulse [nsam,2], distance [R], acc [R], weights [R,S], and for the backward pass / the caller's range check:
    ray_bin [R] int32, the per-ray composites [R,5], status [1] int32 (non-zero: a distance beyond the last bin, where
    the reference's indexed write raises, utils.py:379)."""
    rgb, angle, density = f32c(rgb), f32c(angle), f32c(density)
    power, t_vals, dirs, pulse_bins = f32c(power), f32c(t_vals), f32c(dirs), f32c(pulse_bins)
    r, s = density.shape
    nsam = pulse_bins.shape[0]
    dev = rgb.device
    pulse, dist, acc = torch.empty(nsam, 2, device=dev), torch.empty(r, device=dev), torch.empty(r, device=dev)
    w, alpha = torch.empty(r, s, device=dev), torch.empty(r, s, device=dev)
    ray_ret, comp = torch.empty(r, 2, device=dev), torch.empty(r, 5, device=dev)
    ray_bin = torch.empty(r, device=dev, dtype=torch.int32)
    winner = torch.empty(nsam, device=dev, dtype=torch.int32)
    status = torch.empty(1, device=dev, dtype=torch.int32)
    check(lib().nerf_scatter_fwd(ptr(rgb), ptr(angle), ptr(density), ptr(power), ptr(t_vals), ptr(dirs), ptr(pulse_bins),
                                 r, s, nsam, float(wavelength), ops._act(apply_act, rgb_padding, density_bias), ptr(pulse),
                                 ptr(dist), ptr(acc), ptr(w), ptr(alpha), ptr(ray_ret), ptr(ray_bin, torch.int32),
                                 ptr(comp), ptr(winner, torch.int32), ptr(status, torch.int32), stream()))
    return pulse, dist, acc, w, ray_bin, comp, status


@scatter.register_fake
def _(rgb, angle, density, power, t_vals, dirs, pulse_bins, wavelength, apply_act, rgb_padding, density_bias):
    r, s = density.shape
    e = rgb.new_empty
    return (e(pulse_bins.shape[0], 2), e(r), e(r), e(r, s), e(r, dtype=torch.int32), e(r, 5), e(1, dtype=torch.int32))


@torch.library.custom_op('nerf_amd::scatter_bwd', mutates_args=())
def scatter_bwd(rgb: Tensor, angle: Tensor, density: Tensor, power: Tensor, t_vals: Tensor, dirs: Tensor,
                g_pulse: Tensor, ray_bin: Tensor, comp: Tensor, dist: Tensor, g_dist: Tensor, g_acc: Tensor, nsam: int,
                wavelength: float, apply_act: bool, rgb_padding: float,
                density_bias: float) -> Tuple[Tensor, Tensor, Tensor]:
    rgb, angle, density = f32c(rgb), f32c(angle), f32c(density)
    r, s = density.shape
    g_rgb, g_ang, g_den = torch.empty_like(rgb), torch.empty_like(angle), torch.empty_like(density)
    check(lib().nerf_scatter_bwd(ptr(rgb), ptr(angle), ptr(density), ptr(f32c(power)), ptr(f32c(t_vals)), ptr(f32c(dirs)),
                                 ptr(f32c(g_pulse)), ptr(ray_bin, torch.int32), ptr(f32c(comp)), ptr(f32c(dist)),
                                 ptr(f32c(g_dist)), ptr(f32c(g_acc)), r, s, nsam, float(wavelength),
                                 ops._act(apply_act, rgb_padding, density_bias), ptr(g_rgb), ptr(g_ang), ptr(g_den), stream()))
    return g_rgb, g_ang, g_den


@scatter_bwd.register_fake
def _(rgb, angle, density, power, t_vals, dirs, g_pulse, ray_bin, comp, dist, g_dist, g_acc, nsam, wavelength, apply_act,
      rgb_padding, density_bias):
    return torch.empty_like(rgb), torch.empty_like(angle), torch.empty_like(density)


def _scatter_setup(ctx, inputs, output):
    rgb, angle, density, power, t_vals, dirs, pulse_bins, wavelength, apply_act, rgb_padding, density_bias = inputs
    pulse, dist, acc, w, ray_bin, comp, status = output
    ctx.save_for_backward(rgb, angle, density, power, t_vals, dirs, ray_bin, comp, dist)
    ctx.meta = (pulse_bins.shape[0], wavelength, apply_act, rgb_padding, density_bias)
    ctx.mark_non_differentiable(w, ray_bin, comp, status)


def _scatter_backward(ctx, g_pulse, g_dist, g_acc, *_):
    rgb, angle, density, power, t_vals, dirs, ray_bin, comp, dist = ctx.saved_tensors
    nsam = ctx.meta[0]
    r = density.shape[0]
    zeros = lambda *shape: torch.zeros(*shape, device=rgb.device)
    g_rgb, g_ang, g_den = scatter_bwd(rgb, angle, density, power, t_vals, dirs,
                                      g_pulse if g_pulse is not None else zeros(nsam, 2), ray_bin, comp, dist,
                                      g_dist if g_dist is not None else zeros(r),
                                      g_acc if g_acc is not None else zeros(r), *ctx.meta)
    return (g_rgb, g_ang, g_den) + (None,) * 8


scatter.register_autograd(_scatter_backward, setup_context=_scatter_setup)


# ---- the fused bf16 MLP (csrc/mlp_bf16.hip) -------------------------------------------------------------------------
# `weights`: the network's parameters in `Geom.param_names()` order (the reference's nn.Linear / LayerNorm tensors,
# model.py:54-93, 258-302); `geom` = [enc_dim, view_dim, hidden, width, sar, operand format] (`Geom.ints()`).
# Buffers that a geometry does not use (None in the eager wrappers) are zero-length tensors here.

def _geom(geom):
    from . import mlp_bf16
    return mlp_bf16.Geom.from_ints(list(geom))


def _byte_buffers(sizes, nblk, like):
    return [like.new_empty(nblk * b if b is not None else 0, dtype=torch.uint8) for b in sizes]


def _opt(buffers):
    return [b if b.numel() > 0 else None for b in buffers]


@torch.library.custom_op('nerf_amd::encode_frag', mutates_args=())
def encode_frag(t_vals: Tensor, origins: Tensor, directions: Tensor, radii: Tensor, ray_shape: int, min_deg: int,
                max_deg: int) -> Tensor:
    """Fenceposts -> Gaussians -> integrated positional encoding as bf16 operand fragments (a byte buffer)."""
    from . import mlp_bf16
    return mlp_bf16.encode_frag(t_vals, origins, directions, radii, _RAY_SHAPES[ray_shape], min_deg, max_deg)


@encode_frag.register_fake
def _(t_vals, origins, directions, radii, ray_shape, min_deg, max_deg):
    n, s1 = t_vals.shape
    nblk = ((n * (s1 - 1) + 255) // 256) * 4
    return t_vals.new_empty(nblk * ((6 * (max_deg - min_deg) + 15) // 16) * 2 * 1024, dtype=torch.uint8)


@torch.library.custom_op('nerf_amd::view_frag', mutates_args=())
def view_frag(dirs: Tensor, min_deg: int, max_deg: int) -> Tensor:
    from . import mlp_bf16
    return mlp_bf16.view_frag(dirs, min_deg, max_deg)


@view_frag.register_fake
def _(dirs, min_deg, max_deg):
    return dirs.new_empty(dirs.shape[0] * ((6 * (max_deg - min_deg) + 3 + 15) // 16) * 2 * 16, dtype=torch.uint8)


@torch.library.custom_op('nerf_amd::mlp_fwd', mutates_args=())
def mlp_fwd(weights: Sequence[Tensor], enc: Tensor, view: Optional[Tensor], geom: Sequence[int], n_samples: int,
            ray_div: int, n_rays: int, save: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, List[Tensor]]:
    """-> raw_density [M,1], raw_rgb [M,3] (after the sigmoid), raw_angle [M,2] and LayerNorm statistics [M,2] (SAR; else
    empty), and the buffers the backward pass reads (empty list when `save` is false)."""
    from . import mlp_bf16
    g = _geom(geom)
    pk = mlp_bf16.packed_for_params(list(weights), g, need_bwd=save)
    d, rgb, ang, acts, ln = mlp_bf16.mlp_forward(pk, enc, view, n_samples, ray_div, n_rays, save=save)
    e = lambda x, *shape: x if x is not None else enc.new_empty(*shape, dtype=torch.float32)
    acts = [a if a is not None else enc.new_empty(0) for a in acts] if acts is not None else []
    return d.reshape(-1, 1), rgb, e(ang, 0, 2), e(ln, 0, 2), acts


@mlp_fwd.register_fake
def _(weights, enc, view, geom, n_samples, ray_div, n_rays, save):
    from . import mlp_bf16
    g = _geom(geom)
    f = lambda *shape: enc.new_empty(*shape, dtype=torch.float32)
    acts = _byte_buffers(mlp_bf16.act_block_bytes(g), mlp_bf16.n_blocks(n_samples), enc) if save else []
    return (f(n_samples, 1), f(n_samples, 3), f(n_samples if g.sar else 0, 2), f(n_samples if (g.sar and save) else 0, 2),
            acts)


@torch.library.custom_op('nerf_amd::mlp_bwd', mutates_args=())
def mlp_bwd(weights: Sequence[Tensor], geom: Sequence[int], n_samples: int, g_density: Tensor, g_rgb: Tensor,
            g_angle: Tensor, raw_density: Tensor, raw_rgb: Tensor, raw_angle: Tensor, ln_stats: Tensor,
            acts: Sequence[Tensor]) -> List[Tensor]:
    """The input-gradient chain: dL/d(raw outputs) -> dL/d(every layer's pre-activation) as transposed fragments."""
    from . import mlp_bf16
    g = _geom(geom)
    pk = mlp_bf16.packed_for_params(list(weights), g, need_bwd=True)
    sar = (g_angle, raw_density.reshape(-1), raw_angle, ln_stats) if g.sar else (None, None, None, None)
    dys = mlp_bf16.mlp_backward(pk, _opt(acts), n_samples, g_density.reshape(-1), g_rgb, raw_rgb, *sar)
    return [x if x is not None else g_rgb.new_empty(0, dtype=torch.uint8) for x in dys]


@mlp_bwd.register_fake
def _(weights, geom, n_samples, g_density, g_rgb, g_angle, raw_density, raw_rgb, raw_angle, ln_stats, acts):
    from . import mlp_bf16
    return _byte_buffers(mlp_bf16.dy_block_bytes(_geom(geom)), mlp_bf16.n_blocks(n_samples), acts[0])


@torch.library.custom_op('nerf_amd::mlp_wgrad', mutates_args=())
def mlp_wgrad(weights: Sequence[Tensor], geom: Sequence[int], n_samples: int, acts: Sequence[Tensor],
              dys: Sequence[Tensor]) -> List[Tensor]:
    """The weight-gradient pass: one gradient per entry of `weights`, in the same order."""
    from . import mlp_bf16
    g = _geom(geom)
    names = g.param_names()
    pk = mlp_bf16.packed_for_params(list(weights), g, need_bwd=True)
    grads = {k: torch.zeros_like(w, dtype=torch.float32) for k, w in zip(names, weights)}
    if g.sar:
        dev = weights[0].device
        grads['_dens.weight'], grads['_dens.bias'] = torch.zeros(3, g.H, device=dev), torch.zeros(3, device=dev)
        grads['_heads.weight'] = torch.zeros(5, g.W if g.view else g.H, device=dev)
        grads['_heads.bias'] = torch.zeros(5, device=dev)
    mlp_bf16.mlp_weight_grads(pk, _opt(acts), _opt(dys), n_samples, grads)
    if g.sar:
        grads.update(mlp_bf16.sar_head_gradients({k: w.detach() for k, w in zip(names, weights)}, grads))
    return [grads[k] for k in names]


@mlp_wgrad.register_fake
def _(weights, geom, n_samples, acts, dys):
    return [torch.empty_like(w, dtype=torch.float32) for w in weights]


def _mlp_setup(ctx, inputs, output):
    weights, enc, view, geom, n_samples, ray_div, n_rays, save = inputs
    d, rgb, ang, ln, acts = output
    ctx.geom, ctx.n_samples, ctx.n_weights, ctx.saved_acts = list(geom), n_samples, len(weights), save
    ctx.save_for_backward(*weights, d, rgb, ang, ln, *acts)
    ctx.mark_non_differentiable(ln, *acts)


def _mlp_backward(ctx, g_density, g_rgb, g_angle, _g_ln, _g_acts):
    if not ctx.saved_acts:
        raise RuntimeError('nerf_amd::mlp_fwd was called with save=False: nothing was kept for a backward pass')
    nw = ctx.n_weights
    weights, (d, rgb, ang, ln), acts = ctx.saved_tensors[:nw], ctx.saved_tensors[nw:nw + 4], ctx.saved_tensors[nw + 4:]
    z = lambda g, like: g.contiguous() if g is not None else torch.zeros_like(like)
    dys = mlp_bwd(weights, ctx.geom, ctx.n_samples, z(g_density, d), z(g_rgb, rgb), z(g_angle, ang), d, rgb, ang, ln, acts)
    grads = mlp_wgrad(weights, ctx.geom, ctx.n_samples, acts, dys)
    return grads, None, None, None, None, None, None, None


mlp_fwd.register_autograd(_mlp_backward, setup_context=_mlp_setup)
