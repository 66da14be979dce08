"""Counterpart of the reference's `model.py`: `MipNeRF` and `SARNeRF` with the same constructor
(`config`, `return_raw`), the same submodule / `state_dict` key names (so upstream checkpoints
load and `torch.save(model.state_dict())` is interchangeable) and the same public methods
(`forward`, `render_image`, `configure_optimizers`, `training_step`, `validation_step`,
`train_val_get`, `train`, `eval`) -- model.py:26-216, 228-428 -- running on the HIP kernels.

The `nn.Linear` / `nn.LayerNorm` submodules only *hold* the parameters; their `forward` is never
called.  `config.mlp_precision` selects the MLP path: 'fp32' (layer-by-layer fp32 matrix-core GEMMs, default,
1e-4 parity) or 'bf16' (fused MFMA chain, 2e-2 parity).  `config.level_samples` (optional list) gives the
number of intervals per level; upstream uses `config.num_samples` for every level.

The reference subclasses pytorch_lightning.LightningModule; this module only needs `device`,
`log_dict`, `optimizers`, `lr_schedulers` and `manual_backward` from it, which `_RenderModule`
provides so the package has no Lightning dependency.
"""
import numpy as np
import torch
import torch.nn as nn
from torch import optim

from . import mlp, ops
from .util_modules import MipLRDecay, NeRFLoss, PositionalEncoding, SARNeRFLoss, mse_to_psnr  # noqa: F401 (model.py:430 exports it too)
from .utils import namedtuple_map

SAR_AZ_SIGMA = 0.039269908169872414     # model.py:319
SAR_EL_SIGMA = 0.09599310885968812      # model.py:320


def to8b(img):
    """utils.py:619-623 (NaN -> 0, +-inf -> the clip bounds)."""
    return (255 * np.clip(np.nan_to_num(img), 0, 1)).astype(np.uint8)


def _xavier_init(model):
    """Xavier-uniform weights; biases keep nn.Linear's default init (model.py:219-225)."""
    for module in model.modules():
        if isinstance(module, nn.Linear):
            nn.init.xavier_uniform_(module.weight)


class _RenderModule(nn.Module):
    """The slice of the LightningModule interface the render models use."""

    def __init__(self):
        super().__init__()
        self.automatic_optimization = False
        self._optim = None
        self.logged = {}

    @property
    def device(self):
        p = next(self.parameters(), None)
        return p.device if p is not None else torch.device('cpu')

    def log_dict(self, metrics, **_):
        self.logged = {k: (float(v) if torch.is_tensor(v) else v) for k, v in metrics.items()}

    def _ensure_optim(self):
        if self._optim is None:
            self._optim = self.configure_optimizers()
        return self._optim

    def optimizers(self):
        return self._ensure_optim()['optimizer']

    def lr_schedulers(self):
        return self._ensure_optim()['lr_scheduler']

    def manual_backward(self, loss):
        loss.backward()

    def configure_optimizers(self):
        cfg = self.config
        optimizer = optim.AdamW(self.parameters(), lr=cfg.lr_init, weight_decay=cfg.weight_decay)
        scheduler = MipLRDecay(optimizer, lr_init=cfg.lr_init, lr_final=cfg.lr_final, max_steps=cfg.max_steps,
                               lr_delay_steps=cfg.lr_delay_steps, lr_delay_mult=cfg.lr_delay_mult)
        return {'optimizer': optimizer, 'lr_scheduler': scheduler}

    def training_step(self, batch, batch_idx):
        opt = self.optimizers()
        train_loss = self.train_val_get(batch, batch_idx)
        opt.zero_grad()
        self.manual_backward(train_loss)
        opt.step()
        self.lr_schedulers().step()
        self._weights_changed()
        return train_loss

    def validation_step(self, batch, batch_idx):
        with torch.no_grad():
            self.train_val_get(batch, batch_idx, 'val')

    def train(self, mode=True):
        self.randomized = self.init_randomized
        super().train(mode)
        return self

    def eval(self):
        # As upstream (model.py:209-216): nn.Module.eval() calls self.train(False), which restores
        # `randomized = init_randomized`, so eval() does NOT switch the sampling jitter off.  Set
        # `model.randomized = False` (or config.randomized) for a deterministic render.
        self.randomized = False
        return super().eval()

    def _weights_changed(self):
        """Hook: packed bf16 weight copies go stale after an optimizer step."""
        self._packed = None

    # ---- shared pieces ---------------------------------------------------------------------
    def _build_trunk(self, config):
        H, D = config.hidden, self.density_input

        def stack(first_in):
            return nn.Sequential(nn.Linear(first_in, H), nn.ReLU(True), nn.Linear(H, H), nn.ReLU(True),
                                 nn.Linear(H, H), nn.ReLU(True), nn.Linear(H, H), nn.ReLU(True))
        self.density_net0 = stack(D)
        self.density_net1 = stack(D + H)

    def _param_dict(self):
        return {k: v for k, v in self.named_parameters() if not k.endswith('scales')}

    def _level_samples(self):
        ls = getattr(self.config, 'level_samples', None)
        return list(ls) if ls else [self.num_samples] * self.num_levels

    def _precision(self):
        return getattr(self.config, 'mlp_precision', 'fp32')


class MipNeRF(_RenderModule):
    def __init__(self, config=None, return_raw=False):
        super().__init__()
        self.config = config
        self.use_viewdirs = config.use_viewdirs
        self.init_randomized = config.randomized
        self.randomized = config.randomized
        self.ray_shape = config.ray_shape
        self.white_bkgd = config.white_bkgd
        self.num_levels = config.num_levels
        self.num_samples = config.num_samples
        self.density_input = (config.max_deg - config.min_deg) * 3 * 2
        self.rgb_input = 3 + ((config.viewdirs_max_deg - config.viewdirs_min_deg) * 3 * 2)
        self.density_noise = config.density_noise
        self.rgb_padding = config.rgb_padding
        self.resample_padding = config.resample_padding
        self.density_bias = config.density_bias
        self.hidden = config.hidden
        self.return_raw = return_raw
        self.density_activation = nn.Softplus()
        self.loss_function = NeRFLoss(config.coarse_weight_decay)
        self.positional_encoding = PositionalEncoding(config.min_deg, config.max_deg)
        self._build_trunk(config)
        self.final_density = nn.Sequential(nn.Linear(config.hidden, 1))
        input_shape = config.hidden
        if self.use_viewdirs:
            # width of the colour layer is `num_samples`, as upstream (model.py:80,87)
            input_shape = config.num_samples
            self.rgb_net0 = nn.Sequential(nn.Linear(config.hidden, config.hidden))
            self.viewdirs_encoding = PositionalEncoding(config.viewdirs_min_deg, config.viewdirs_max_deg)
            self.rgb_net1 = nn.Sequential(nn.Linear(config.hidden + self.rgb_input, config.num_samples), nn.ReLU(True))
        self.final_rgb = nn.Sequential(nn.Linear(input_shape, 3), nn.Sigmoid())
        _xavier_init(self)
        self._packed = None

    # -- one level of the MLP ----------------------------------------------------------------
    def _query(self, t_vals, rays, view_feat, S):
        cfg = self.config
        if self._precision() == 'bf16':
            from . import mlp_bf16
            return mlp_bf16.query_nerf(self, t_vals, rays, view_feat, S)
        n = t_vals.shape[0]
        with torch.no_grad():
            mean, cov = ops.cast_rays(t_vals, rays.origins, rays.directions, rays.radii, self.ray_shape)
            enc = ops.ipe_encode(mean, cov, cfg.min_deg, cfg.max_deg).reshape(n * S, -1)
        view_rows = None
        if view_feat is not None:
            view_rows = view_feat[:, None, :].expand(-1, S, -1).reshape(-1, view_feat.shape[-1])   # model.py:127-128
        return mlp.mlp_forward_fp32(self._param_dict(), enc, view_rows, sar=False)

    def forward(self, rays, t_rand=None, u_rand=None, noise=None):
        """rays: `Rays` of `[N, .]` tensors on the model's device.  Optional pre-drawn random
        tensors (`t_rand [N,S0+1]`, `u_rand {level: [N,S_l+1]}`, `noise {level: [N,S_l,1]}`) replace
        the draws the reference makes internally, in the same places (model.py:100-151)."""
        cfg = self.config
        if self._precision() == 'bf16' and (torch.compiler.is_compiling() or getattr(cfg, 'use_custom_ops', False)):
            return self._forward_custom_ops(rays, t_rand, u_rand, noise)
        levels = self._level_samples()
        o, d = rays.origins, rays.directions
        n = o.shape[0]
        if not self.use_viewdirs:
            view_feat = None
        elif self._precision() == 'bf16':
            from . import mlp_bf16
            view_feat = mlp_bf16.view_frag(rays.viewdirs, cfg.viewdirs_min_deg, cfg.viewdirs_max_deg)
        else:
            view_feat = ops.pe_encode(rays.viewdirs, cfg.viewdirs_min_deg, cfg.viewdirs_max_deg, append_input=True)
        comp_rgbs, distances, accs = [], [], []
        t_vals = weights = rgb = density = None
        for l, S in enumerate(levels):
            with torch.no_grad():
                if l == 0:
                    tr = None
                    if self.randomized:
                        tr = t_rand if t_rand is not None else torch.rand(n, S + 1, device=o.device)
                    t_vals = ops.stratified_t(rays.near, rays.far, S, tr, False)
                else:
                    ur = None
                    if self.randomized:
                        ur = (u_rand[l] if u_rand is not None else
                              torch.empty(n, S + 1, device=o.device).uniform_(0, 1 / (S + 1) - float(torch.finfo(torch.float32).eps)))
                    t_vals = ops.resample(t_vals, weights, S + 1, self.resample_padding, ur, blur=True)
            if self._precision() == 'bf16':
                from . import mlp_bf16
                if mlp_bf16.can_render_level(self, S):      # inference: MLP and compositing of the level in one launch
                    comp_rgb, distance, acc, weights = mlp_bf16.render_level(self, t_vals, rays, view_feat, S,
                                                                             need_weights=l + 1 < len(levels))
                    comp_rgbs.append(comp_rgb)
                    distances.append(distance)
                    accs.append(acc)
                    continue
            raw_density, raw_rgb = self._query(t_vals, rays, view_feat, S)
            raw_density = raw_density.reshape(n, S, 1)
            raw_rgb = raw_rgb.reshape(n, S, 3)
            if self.randomized and self.density_noise:
                nz = noise[l] if noise is not None else torch.rand(raw_density.shape, device=o.device)
                raw_density = raw_density + self.density_noise * nz
            comp_rgb, distance, acc, weights, alpha = ops.composite(
                raw_rgb, raw_density, t_vals, d, self.white_bkgd, act=(True, self.rgb_padding, self.density_bias))
            comp_rgbs.append(comp_rgb)
            distances.append(distance)
            accs.append(acc)
        out = (torch.stack(comp_rgbs), torch.stack(distances), torch.stack(accs))
        if self.return_raw:
            with torch.no_grad():
                rgb = raw_rgb * (1 + 2 * self.rgb_padding) - self.rgb_padding
                density = torch.nn.functional.softplus(raw_density + self.density_bias)
                raws = torch.cat((rgb, density), -1).cpu()
            return out + (raws,)
        return out

    def _forward_custom_ops(self, rays, t_rand=None, u_rand=None, noise=None):
        """`forward` on the bf16 chain written with the `torch.ops.nerf_amd.*` custom ops only (nerf_amd.torch_ops):
        every kernel is one opaque, differentiable node, so `torch.compile(fullgraph=True)`, `torch.export` and
        `make_fx` trace the whole coarse + fine render (model.py:100-151) without a graph break.  Taken automatically
        while a compiler traces, or with `config.use_custom_ops = True`; same kernels, same results as the eager
        wrappers (which add the persistent weight cache and the in-place gradient accumulation of the train loop)."""
        from . import mlp_bf16, torch_ops
        K = torch.ops.nerf_amd
        cfg = self.config
        levels = self._level_samples()
        o, d = rays.origins, rays.directions
        n = o.shape[0]
        geom = mlp_bf16.Geom.of_model(self)
        params = dict(self.named_parameters())
        weights = [params[k] for k in geom.param_names()]
        shape = mlp_bf16.RAY_SHAPES[self.ray_shape]
        radii = rays.radii.reshape(n)
        view = K.view_frag(rays.viewdirs, cfg.viewdirs_min_deg, cfg.viewdirs_max_deg) if self.use_viewdirs else None
        save = torch.is_grad_enabled()
        eps = float(torch.finfo(torch.float32).eps)
        comp_rgbs, distances, accs = [], [], []
        t_vals = weights_l = raw_rgb = raw_density = None
        for l, S in enumerate(levels):
            if l == 0:
                tr = None
                if self.randomized:
                    tr = t_rand if t_rand is not None else torch.rand(n, S + 1, device=o.device)
                t_vals = K.stratified_t(rays.near, rays.far, S, tr, False)
            else:
                ur = None
                if self.randomized:
                    ur = u_rand[l] if u_rand is not None else torch.rand(n, S + 1, device=o.device) * (1 / (S + 1) - eps)
                t_vals = K.resample(t_vals, weights_l.detach(), S + 1, float(self.resample_padding), ur, True)
            enc = K.encode_frag(t_vals, o, d, radii, shape, cfg.min_deg, cfg.max_deg)
            raw_density, raw_rgb, _, _, _ = K.mlp_fwd(weights, enc, view, geom.ints(), n * S, S, n, save)
            raw_density = raw_density.reshape(n, S)
            raw_rgb = raw_rgb.reshape(n, S, 3)
            if self.randomized and self.density_noise:
                nz = noise[l].reshape(n, S) if noise is not None else torch.rand(n, S, device=o.device)
                raw_density = raw_density + self.density_noise * nz
            comp_rgb, distance, acc, weights_l = K.composite(raw_rgb, raw_density, t_vals, d, bool(self.white_bkgd), True,
                                                             float(self.rgb_padding), float(self.density_bias))
            comp_rgbs.append(comp_rgb)
            distances.append(distance)
            accs.append(acc)
        out = (torch.stack(comp_rgbs), torch.stack(distances), torch.stack(accs))
        if self.return_raw:
            with torch.no_grad():
                rgb = raw_rgb * (1 + 2 * self.rgb_padding) - self.rgb_padding
                density = torch.nn.functional.softplus(raw_density[..., None] + self.density_bias)
                raws = torch.cat((rgb, density), -1).cpu()
            return out + (raws,)
        return out

    def render_image(self, rays, height, width, chunks=8192):
        """Chunked no-grad render of one image's rays; keeps the last level (model.py:153-175)."""
        length = rays[0].shape[0]
        dev = self.device
        # outputs stay in HBM until the image is complete: one device-to-host copy (and one synchronisation)
        # per image instead of three per chunk; colours are quantised to 8 bits on the device first
        rgbs = torch.empty(length, 3, device=dev, dtype=torch.uint8)
        dists = torch.empty(length, device=dev, dtype=torch.float32)
        accs = torch.empty(length, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for i in range(0, length, chunks):
                chunk_rays = namedtuple_map(lambda r: r[i:i + chunks].to(dev, non_blocking=True), rays)
                rgb, distance, acc = self(chunk_rays)[:3]
                rgbs[i:i + chunks] = (255 * torch.nan_to_num(rgb[-1]).clamp(0, 1)).to(torch.uint8)      # to8b, utils.py:619-623
                dists[i:i + chunks] = distance[-1].reshape(-1)
                accs[i:i + chunks] = acc[-1].reshape(-1)
        rgbs = rgbs.reshape(height, width, 3).cpu().numpy()
        dists = dists.reshape(height, width).cpu().numpy()
        accs = accs.reshape(height, width).cpu().numpy()
        return rgbs, dists, accs

    def train_val_get(self, batch, batch_idx, kind='train'):
        rays, pixels = batch
        rgb = self.forward(rays)[0]
        train_loss, psnrs = self.loss_function(rgb, pixels, rays.lossmult.to(self.device))
        self.log_dict({f'{kind}_loss': train_loss, 'coarse_psnr': torch.mean(psnrs[:-1]), 'fine_psnr': psnrs[-1],
                       'avg_psnr': torch.mean(psnrs), 'LR': self.lr_schedulers().get_last_lr()[0]}, on_epoch=True,
                      prog_bar=True, rank_zero_only=True)
        return train_loss


class SARNeRF(_RenderModule):
    def __init__(self, config=None, return_raw: bool = False, *args, **kwargs):
        super().__init__()
        self.config = config
        self.use_viewdirs = config.use_viewdirs
        self.init_randomized = config.randomized
        self.randomized = config.randomized
        self.ray_shape = config.ray_shape
        self.num_levels = config.num_levels
        self.num_samples = config.num_samples
        self.density_input = (config.max_deg - config.min_deg) * 3 * 2
        self.rgb_input = 3 + ((config.viewdirs_max_deg - config.viewdirs_min_deg) * 3 * 2)
        self.density_noise = config.density_noise
        self.rgb_padding = config.rgb_padding
        self.resample_padding = config.resample_padding
        self.density_bias = config.density_bias
        self.hidden = config.hidden
        self.wavelength = config.wavelength
        self.total_num_rays = config.total_num_rays
        self.rays_per_chunk = config.rays_per_chunk
        self.return_raw = return_raw
        self.density_activation = nn.Softplus()
        self.num_ray_iters = int(self.total_num_rays // self.rays_per_chunk)
        self.cube_dim = int(np.sqrt(self.rays_per_chunk))
        self.loss_function = SARNeRFLoss(config.coarse_weight_decay)
        self.positional_encoding = PositionalEncoding(config.min_deg, config.max_deg)
        self._build_trunk(config)
        self.final_density = nn.Sequential(nn.LayerNorm(config.hidden), nn.Linear(config.hidden, 1))
        input_shape = config.hidden
        if self.use_viewdirs:
            input_shape = config.num_samples
            self.rgb_net0 = nn.Sequential(nn.Linear(config.hidden, config.hidden))
            self.viewdirs_encoding = PositionalEncoding(config.viewdirs_min_deg, config.viewdirs_max_deg)
            self.rgb_net1 = nn.Sequential(nn.Linear(config.hidden + self.rgb_input, config.num_samples), nn.ReLU(True))
        self.final_rgb = nn.Sequential(nn.Linear(input_shape, 3), nn.Sigmoid())
        self.final_angle = nn.Sequential(nn.Linear(input_shape, 2), nn.Sigmoid())
        _xavier_init(self)
        self._packed = None

    def _query(self, enc, view_feat, R, S):
        # Row i of the MLP input belongs to ray i // S, but upstream tiles the [1,R,V] view features
        # along the *sample* axis (model.py:351-354), so row i receives the features of ray i mod R.
        view_rows = None
        if view_feat is not None:
            idx = torch.arange(R * S, device=enc.device) % R
            view_rows = view_feat[idx]
        return mlp.mlp_forward_fp32(self._param_dict(), enc, view_rows, sar=True)

    def forward(self, ray_o, pan, tilt, mfilt, _radii, _near, _far, nsam, mpp, az_el=None, t_rand=None,
                u_rand=None, rank=0, world_size=1):
        """One pulse (B = 1), as the reference supports (model.py:305-387).  `pan`/`tilt` are unused
        upstream too.  `az_el[c] = (az, el)` optionally supplies chunk c's beam offsets (already scaled
        by the beam sigmas); otherwise they are drawn on the CPU generator like upstream.  `t_rand[c]`
        and `u_rand[c]` optionally supply the sampling draws of chunk c.
        With `world_size > 1` (torch.distributed initialised) the ray chunks are dealt round-robin to the ranks
        and the per-level range profiles are SUM-all-reduced ahead of the matched filter: the pulses are those of
        the one-device loop; distance / acc are those of this rank's last chunk.
        A ray whose expected distance lies beyond the last pulse bin raises IndexError, as the reference's indexed
        write does (utils.py:379); the check reads one status word back per call (one stream synchronisation) and
        `config.check_range = False` switches it off (such rays are then dropped from the profile).
        -> pulses [L,1,nsam,2], distance [L,1,R], acc [L,1,R]"""
        cfg = self.config
        dev = self.device
        if ray_o.shape[0] != 1:
            raise ValueError('SARNeRF.forward handles one pulse per call (B = 1), like the reference')
        L, S, R = self.num_levels, self.num_samples, self.rays_per_chunk
        f32 = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev).reshape(-1)
        near_s, far_s, rad_s, mpp_s = f32(_near)[0], f32(_far)[0], f32(_radii)[0], f32(mpp)[0]
        near = near_s.expand(R).contiguous()
        far = far_s.expand(R).contiguous()
        radii = rad_s.expand(R).contiguous()
        pulse_bins = torch.arange(nsam, device=dev) * mpp_s + near_s                           # model.py:316
        origins = ray_o.to(dev, torch.float32).reshape(1, 3).expand(R, 3).contiguous()
        comp = [torch.zeros(nsam, 2, device=dev) for _ in range(L)]
        distances, accs = [None] * L, [None] * L
        raw_rgb = raw_density = None
        statuses = []
        for c in range(self.num_ray_iters):
            if az_el is not None:
                az, el = az_el[c]
            else:                                                                              # model.py:319-320
                az = torch.randn(R) * SAR_AZ_SIGMA
                el = torch.randn(R) * SAR_EL_SIGMA
            if c % world_size != rank:          # another rank's chunk (its beam offsets were drawn above, so that
                continue                        # every rank walks the same CPU generator sequence)
            ray_d, ray_p = ops.sar_raygen(az.to(dev), el.to(dev))
            bf16 = self._precision() == 'bf16'
            if bf16:
                from . import mlp_bf16
            if not self.use_viewdirs:
                view_feat = None
            elif bf16:
                view_feat = mlp_bf16.view_frag(ray_d, cfg.viewdirs_min_deg, cfg.viewdirs_max_deg)
            else:
                view_feat = ops.pe_encode(ray_d, cfg.viewdirs_min_deg, cfg.viewdirs_max_deg, append_input=True)
            t_vals = weights = None
            for l in range(L):
                with torch.no_grad():
                    if l == 0:
                        tr = None
                        if self.randomized:
                            tr = t_rand[c] if t_rand is not None else torch.rand(R, S + 1, device=dev)
                        t_vals = ops.stratified_t(near, far, S, tr, False)
                    else:
                        ur = None
                        if self.randomized:
                            ur = (u_rand[c] if u_rand is not None else
                                  torch.empty(R, S + 1, device=dev).uniform_(0, 1 / (S + 1) - float(torch.finfo(torch.float32).eps)))
                        t_vals = ops.resample(t_vals, weights, S + 1, self.resample_padding, ur, blur=True)
                    if not bf16:
                        mean, cov = ops.cast_rays(t_vals, origins, ray_d, radii, self.ray_shape)
                        enc = ops.ipe_encode(mean, cov, cfg.min_deg, cfg.max_deg).reshape(R * S, -1)
                if bf16:        # cast + encoding fused into the operand fragments of the MFMA chain
                    raw_density, raw_rgb, angle = mlp_bf16.query_sar(self, t_vals, origins, ray_d, radii, view_feat, R, S)
                else:
                    raw_density, raw_rgb, angle = self._query(enc, view_feat, R, S)
                if self.randomized and self.density_noise:
                    raw_density = raw_density + self.density_noise * torch.rand(raw_density.shape, device=dev)
                pulse, distance, acc, weights, alpha, _, status = ops.scatter(
                    raw_rgb.reshape(R, S, 3), angle.reshape(R, S, 2), raw_density.reshape(R, S), ray_p, t_vals, ray_d,
                    pulse_bins, self.wavelength, act=(True, self.rgb_padding, self.density_bias))
                comp[l] = comp[l] + pulse
                distances[l] = distance
                accs[l] = acc
                statuses.append(status)
        if statuses and getattr(cfg, 'check_range', True) and int(torch.cat(statuses).min()) != 0:
            raise IndexError('range bin index out of bounds: a ray distance lies beyond the last pulse bin '
                             '(far > near + (nsam - 1) * mpp)')
        mf = mfilt.to(dev)
        outs = []
        if world_size > 1:
            from .train import allreduce_sum_shared_loss
            comp = [allreduce_sum_shared_loss(x, world_size) for x in comp]
            for l in range(L):                  # a rank that was dealt no chunk still returns tensors
                if distances[l] is None:
                    distances[l], accs[l] = torch.zeros(R, device=dev), torch.zeros(R, device=dev)
        for l in range(L):                                                                     # model.py:377-380
            z = torch.view_as_complex(comp[l][None].contiguous())
            z = torch.fft.ifft(mf * torch.fft.fft(z, mf.shape[-1], dim=-1), dim=-1)[..., :nsam]
            z = z / torch.std(z, dim=-1)[..., None]
            outs.append(torch.view_as_real(z))
        out = (torch.stack(outs), torch.stack(distances)[:, None], torch.stack(accs)[:, None])
        if self.return_raw:
            with torch.no_grad():
                rgb = raw_rgb.reshape(1, R, S, 3) * (1 + 2 * self.rgb_padding) - self.rgb_padding
                density = torch.nn.functional.softplus(raw_density.reshape(1, R, S, 1) + self.density_bias)
                raws = torch.cat((rgb, density), -1).cpu()
            return out + (raws,)
        return out

    def train_val_get(self, batch, batch_idx, kind='train'):
        ray_o, pan, tilt, mfilt, radii, near, far, mpp, pulse_data = batch
        rgb = self.forward(ray_o, pan, tilt, mfilt, radii, near, far, pulse_data.shape[1], mpp)[0]
        train_loss, psnrs = self.loss_function(rgb, pulse_data)
        self.log_dict({f'{kind}_loss': train_loss, 'coarse_psnr': torch.mean(psnrs[:-1]), 'fine_psnr': psnrs[-1],
                       'avg_psnr': torch.mean(psnrs), 'LR': self.lr_schedulers().get_last_lr()[0]}, on_epoch=True,
                      prog_bar=True, rank_zero_only=True)
        return train_loss
