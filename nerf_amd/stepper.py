"""One mip-NeRF training step as a fixed sequence of kernel launches on persistent buffers, without autograd.

What the reference does in `training_step` (model.py:185-191: forward over the levels, model.py:100-151; the
masked MSE, util_modules.py:58-71; `manual_backward`; optimizer step) is here one hand-ordered pass:

    per level:  fenceposts (stratified / resampled)  ->  cast + integrated encoding as operand fragments
                ->  forward chain (saves what the backward pass reads)
                ->  compositing + loss + compositing backward in one kernel
    then:       ONE backward chain over the samples of all levels
                ->  ONE weight-gradient launch per layer over the samples of all levels, adding into the flat
                    gradient bucket  ->  (all-reduce)  ->  optimizer  ->  weight repack

The levels' samples lie behind one another in every saved buffer (level l starts at a whole 256-sample workgroup
tile), so the passes that do not depend on the level order -- the input-gradient chain and the weight gradients --
run once per step instead of once per level.  Every buffer is allocated once for a given batch geometry.
"""
import ctypes
from ctypes import c_void_p

import torch

from . import mlp_bf16
from ._lib import NerfActivation, check, lib, ptr, stream

_EPS32 = float(torch.finfo(torch.float32).eps)


def _grads_ready(model):
    """Every trainable tensor of the network takes part and owns a contiguous fp32 `.grad` to add into (a frozen
    parameter, or one without a gradient buffer, sends the step through autograd instead)."""
    for k, p in model.named_parameters():
        if k.endswith('scales'):
            continue
        if not p.requires_grad or p.grad is None or p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
            return False
    return True


def supported(model):
    """The fused step covers the bf16 network with every parameter trainable; `return_raw` goes through autograd."""
    cfg = model.config
    return (getattr(cfg, 'mlp_precision', 'fp32') == 'bf16' and not getattr(model, 'return_raw', False)
            and hasattr(model, 'white_bkgd') and _grads_ready(model))


class NerfStepper:
    """Persistent buffers + launch sequence for `n_rays` rays per step at the model's level sample counts."""

    def __init__(self, model, n_rays):
        self.model, self.n = model, int(n_rays)
        cfg = model.config
        self.levels = model._level_samples()
        self.dev = dev = model.device
        pk = mlp_bf16.packed_for(model, need_bwd=True)
        n = self.n
        self.nblk = [mlp_bf16.n_blocks(n * S) for S in self.levels]
        self.blk0 = [sum(self.nblk[:l]) for l in range(len(self.levels))]
        self.nblk_total = sum(self.nblk)
        u8 = lambda nbytes: torch.empty(nbytes, dtype=torch.uint8, device=dev)
        f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        self.act_bytes = mlp_bf16.act_block_bytes(pk)
        self.dy_bytes = mlp_bf16.dy_block_bytes(pk)
        self.acts = [u8(self.nblk_total * b) if b is not None else None for b in self.act_bytes]
        self.dys = [u8(self.nblk_total * b) if b is not None else None for b in self.dy_bytes]
        # per-sample fp32 arrays, level after level at the same block offsets; the padding samples of a level's
        # last tile are never written, so they are zeroed once: a zero gradient there keeps them out of every sum
        m_all = 64 * self.nblk_total
        self.raw_density, self.raw_rgb = torch.zeros(m_all, device=dev), torch.zeros(m_all, 3, device=dev)
        self.g_density, self.g_rgb = torch.zeros(m_all, device=dev), torch.zeros(m_all, 3, device=dev)
        dks = pk.DKS
        self.enc = [u8(nb * dks * 2 * 1024) for nb in self.nblk]
        self.view = u8(n * pk.VKS * 2 * 16) if pk.view else None
        self.t_vals = [f32(n, S + 1) for S in self.levels]
        self.rand = [f32(n, S + 1) for S in self.levels]
        self.weights = [f32(n, S) for S in self.levels]
        L = len(self.levels)
        self.comp, self.dist, self.acc = f32(L, n, 3), f32(L, n), f32(L, n)
        self.ray_sq = f32(L, n)
        self.inv_mask_sum = f32(1)

    # ---- helpers ---------------------------------------------------------------------------
    def _level(self, bufs, per_block, l):
        lo, hi = self.blk0[l], self.blk0[l] + self.nblk[l]
        return [b[lo * pb:hi * pb] if b is not None else None for b, pb in zip(bufs, per_block)]

    def _samples(self, t, l):
        S = self.levels[l]
        lo = 64 * self.blk0[l]
        return t[lo:lo + self.n * S]

    def step(self, rays, pixels, global_mask_sum=None, t_rand=None, u_rand=None, noise=None, after_trunk=None):
        """Forward + loss + backward of one batch; the weight gradients are ADDED into the parameters' `.grad`
        (views of the caller's gradient bucket).  -> loss (device scalar), comp_rgb [L,N,3] (this step's buffers).
        `t_rand`, `u_rand`, `noise` as in MipNeRF.forward: pre-drawn random tensors in place of the device draws.
        `after_trunk`: see mlp_bf16.mlp_weight_grads (where a multi-rank step starts reducing the trunk gradients)."""
        model, n, dev = self.model, self.n, self.dev
        cfg = model.config
        assert rays.origins.shape[0] == n
        pk = mlp_bf16.packed_for(model, need_bwd=True)
        L_ = lib()
        st = stream()
        f = lambda t, *shape: t.detach().to(torch.float32).reshape(*shape).contiguous()
        o, d, vd = f(rays.origins, n, 3), f(rays.directions, n, 3), f(rays.viewdirs, n, 3)
        radii, near, far = f(rays.radii, n), f(rays.near, n), f(rays.far, n)
        lossmult, target = f(rays.lossmult, n), f(pixels, n, 3)
        if global_mask_sum is None:
            global_mask_sum = lossmult.sum()
        torch.reciprocal(torch.as_tensor(global_mask_sum, dtype=torch.float32, device=dev).reshape(1), out=self.inv_mask_sum)
        if pk.view:
            check(L_.nerf_view_frag(ptr(vd), n, cfg.viewdirs_min_deg, cfg.viewdirs_max_deg, c_void_p(self.view.data_ptr()), st))
        act = ctypes.byref(NerfActivation(1, float(model.rgb_padding), float(model.density_bias)))
        cwd = float(model.loss_function.coarse_weight_decay)
        nl = len(self.levels)
        shape = mlp_bf16.RAY_SHAPES[model.ray_shape]
        for l, S in enumerate(self.levels):
            t = self.t_vals[l]
            if l == 0:                                                          # utils.py:214-228
                r = None
                if model.randomized:
                    r = f(t_rand, n, S + 1) if t_rand is not None else torch.rand(n, S + 1, out=self.rand[0])
                check(L_.nerf_stratified_t(ptr(near), ptr(far), ptr(r), n, S, 0, ptr(t), st))
            else:                                                               # utils.py:252-264, 16-63
                r = None
                if model.randomized:
                    r = (f(u_rand[l], n, S + 1) if u_rand is not None else
                         self.rand[l].uniform_(0, 1 / (S + 1) - _EPS32))
                S0 = self.levels[l - 1]
                check(L_.nerf_resample(ptr(self.t_vals[l - 1]), ptr(self.weights[l - 1]), ptr(r), n, S0, S + 1,
                                       float(model.resample_padding), 1, ptr(t), None, None, st))
            check(L_.nerf_encode_frag(ptr(t), ptr(o), ptr(d), ptr(radii), n, S, shape, cfg.min_deg, cfg.max_deg,
                                      c_void_p(self.enc[l].data_ptr()), st))
            raw_d, raw_c = self._samples(self.raw_density, l), self._samples(self.raw_rgb, l)
            mlp_bf16.mlp_forward(pk, self.enc[l], self.view, n * S, S, n, acts=self._level(self.acts, self.act_bytes, l),
                                 raw=(raw_d, raw_c))
            if model.randomized and model.density_noise:                        # model.py:135-136
                # U(0,1) noise on the raw density between the MLP and the compositing; it shifts the point at which the
                # density activation is evaluated and leaves d(noisy) / d(raw) = 1, so the buffer is perturbed in place
                # (the NeRF backward chain does not read the raw density again)
                nz = f(noise[l], n * S) if noise is not None else torch.rand(n * S, device=dev)
                raw_d.add_(nz * float(model.density_noise))      # (product rounded, then the sum: as the reference's +=)
            check(L_.nerf_composite_train(ptr(raw_c), ptr(raw_d), ptr(t), ptr(d), ptr(target), ptr(lossmult),
                                          ptr(self.inv_mask_sum), 1.0 if l == nl - 1 else cwd, n, S,
                                          int(bool(model.white_bkgd)), act, ptr(self.comp[l]), ptr(self.dist[l]),
                                          ptr(self.acc[l]), ptr(self.weights[l]), ptr(self.ray_sq[l]),
                                          ptr(self._samples(self.g_rgb, l)), ptr(self._samples(self.g_density, l)), st))
        m_all = 64 * self.nblk_total
        n_real = n * sum(self.levels)
        mlp_bf16.mlp_backward(pk, self.acts, m_all, self.g_density, self.g_rgb, self.raw_rgb, dys=self.dys, n_real=n_real)
        grads = {k: q.grad for k, q in model.named_parameters() if q.requires_grad}
        mlp_bf16.mlp_weight_grads(pk, self.acts, self.dys, m_all, grads, after_trunk)
        sq = self.ray_sq.sum(dim=1) * self.inv_mask_sum
        return cwd * sq[:-1].sum() + sq[-1], self.comp


def sar_supported(model):
    """The fused SAR step covers the bf16 network with every parameter trainable."""
    cfg = model.config
    return (getattr(cfg, 'mlp_precision', 'fp32') == 'bf16' and not getattr(model, 'return_raw', False)
            and hasattr(model, 'rays_per_chunk') and _grads_ready(model))


class SarStepper:
    """One SAR pulse (model.py:305-387 + loss + backward) as a fixed launch sequence on persistent buffers.

    The forward runs the ray chunks and sampling levels one after another, each pass (chunk, level) into its own
    slice of every saved buffer (whole 256-sample tiles).  The matched filter and the loss (model.py:377-380,
    util_modules.py:74-89) go through torch's autograd with the accumulated range profiles as leaves; their gradient
    is scattered back per pass (`nerf_scatter_bwd`), and then ONE backward chain and ONE weight-gradient launch per
    layer cover the samples of all passes, adding into the parameters' `.grad`.

    With `world_size > 1` the ray chunks are dealt round-robin to the ranks as `SARNeRF.forward` deals them (the
    reference adds the chunks' profiles, model.py:371, so whole chunks shard without changing the result): this
    rank renders chunks `rank, rank + world_size, ...`, the per-level profiles are SUM-all-reduced ahead of the
    matched filter, every rank evaluates the same loss and back-propagates it into its own chunks; the caller
    SUM-all-reduces the weight gradients."""

    def __init__(self, model, rank=0, world_size=1):
        self.model = model
        self.dev = dev = model.device
        self.rank, self.world = int(rank), int(world_size)
        pk = mlp_bf16.packed_for(model, need_bwd=True)
        self.R, self.S, self.L = model.rays_per_chunk, model.num_samples, model.num_levels
        self.chunks = model.num_ray_iters
        self.own = [c for c in range(self.chunks) if c % self.world == self.rank]
        self.passes = len(self.own) * self.L
        nb = mlp_bf16.n_blocks(self.R * self.S)
        self.nblk, self.nblk_total = nb, nb * self.passes
        u8 = lambda nbytes: torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.act_bytes = mlp_bf16.act_block_bytes(pk)
        self.dy_bytes = mlp_bf16.dy_block_bytes(pk)
        self.acts = [u8(self.nblk_total * b) if b is not None else None for b in self.act_bytes]
        self.dys = [u8(self.nblk_total * b) if b is not None else None for b in self.dy_bytes]
        m_all = 64 * self.nblk_total
        z = lambda *shape: torch.zeros(*shape, device=dev)      # (padding samples of a pass stay zero: no gradient)
        self.raw_density, self.raw_rgb, self.raw_angle, self.ln_stats = z(m_all), z(m_all, 3), z(m_all, 2), z(m_all, 2)
        self.g_density, self.g_rgb, self.g_angle = z(m_all), z(m_all, 3), z(m_all, 2)
        # scratch of the weight-gradient pass (sums the head gradients are assembled from), one buffer, one fill per step
        hw = pk.W if pk.view else pk.H
        shapes = {'_dens.weight': (3, pk.H), '_dens.bias': (3,), '_heads.weight': (5, hw), '_heads.bias': (5,)}
        shapes.update({'_g98.weight': (pk.W, pk.H), '_g98.bias': (pk.W,)} if pk.view else
                      {'_hd.weight': (8, pk.H), '_hd.bias': (8,)})
        sizes = {k: -(-int(torch.Size(v).numel()) // 64) * 64 for k, v in shapes.items()}      # 256-byte aligned pieces
        self.scratch = z(sum(sizes.values()))
        self.scratch_grads, off = {}, 0
        for k, shp in shapes.items():
            self.scratch_grads[k] = self.scratch[off:off + int(torch.Size(shp).numel())].view(shp)
            off += sizes[k]

    def _slice(self, bufs, per_block, q):
        lo, hi = q * self.nblk, (q + 1) * self.nblk
        return [b[lo * pb:hi * pb] if b is not None else None for b, pb in zip(bufs, per_block)]

    def step(self, ray_o, mfilt, _radii, _near, _far, nsam, mpp, target, az_el=None, t_rand=None, u_rand=None):
        """-> loss (device scalar).  Arguments as SARNeRF.forward; the weight gradients are ADDED into `.grad`."""
        from . import ops
        from .model import SAR_AZ_SIGMA, SAR_EL_SIGMA
        model, dev, R, S, L = self.model, self.dev, self.R, self.S, self.L
        cfg = model.config
        pk = mlp_bf16.packed_for(model, need_bwd=True)
        L_ = lib()
        st = stream()
        f32 = lambda x: torch.as_tensor(x, dtype=torch.float32, device=dev).reshape(-1)
        near_s, far_s, rad_s, mpp_s = f32(_near)[0], f32(_far)[0], f32(_radii)[0], f32(mpp)[0]
        near, far, radii = near_s.expand(R).contiguous(), far_s.expand(R).contiguous(), rad_s.expand(R).contiguous()
        pulse_bins = (torch.arange(nsam, device=dev) * mpp_s + near_s).contiguous()                # model.py:316
        origins = ray_o.to(dev, torch.float32).reshape(1, 3).expand(R, 3).contiguous()
        act = ctypes.byref(NerfActivation(1, float(model.rgb_padding), float(model.density_bias)))
        wl = float(model.wavelength)
        M = R * S
        comp = [torch.zeros(nsam, 2, device=dev) for _ in range(L)]
        saved, statuses = [], []
        i32 = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.int32)
        for c in range(model.num_ray_iters):
            if az_el is not None:
                az, el = az_el[c]
            else:                                                                              # model.py:319-320
                az, el = torch.randn(R) * SAR_AZ_SIGMA, torch.randn(R) * SAR_EL_SIGMA
            if c % self.world != self.rank:     # another rank's chunk (its beam offsets were drawn above, so that every
                continue                        # rank walks the same host generator sequence)
            ray_d, ray_p = ops.sar_raygen(az.to(dev), el.to(dev))
            viewf = mlp_bf16.view_frag(ray_d, cfg.viewdirs_min_deg, cfg.viewdirs_max_deg) if pk.view else None
            t_vals = weights = None
            for l in range(L):
                q = (c // self.world) * L + l
                if l == 0:
                    tr = None
                    if model.randomized:
                        tr = t_rand[c] if t_rand is not None else torch.rand(R, S + 1, device=dev)
                    t_vals = ops.stratified_t(near, far, S, tr, False)
                else:
                    ur = None
                    if model.randomized:
                        ur = (u_rand[c] if u_rand is not None else
                              torch.empty(R, S + 1, device=dev).uniform_(0, 1 / (S + 1) - _EPS32))
                    t_vals = ops.resample(t_vals, weights, S + 1, model.resample_padding, ur, blur=True)
                encf = mlp_bf16.encode_frag(t_vals, origins, ray_d, radii, model.ray_shape, cfg.min_deg, cfg.max_deg)
                lo = 64 * q * self.nblk
                raw = tuple(t[lo:lo + M] for t in (self.raw_density, self.raw_rgb, self.raw_angle, self.ln_stats))
                mlp_bf16.mlp_forward(pk, encf, viewf, M, R, R, save=True, acts=self._slice(self.acts, self.act_bytes, q), raw=raw)
                dens = raw[0]
                if model.randomized and model.density_noise:                                   # model.py:362-363
                    # the compositing sees the perturbed density; the LayerNorm backward needs the network's own output,
                    # so the perturbed copy is a separate tensor (kept for nerf_scatter_bwd)
                    dens = torch.rand(M, device=dev).mul_(float(model.density_noise)).add_(raw[0])
                pulse, dist, acc = torch.empty(nsam, 2, device=dev), torch.empty(R, device=dev), torch.empty(R, device=dev)
                weights, alpha = torch.empty(R, S, device=dev), torch.empty(R, S, device=dev)
                ray_ret, ray_bin, comp5 = torch.empty(R, 2, device=dev), i32(R), torch.empty(R, 5, device=dev)
                winner, status = i32(nsam), i32(1)
                check(L_.nerf_scatter_fwd(ptr(raw[1]), ptr(raw[2]), ptr(dens), ptr(ray_p), ptr(t_vals), ptr(ray_d),
                                          ptr(pulse_bins), R, S, nsam, wl, act, ptr(pulse), ptr(dist), ptr(acc),
                                          ptr(weights), ptr(alpha), ptr(ray_ret), ptr(ray_bin, torch.int32), ptr(comp5),
                                          ptr(winner, torch.int32), ptr(status, torch.int32), st))
                comp[l] += pulse
                saved.append((ray_p, t_vals, ray_d, ray_bin, comp5, dist, dens if dens is not raw[0] else None))
                statuses.append(status)
        if statuses and getattr(cfg, 'check_range', True) and int(torch.cat(statuses).min()) != 0:
            raise IndexError('range bin index out of bounds: a ray distance lies beyond the last pulse bin '
                             '(far > near + (nsam - 1) * mpp)')
        if self.world > 1:                      # the one exchange step of the pulse: nsam x 2 floats per level
            from .train import _all_reduce_sum
            for x in comp:
                _all_reduce_sum(x)
        # matched filter, per-pulse normalisation, loss and the gradient of the range profiles: one launch when the
        # filter's length is a power of two up to 8192, else torch.fft (rocFFT) and autograd with the profiles as leaves
        mf = mfilt.to(dev)
        cwd = float(model.loss_function.coarse_weight_decay)
        if ops.sar_pulse_supported(nsam, mf.shape[-1]) and mf.numel() == mf.shape[-1]:
            loss, _, _, g_comp = ops.sar_pulse_loss(torch.stack(comp), mf, target.to(dev), cwd)
        else:
            leaves = [x.requires_grad_(True) for x in comp]
            outs = []
            for l in range(L):                                                                 # model.py:377-380
                zc = torch.view_as_complex(leaves[l][None].contiguous())
                zc = torch.fft.ifft(mf * torch.fft.fft(zc, mf.shape[-1], dim=-1), dim=-1)[..., :nsam]
                outs.append(torch.view_as_real(zc / torch.std(zc, dim=-1)[..., None]))
            loss, _ = model.loss_function.level_losses(torch.stack(outs), target.to(dev))
            g_comp = torch.autograd.grad(loss, leaves)
        for q, (ray_p, t_vals, ray_d, ray_bin, comp5, dist, dens) in enumerate(saved):
            lo = 64 * q * self.nblk
            sl = lambda t: t[lo:lo + M]
            check(L_.nerf_scatter_bwd(ptr(sl(self.raw_rgb)), ptr(sl(self.raw_angle)),
                                      ptr(dens if dens is not None else sl(self.raw_density)), ptr(ray_p),
                                      ptr(t_vals), ptr(ray_d), ptr(g_comp[q % L].contiguous()), ptr(ray_bin, torch.int32),
                                      ptr(comp5), ptr(dist), None, None, R, S, nsam, wl, act, ptr(sl(self.g_rgb)),
                                      ptr(sl(self.g_angle)), ptr(sl(self.g_density)), st))
        m_all = 64 * self.nblk_total
        if m_all == 0:                          # a rank that was dealt no chunk has nothing to differentiate
            return loss.detach()
        mlp_bf16.mlp_backward(pk, self.acts, m_all, self.g_density, self.g_rgb, self.raw_rgb, self.g_angle, self.raw_density,
                              self.raw_angle, self.ln_stats, dys=self.dys, n_real=M * self.passes)
        P = {k: v.detach() for k, v in model.named_parameters()}
        grads = {k: q.grad for k, q in model.named_parameters() if q.requires_grad}
        self.scratch.zero_()
        grads.update(self.scratch_grads)
        mlp_bf16.mlp_weight_grads(pk, self.acts, self.dys, m_all, grads)
        mlp_bf16.add_sar_head_gradients(P, grads)
        return loss.detach()


def sar_stepper_for(model, rank=0, world_size=1):
    s = getattr(model, '_sar_stepper', None)
    if (s is None or s.dev != model.device or (s.R, s.S, s.L) != (model.rays_per_chunk, model.num_samples, model.num_levels)
            or s.chunks != model.num_ray_iters or (s.rank, s.world) != (rank, world_size)):
        model._sar_stepper = s = None   # release the old buffers before the new ones are allocated
        s = SarStepper(model, rank, world_size)
        model._sar_stepper = s
    return s


def stepper_for(model, n_rays):
    s = getattr(model, '_stepper', None)
    if s is None or s.n != n_rays or s.levels != model._level_samples() or s.dev != model.device:
        model._stepper = s = None       # release the old buffers (no name may still hold them) before the new ones are allocated
        s = NerfStepper(model, n_rays)
        model._stepper = s
    return s
