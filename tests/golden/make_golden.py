"""Generate the golden fixtures under tests/golden/ by running the upstream reference.

Runs ONLY in the build container (it imports the reference's `utils`, `util_modules`,
`model` and `dataloader` modules from /root/reference behind empty stub modules for the
packages that are absent here and that the functions called below never touch:
`pytorch_lightning`, `mcubes`, `sdrparse`, `simulib`, `cv2`).  The
fixtures (inputs, pre-drawn random tensors, expected outputs and gradients) are plain
`.npz` data; nothing of the reference's source travels with them.

    python tests/golden/make_golden.py

Weights are NOT stored: fixtures name a seed and the test rebuilds them with
`oracle.nerf_oracle.init_params(cfg, seed=...)`, the same call used here before
`load_state_dict`.
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
_ARGS = sys.argv[1:]
sys.argv = ['x']
sys.path.insert(0, '/root/reference')

pl = types.ModuleType('pytorch_lightning')


class LightningModule(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    @property
    def device(self):
        p = next(self.parameters(), None)
        return p.device if p is not None else torch.device('cpu')


pl.LightningModule = LightningModule
pl.LightningDataModule = type('LightningDataModule', (object,), {'__init__': lambda self, *a, **k: None})
sys.modules['pytorch_lightning'] = pl
# names dataloader.py imports at module level (dataloader.py:3-8); the ray generators never call them
for _name, _attrs in (('mcubes', ()), ('sdrparse', ('load',)), ('simulib', ()), ('simulib.platform_helper', ('SDRPlatform',)),
                      ('simulib.simulation_functions', ('azelToVec', 'findPowerOf2')), ('cv2', ())):
    _m = types.ModuleType(_name)
    for _a in _attrs:
        setattr(_m, _a, None)
    sys.modules[_name] = _m

import utils as R_utils                    # noqa: E402  (reference)
import util_modules as R_mods              # noqa: E402
import model as R_model                    # noqa: E402
import dataloader as R_data                # noqa: E402
from oracle import nerf_oracle as O        # noqa: E402  (only for init_params / config helpers)

torch.set_num_threads(1)                   # makes the reference's duplicate-bin scatter deterministic


def np_(x):
    return x.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (np_(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024))


def make_rays(n, seed, normalize=True, near=2., far=6., radius=0.005):
    g = torch.Generator().manual_seed(seed)
    o = 0.1 * torch.randn(n, 3, generator=g) + torch.tensor([0., 0., 4.])
    d = torch.randn(n, 3, generator=g)
    if normalize:
        d = d / d.norm(dim=-1, keepdim=True)
    else:
        d = d / d.norm(dim=-1, keepdim=True) * (1.0 + 0.2 * torch.rand(n, 1, generator=g))
    v = d / d.norm(dim=-1, keepdim=True)
    return dict(origins=o, directions=d, viewdirs=v, radii=torch.full((n, 1), radius),
                lossmult=torch.ones(n, 1), near=torch.full((n, 1), near), far=torch.full((n, 1), far))


def nerf_cfg(**kw):
    c = dict(use_viewdirs=True, randomized=True, ray_shape='cone', white_bkgd=False, num_levels=2,
             num_samples=64, min_deg=0, max_deg=10, viewdirs_min_deg=0, viewdirs_max_deg=4,
             density_noise=0.0, rgb_padding=0.001, resample_padding=0.01, density_bias=-1.0, hidden=256,
             coarse_weight_decay=0.1, lr_init=1e-3, lr_final=5e-5, max_steps=2500, lr_delay_steps=2500,
             lr_delay_mult=0.1, weight_decay=1e-5)
    c.update(kw)
    return SimpleNamespace(**c)


# ------------------------------------------------------------------------------------------
def gen_sample():
    out = {}
    r = make_rays(12, 1, normalize=False)
    S = 9
    for shape in ('cone', 'cylinder'):
        for rnd in (0, 1):
            for lin in (0, 1):
                torch.manual_seed(100 + rnd)
                t_rand = torch.rand(1, 12, S + 1)
                torch.manual_seed(100 + rnd)
                t, (m, c) = R_utils.sample_along_rays(r['origins'][None], r['directions'][None], r['radii'][None],
                                                      S, r['near'][None], r['far'][None], bool(rnd), bool(lin), shape)
                key = f'{shape}_{rnd}_{lin}'
                out['t_' + key] = t[0]
                out['mean_' + key] = m[0]
                out['cov_' + key] = c[0]
                out['trand_' + key] = t_rand[0]
    save('sample', S=S, **{k: v for k, v in r.items()}, **out)


def gen_posenc():
    g = torch.Generator().manual_seed(2)
    mean = torch.randn(7, 5, 3, generator=g) * 2 + 1
    cov = torch.rand(7, 5, 3, generator=g) * 1e-3
    out = dict(mean=mean, cov=cov)
    for L in (4, 10, 16):
        pe = R_mods.PositionalEncoding(0, L)
        out[f'ipe_{L}'], out[f'ipe_var_{L}'] = pe(mean, cov)
    pe = R_mods.PositionalEncoding(0, 4)
    v = torch.randn(9, 3, generator=g)
    v = v / v.norm(dim=-1, keepdim=True)
    out['view'] = v
    out['pe_4'] = pe(v)
    pe2 = R_mods.PositionalEncoding(1, 5)
    out['ipe_1_5'], out['ipe_var_1_5'] = pe2(mean, cov)
    save('posenc', **out)


def gen_resample():
    out = {}
    g = torch.Generator().manual_seed(3)
    for tag, n, s0, n_out in (('65_65', 6, 64, 65), ('65_129', 6, 64, 129), ('181_181', 3, 180, 181), ('9_5', 4, 8, 5)):
        bins = torch.sort(torch.rand(1, n, s0 + 1, generator=g) * 4 + 2, dim=-1)[0]
        w = torch.rand(1, n, s0, generator=g) ** 4
        w[0, 0] = 0.0                                   # all-zero row -> eps padding branch
        w[0, 1, : s0 // 2] = 0.0                        # zero-pdf plateau -> tied cdf entries
        w[0, 2] = w[0, 2] * 1e-7                        # tiny sum -> partial eps padding
        for rnd in (0, 1):
            torch.manual_seed(30 + rnd)
            u_rand = torch.empty(1, n, n_out).uniform_(to=(1 / n_out - torch.finfo(torch.float32).eps))
            torch.manual_seed(30 + rnd)
            samples = R_utils.sorted_piecewise_constant_pdf(bins, w.clone(), n_out, bool(rnd))
            out[f'spdf_{tag}_{rnd}'] = samples[0]
            out[f'urand_{tag}_{rnd}'] = u_rand[0]
        out[f'bins_{tag}'] = bins[0]
        out[f'w_{tag}'] = w[0]
    # resample_along_rays end to end (blur + padding + cast), 17 fenceposts
    r = make_rays(5, 4)
    S = 16
    t = torch.sort(torch.rand(1, 5, S + 1, generator=g) * 4 + 2, dim=-1)[0]
    w = torch.rand(1, 5, S, generator=g) ** 3
    for rnd in (0, 1):
        torch.manual_seed(40 + rnd)
        u_rand = torch.empty(1, 5, S + 1).uniform_(to=(1 / (S + 1) - torch.finfo(torch.float32).eps))
        torch.manual_seed(40 + rnd)
        nt, (m, c) = R_utils.resample_along_rays(r['origins'][None], r['directions'][None], r['radii'][None], t, w,
                                                 bool(rnd), True, 0.01, 'cone')
        out[f'rar_t_{rnd}'] = nt[0]
        out[f'rar_mean_{rnd}'] = m[0]
        out[f'rar_cov_{rnd}'] = c[0]
        out[f'rar_urand_{rnd}'] = u_rand[0]
    out['rar_tin'] = t[0]
    out['rar_w'] = w[0]
    for k, v in r.items():
        out['rar_' + k] = v
    save('resample', **out)


def gen_composite():
    g = torch.Generator().manual_seed(5)
    n, S = 10, 24
    r = make_rays(n, 6, normalize=False)
    t = torch.sort(torch.rand(n, S + 1, generator=g) * 4 + 2, dim=-1)[0]
    rgb = torch.rand(n, S, 3, generator=g)
    density = torch.rand(n, S, 1, generator=g) * 3
    density[0] = 0.0                                    # empty ray: acc = 0, distance nan -> 0 -> clamped to t0
    density[1] = 80.0                                   # opaque at the first sample
    out = dict(t_vals=t, rgb=rgb, density=density, dirs=r['directions'])
    for wb in (0, 1):
        rgb_ = rgb.clone().requires_grad_(True)
        den_ = density.clone().requires_grad_(True)
        comp, dist, acc, w, alpha = R_utils.volumetric_rendering(rgb_, den_, t, r['directions'], bool(wb))
        gc = torch.randn(n, 3, generator=g)
        ga = torch.randn(n, generator=g)
        gd = torch.randn(n, generator=g)
        loss = (comp * gc).sum() + (acc * ga).sum() + (dist * gd).sum()
        d_rgb, d_den = torch.autograd.grad(loss, (rgb_, den_))
        out.update({f'comp_{wb}': comp, f'dist_{wb}': dist, f'acc_{wb}': acc, f'w_{wb}': w, f'alpha_{wb}': alpha,
                    f'gc_{wb}': gc, f'ga_{wb}': ga, f'gd_{wb}': gd, f'drgb_{wb}': d_rgb, f'dden_{wb}': d_den})
    save('composite', **out)


def _compose_nerf(m, rays, cfg, level_samples, seed):
    """model.py:100-151 composed from the reference's own functions/submodules on
    [N,.] rays (the module's forward does not run for general batches at HEAD)."""
    comp_rgbs, dists, accs = [], [], []
    inter = dict(t_vals=[], weights=[], raw_density=[], raw_rgb=[])
    torch.manual_seed(seed)
    lift = lambda x: x[None]
    for l, S in enumerate(level_samples):
        if l == 0:
            t_vals, (mean, var) = R_utils.sample_along_rays(lift(rays['origins']), lift(rays['directions']),
                                                            lift(rays['radii']), S, lift(rays['near']), lift(rays['far']),
                                                            randomized=m.randomized, lindisp=False, ray_shape=m.ray_shape)
        else:
            with torch.no_grad():
                wp = torch.cat([weights[..., :1], weights, weights[..., -1:]], dim=-1)
                wm = torch.maximum(wp[..., :-1], wp[..., 1:])
                wb = 0.5 * (wm[..., :-1] + wm[..., 1:]) + m.resample_padding
                t_vals = R_utils.sorted_piecewise_constant_pdf(t_vals, wb, S + 1, m.randomized)
            mean, var = R_utils.cast_rays(t_vals, lift(rays['origins']), lift(rays['directions']), lift(rays['radii']),
                                          m.ray_shape)
        t_vals, mean, var = t_vals[0], mean[0], var[0]
        enc = m.positional_encoding(mean, var)[0]
        enc = enc.reshape([-1, enc.shape[-1]])
        h = m.density_net0(enc)
        h = torch.cat((h, enc), -1)
        h = m.density_net1(h)
        raw_density = m.final_density(h).reshape((-1, S, 1))
        if m.use_viewdirs:
            vd = m.viewdirs_encoding(rays['viewdirs'])
            vd = torch.cat((vd, rays['viewdirs']), -1)
            vd = torch.tile(vd[:, None, :], (1, S, 1)).reshape((-1, vd.shape[-1]))
            h = m.rgb_net0(h)
            h = torch.cat((h, vd), -1)
            h = m.rgb_net1(h)
        raw_rgb = m.final_rgb(h).reshape((-1, S, 3))
        rgb = raw_rgb * (1 + 2 * m.rgb_padding) - m.rgb_padding
        density = m.density_activation(raw_density + m.density_bias)
        comp_rgb, distance, acc, weights, alpha = R_utils.volumetric_rendering(rgb, density, t_vals, rays['directions'],
                                                                             m.white_bkgd)
        comp_rgbs.append(comp_rgb)
        dists.append(distance)
        accs.append(acc)
        inter['t_vals'].append(t_vals)
        inter['weights'].append(weights)
        inter['raw_density'].append(raw_density)
        inter['raw_rgb'].append(raw_rgb)
        t_vals, weights = t_vals[None], weights[None]
    return torch.stack(comp_rgbs), torch.stack(dists), torch.stack(accs), inter


def _draws_nerf(n, level_samples, seed, randomized):
    torch.manual_seed(seed)
    if not randomized:
        return None, None
    t_rand = torch.rand(1, n, level_samples[0] + 1)[0]
    u = {}
    for l in range(1, len(level_samples)):
        k = level_samples[l] + 1
        u[l] = torch.empty(1, n, k).uniform_(to=(1 / k - torch.finfo(torch.float32).eps))[0]
    return t_rand, u


def gen_nerf_e2e(only=None):
    # (tag, config, samples per level, rays, weight seed, ray seed, draw seed)
    for tag, cfgkw, levels, n, wseed, rseed, dseed in (
            ('small_noview', dict(hidden=64, num_samples=16, max_deg=4, use_viewdirs=False, ray_shape='cylinder'),
             [16, 16], 48, 14, 20, 77),
            ('small', dict(hidden=64, num_samples=16, max_deg=4), [16, 16], 48, 11, 20, 77),
            ('small_det', dict(hidden=64, num_samples=16, max_deg=4, randomized=False, white_bkgd=True), [16, 16], 48, 11,
             20, 77),
            ('c1', dict(), [64, 64], 64, 12, 20, 77),
            ('c3', dict(num_samples=128), [64, 128], 64, 13, 20, 77),
            ('c3_1k', dict(num_samples=128), [64, 128], 1024, 13, 20, 77),
            # a second 1024-ray batch of the same configuration: other weights, rays, targets and draws
            ('c3_1k_b', dict(num_samples=128), [64, 128], 1024, 29, 41, 83),
            # the reference's own NeRF configuration (params.yaml:15-25): 128 + 128 samples, max_deg 16
            ('yaml_1k', dict(num_samples=128, max_deg=16), [128, 128], 1024, 17, 43, 89)):
        if only is not None and tag not in only:
            continue
        cfg = nerf_cfg(**cfgkw)
        m = R_model.MipNeRF(cfg)
        p = O.init_params(cfg, sar=False, seed=wseed)
        sd = m.state_dict()
        for k in p:
            assert sd[k].shape == p[k].shape, (k, sd[k].shape, p[k].shape)
        m.load_state_dict({**sd, **p})
        m.train()
        rays = make_rays(n, rseed, normalize=False)
        g = torch.Generator().manual_seed(rseed + 1)
        target = torch.rand(n, 3, generator=g)
        rgb, dist, acc, inter = _compose_nerf(m, rays, cfg, levels, seed=dseed)
        loss, psnr = m.loss_function(rgb, target, rays['lossmult'])
        names = [k for k, v in m.named_parameters() if v.requires_grad]
        grads = torch.autograd.grad(loss, [dict(m.named_parameters())[k] for k in names])
        t_rand, u = _draws_nerf(n, levels, dseed, cfg.randomized)
        out = dict(rgb=rgb, dist=dist, acc=acc, loss=loss, psnr=psnr, target=target, levels=np.array(levels),
                   wseed=wseed, cfg=np.array(repr(vars(cfg))))
        for k, v in rays.items():
            out['ray_' + k] = v
        if t_rand is not None:
            out['t_rand'] = t_rand
            for l, v in u.items():
                out[f'u_rand_{l}'] = v
        for l in range(len(levels)):
            out[f't_vals_{l}'] = inter['t_vals'][l]
            out[f'weights_{l}'] = inter['weights'][l]
            out[f'raw_density_{l}'] = inter['raw_density'][l]
            out[f'raw_rgb_{l}'] = inter['raw_rgb'][l]
        big = cfg.hidden > 64
        if n > 256:                      # large batches: the per-ray tensors would dominate the file
            for key in [k for k in out if k.startswith(('t_vals_', 'weights_', 'raw_density_', 'raw_rgb_'))]:
                del out[key]
        for k, gr in zip(names, grads):
            if big:                      # full-size nets: norms + a strided slice keep the file small
                out['gnorm_' + k] = gr.norm()
                out['gslice_' + k] = gr.reshape(-1)[::97][:256]
                if n > 256:              # a 4096-element sample per tensor: enough for a relative Frobenius error
                    step = max(1, gr.numel() // 4096)
                    out['gsample_' + k] = gr.reshape(-1)[::step][:4096]
            else:
                out['grad_' + k] = gr
        save('nerf_e2e_' + tag, **out)


def gen_scatter():
    g = torch.Generator().manual_seed(8)
    B, R, S, nsam = 1, 40, 12, 64
    near, mpp = 900., 0.5
    far = near + mpp * (nsam - 2)
    t = torch.sort(torch.rand(B, R, S + 1, generator=g) * (far - near) + near, dim=-1)[0]
    # make rays 3,4,5 identical so they land in one bin (duplicate-bin case)
    t[0, 4] = t[0, 3]
    t[0, 5] = t[0, 3]
    rgb = torch.rand(B, R, S, 3, generator=g)
    angle = torch.rand(B, R, S, 2, generator=g)
    density = torch.rand(B, R, S, 1, generator=g) * 0.2
    density[0, 4] = density[0, 3]
    density[0, 5] = density[0, 3]
    az = torch.randn(R, generator=g) * O.SAR_AZ_SIGMA
    el = torch.randn(R, generator=g) * O.SAR_EL_SIGMA
    dirs = torch.stack([torch.sin(az) * torch.cos(el), torch.cos(az) * torch.cos(el), -torch.sin(el)], -1)[None]
    power = (torch.sinc(az / O.SAR_AZ_SIGMA) ** 2 * torch.sinc(el / O.SAR_EL_SIGMA) ** 2 * 1e9)[None]
    bins = (torch.arange(nsam)[None, :] * torch.tensor([mpp])[:, None] + torch.tensor([near])[:, None])
    wl = 0.031228381041666666
    rgb_ = rgb.clone().requires_grad_(True)
    ang_ = angle.clone().requires_grad_(True)
    den_ = density.clone().requires_grad_(True)
    pulse, dist, acc, w, alpha = R_utils.volumetric_scattering(rgb_, ang_, den_, power, t, dirs, bins, wl)
    gp = torch.randn(B, nsam, 2, generator=g)
    loss = (pulse * gp).sum()
    d_rgb, d_ang, d_den = torch.autograd.grad(loss, (rgb_, ang_, den_))
    save('scatter', t_vals=t[0], rgb=rgb[0], angle=angle[0], density=density[0], power=power[0], dirs=dirs[0],
         pulse_bins=bins[0], wavelength=wl, pulse=pulse[0], dist=dist[0], acc=acc[0], w=w[0], alpha=alpha[0],
         gp=gp[0], d_rgb=d_rgb[0], d_ang=d_ang[0], d_den=d_den[0])


def sar_cfg(**kw):
    c = dict(use_viewdirs=True, randomized=True, ray_shape='cone', num_levels=2, num_samples=32, hidden=64,
             density_noise=0.0, density_bias=-1.0, rgb_padding=1e-4, resample_padding=1e-3, min_deg=0, max_deg=4,
             viewdirs_min_deg=0, viewdirs_max_deg=4, rays_per_chunk=128, total_num_rays=256,
             wavelength=0.031228381041666666, coarse_weight_decay=0.01, lr_init=.5, lr_final=5e-5,
             lr_delay_steps=5500, lr_delay_mult=0.05, weight_decay=1e-6, max_steps=2500, white_bkgd=False)
    c.update(kw)
    return SimpleNamespace(**c)


def gen_sar_e2e(only=None):
    for tag, kw in (('rand', dict()), ('det', dict(randomized=False, total_num_rays=128)),
                    ('det_noview', dict(randomized=False, total_num_rays=128, use_viewdirs=False))):
        if only is not None and tag not in only:
            continue
        cfg = sar_cfg(**kw)
        m = R_model.SARNeRF(cfg)
        p = O.init_params(cfg, sar=True, seed=31)
        sd = m.state_dict()
        for k in p:
            assert sd[k].shape == p[k].shape, (k, sd[k].shape, p[k].shape)
        m.load_state_dict({**sd, **p})
        m.train()
        nsam, nfft = 256, 512
        near = torch.tensor([900.])
        mpp = torch.tensor([0.5])
        far = near + mpp * (nsam - 2)
        radii = torch.tensor([1.12])
        ray_o = torch.tensor([[0., -1000., 500.]])
        g = torch.Generator().manual_seed(32)
        mfilt = torch.complex(torch.randn(1, nfft, generator=g), torch.randn(1, nfft, generator=g))
        target = torch.randn(1, nsam, 2, generator=g)
        R, S = cfg.rays_per_chunk, cfg.num_samples
        chunks = cfg.total_num_rays // R
        # replicate the draw order of model.py:319-320 and utils.py:224 / :39
        torch.manual_seed(55)
        draws = {}
        for c in range(chunks):
            draws[f'az_{c}'] = torch.randn(R) * O.SAR_AZ_SIGMA
            draws[f'el_{c}'] = torch.randn(R) * O.SAR_EL_SIGMA
            if cfg.randomized:
                draws[f't_rand_{c}'] = torch.rand(1, R, S + 1)[0]
                draws[f'u_rand_{c}'] = torch.empty(1, R, S + 1).uniform_(
                    to=(1 / (S + 1) - torch.finfo(torch.float32).eps))[0]
        torch.manual_seed(55)
        pulses, dist, acc = m(ray_o, None, None, mfilt, radii, near, far, nsam, mpp)
        loss, psnr = m.loss_function(pulses, target)
        names = [k for k, v in m.named_parameters() if v.requires_grad]
        grads = torch.autograd.grad(loss, [dict(m.named_parameters())[k] for k in names])
        out = dict(pulses=pulses, dist=dist, acc=acc, loss=loss, psnr=psnr, target=target, mfilt_re=mfilt.real,
                   mfilt_im=mfilt.imag, nsam=nsam, near=near, far=far, mpp=mpp, radii=radii, ray_o=ray_o, wseed=31,
                   cfg=np.array(repr(vars(cfg))), **draws)
        for k, gr in zip(names, grads):
            out['grad_' + k] = gr
        save('sar_e2e_' + tag, **out)


def gen_misc():
    g = torch.Generator().manual_seed(9)
    rgb = torch.rand(2, 33, 3, generator=g)
    tgt = torch.rand(33, 3, generator=g)
    mask = (torch.rand(33, 1, generator=g) > 0.3).float()
    lf = R_mods.NeRFLoss(0.1)
    l_m, p_m = lf(rgb, tgt, mask)
    l_n, p_n = lf(rgb, tgt)
    sl = R_mods.SARNeRFLoss(0.01)
    pul = torch.randn(2, 1, 40, 2, generator=g)
    ptg = torch.randn(1, 40, 2, generator=g)
    l_s, p_s = sl(pul, ptg)
    lin = nn.Linear(2, 2)
    opt = torch.optim.SGD(lin.parameters(), lr=1.0)
    sched = R_mods.MipLRDecay(opt, 1e-3, 5e-5, 100, 30, 0.1)
    lrs = []
    for _ in range(120):
        lrs.append(sched.get_last_lr()[0])
        opt.step()
        sched.step()
    sched2 = R_mods.MipLRDecay(opt, 0.5, 5e-5, 50)
    lrs2 = []
    for _ in range(60):
        lrs2.append(sched2.get_last_lr()[0])
        opt.step()
        sched2.step()
    # SAR ray generation (model.py:319-325), draws supplied
    az = torch.randn(50, generator=g).unsqueeze(1) * O.SAR_AZ_SIGMA
    el = torch.randn(50, generator=g).unsqueeze(1) * O.SAR_EL_SIGMA
    ray_d = torch.stack([torch.sin(az) * torch.cos(el), torch.cos(az) * torch.cos(el), -torch.sin(el)]).swapaxes(0, 2)
    ray_p = torch.sinc(az / O.SAR_AZ_SIGMA) ** 2 * np.sinc(el / O.SAR_EL_SIGMA) ** 2 * 1e9
    save('misc', rgb=rgb, tgt=tgt, mask=mask, loss_mask=l_m, psnr_mask=p_m, loss_nomask=l_n, psnr_nomask=p_n,
         pul=pul, ptg=ptg, loss_sar=l_s, psnr_sar=p_s, lrs=np.array(lrs), lrs2=np.array(lrs2),
         az=az[:, 0], el=el[:, 0], ray_d=ray_d[0], ray_p=ray_p.T[0])


def gen_ndc():
    """`convert_to_ndc` (utils.py:66-86) on float32 pinhole rays with a float32 focal (the LLFF loader's types,
    dataloader.py:338,440), and `to8b` (utils.py:619-623)."""
    rng = np.random.default_rng(21)
    n_cams, h, w = 2, 7, 9
    focal = np.float32(11.3)
    c2w = np.zeros((n_cams, 3, 4), np.float32)
    for c in range(n_cams):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        c2w[c, :, :3] = q.astype(np.float32)
        c2w[c, :, 3] = (rng.standard_normal(3) * 0.3 + [0, 0, 1.5]).astype(np.float32)
    x, y = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32), indexing='xy')
    cam = np.stack([(x - w * 0.5 + 0.5) / focal, -(y - h * 0.5 + 0.5) / focal, -np.ones_like(x)], axis=-1)
    directions = (cam[None, ..., None, :] * c2w[:, None, None, :3, :3]).sum(axis=-1)
    origins = np.broadcast_to(c2w[:, None, None, :3, -1], directions.shape)
    assert directions.dtype == np.float32
    o_ndc, d_ndc = R_utils.convert_to_ndc(origins, directions, focal, w, h)
    assert o_ndc.dtype == np.float32 and d_ndc.dtype == np.float32
    o_ndc2, d_ndc2 = R_utils.convert_to_ndc(origins, directions, focal, w, h, near=0.5)
    img = rng.standard_normal((5, 6, 3)).astype(np.float32) * 0.7 + 0.5
    img[0, 0, 0], img[0, 1, 1], img[1, 0, 2] = np.nan, np.inf, -np.inf          # to8b maps them to 0, 255, 0
    save('ndc', c2w=c2w, h=h, w=w, focal=focal, origins=np.ascontiguousarray(origins), directions=directions,
         o_ndc=o_ndc, d_ndc=d_ndc, o_ndc_near05=o_ndc2, d_ndc_near05=d_ndc2, img=img, img8=R_utils.to8b(img))


def gen_rays():
    """Ray generation of the dataset classes on bare instances (no files are read): `NeRFDataset.generate_rays`
    (dataloader.py:103-138) and `LLFF.generate_rays` (dataloader.py:437-460), then `flatten_to_pytorch`
    (dataloader.py:140-144), i.e. exactly the `[n, .]` float32 tensors a training batch is gathered from."""
    rng = np.random.default_rng(33)
    out = {}
    for tag, cls, C, h, w, focal, near, far in (('pinhole', R_data.NeRFDataset, 3, 11, 13, 17.25, 2., 6.),
                                               ('pinhole_tele', R_data.NeRFDataset, 2, 5, 8, 1111.0, 2., 6.),
                                               ('llff', R_data.LLFF, 3, 9, 11, np.float32(13.7), 0., 1.)):
        c2w = np.zeros((C, 3, 4), np.float32)
        for c in range(C):
            q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
            c2w[c, :, :3] = q.astype(np.float32)
            c2w[c, :, 3] = (rng.standard_normal(3) * 0.3 + [0, 0, 1.5]).astype(np.float32)
        ds = object.__new__(cls)
        ds.h, ds.w, ds.focal, ds.near, ds.far, ds.cam_to_world = h, w, focal, near, far, c2w
        ds.rays, ds.images = None, None
        ds.generate_rays()
        ds.flatten_to_pytorch()
        out.update({f'{tag}_c2w': c2w, f'{tag}_hwf': np.array([h, w, float(focal)], np.float64),
                    f'{tag}_nearfar': np.array([near, far], np.float64)})
        for k, v in ds.rays._asdict().items():
            assert v.dtype == torch.float32 and v.shape[0] == C * h * w
            out[f'{tag}_{k}'] = v
    save('rays', **out)


if __name__ == '__main__':
    if _ARGS == ['rays']:
        gen_rays()
        sys.exit(0)
    if _ARGS == ['posenc']:
        gen_posenc()
        sys.exit(0)
    if _ARGS == ['c3_1k']:
        gen_nerf_e2e(only=('c3_1k',))
        sys.exit(0)
    if _ARGS and _ARGS[0] == 'nerf':          # nerf <tag> ...: only these end-to-end fixtures
        gen_nerf_e2e(only=tuple(_ARGS[1:]))
        sys.exit(0)
    if _ARGS == ['ndc']:
        gen_ndc()
        sys.exit(0)
    if _ARGS == ['noview']:
        gen_sar_e2e(only=('det_noview',))
        gen_nerf_e2e(only=('small_noview',))
        sys.exit(0)
    gen_ndc()
    gen_rays()
    gen_sample()
    gen_posenc()
    gen_resample()
    gen_composite()
    gen_scatter()
    gen_misc()
    gen_nerf_e2e()
    gen_sar_e2e()
