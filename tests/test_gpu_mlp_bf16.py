"""Fused bf16 MFMA MLP kernels against the CPU oracle (fp32).  Tolerance 2e-2 of the output range
(bf16 inputs to every layer, fp32 accumulation), as the parity gate for the bf16 MLP states."""
from types import SimpleNamespace

import pytest
import torch

from oracle import nerf_oracle as O
from tests.conftest import T, cfg_from, relerr

pytestmark = pytest.mark.gpu


def make_cfg(**kw):
    c = dict(use_viewdirs=True, randomized=True, ray_shape='cone', white_bkgd=False, num_levels=2, num_samples=128,
             min_deg=0, max_deg=10, viewdirs_min_deg=0, viewdirs_max_deg=4, density_noise=0.0, rgb_padding=0.001,
             resample_padding=0.01, density_bias=-1.0, hidden=256, coarse_weight_decay=0.1, lr_init=1e-3,
             lr_final=5e-5, max_steps=100, lr_delay_steps=0, lr_delay_mult=1, weight_decay=1e-5, mlp_precision='bf16')
    c.update(kw)
    return SimpleNamespace(**c)


def make_rays(n, seed):
    g = torch.Generator().manual_seed(seed)
    o = 0.1 * torch.randn(n, 3, generator=g) + torch.tensor([0., 0., 4.])
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True) * (1.0 + 0.2 * torch.rand(n, 1, generator=g))
    v = d / d.norm(dim=-1, keepdim=True)
    return O.Rays(o, d, v, torch.full((n, 1), 0.005), torch.ones(n, 1), torch.full((n, 1), 2.), torch.full((n, 1), 6.))


CASES = [('small', dict(hidden=64, num_samples=16, max_deg=4), 16, 37),
         ('c3', dict(), 128, 19),
         ('c3_coarse', dict(), 64, 23),
         ('c1', dict(num_samples=64), 64, 11),
         ('deg16', dict(max_deg=16), 128, 7)]


@pytest.mark.parametrize('tag,kw,S,n', CASES)
def test_mlp_forward_vs_oracle(tag, kw, S, n):
    from nerf_amd import mlp_bf16
    from nerf_amd.model import MipNeRF
    cfg = make_cfg(**kw)
    p = O.init_params(cfg, seed=5)
    m = MipNeRF(cfg).cuda()
    m.load_state_dict(p, strict=False)
    rays = make_rays(n, 3)
    g = torch.Generator().manual_seed(4)
    t = O.stratified_t(rays.near, rays.far, S, torch.rand(n, S + 1, generator=g))
    mean, cov = O.cast_rays(t, rays.origins, rays.directions, rays.radii, 'cone')
    enc = O.ipe(mean, cov, cfg.min_deg, cfg.max_deg).reshape(n * S, -1)
    vf = O.view_features(rays.viewdirs, 0, 4)
    ref_d, ref_rgb = O.mlp_nerf(p, enc, vf[:, None, :].expand(-1, S, -1).reshape(n * S, -1))

    pk = mlp_bf16.packed_for(m, need_bwd=False)
    encf = mlp_bf16.encode_frag(t.cuda(), rays.origins.cuda(), rays.directions.cuda(), rays.radii.cuda(), 'cone',
                                cfg.min_deg, cfg.max_deg)
    viewf = mlp_bf16.view_frag(rays.viewdirs.cuda(), 0, 4)
    for save in (False, True):
        d, rgb, _, acts, _ = mlp_bf16.mlp_forward(pk, encf, viewf, n * S, S, n, save=save)
        assert relerr(d.cpu(), ref_d[:, 0]) < 2e-2, (tag, save)
        assert float((rgb.cpu() - ref_rgb).abs().max()) < 2e-2, (tag, save)
