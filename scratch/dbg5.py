import sys, torch
sys.path.insert(0, '.')
from oracle import nerf_oracle as O
from nerf_amd import mlp_bf16
from nerf_amd.model import MipNeRF
from tests.test_gpu_mlp_bf16 import make_cfg, make_rays
from scratch.dbg3_lib import frag_to_dense
def fro(a, b): return float((a - b).norm() / b.norm())
for kw, S, n in ((dict(hidden=64, num_samples=16, max_deg=4), 16, 37), (dict(), 128, 19)):
    cfg = make_cfg(**kw)
    p = O.init_params(cfg, seed=5)
    m = MipNeRF(cfg).cuda(); m.load_state_dict(p, strict=False)
    rays = make_rays(n, 3)
    g = torch.Generator().manual_seed(4)
    t = O.stratified_t(rays.near, rays.far, S, torch.rand(n, S + 1, generator=g))
    M = n * S
    ga = torch.randn(M, 1, generator=g); gb = torch.randn(M, 3, generator=g)
    pk = mlp_bf16.packed_for(m, True)
    encf = mlp_bf16.encode_frag(t.cuda(), rays.origins.cuda(), rays.directions.cuda(), rays.radii.cuda(), 'cone', cfg.min_deg, cfg.max_deg)
    viewf = mlp_bf16.view_frag(rays.viewdirs.cuda(), 0, 4)
    d, rgbk, _, acts, _ = mlp_bf16.mlp_forward(pk, encf, viewf, M, S, n, save=True)
    dys = mlp_bf16.mlp_backward(pk, acts, M, ga.cuda().reshape(-1), gb.cuda(), rgbk)
    H, W = cfg.hidden, cfg.num_samples
    HP, WP = pk.HT * 32, pk.WT * 32
    X = [frag_to_dense(acts[i], HP if i < 9 else WP, M) for i in range(10)]
    DY = [frag_to_dense(dys[i], HP if i < 9 else WP, M) for i in range(10)]
    bf = lambda w: w.to(torch.bfloat16).float()
    s = rgbk.cpu()
    dz = gb * s * (1 - s)
    names = mlp_bf16.STREAMED
    Wt = {i: bf(p[names[i] + '.weight']) for i in range(10)}
    r9 = (bf(dz) @ bf(p['final_rgb.0.weight'])) * (X[9][:, :W] > 0)
    print('dy9', fro(DY[9][:, :W], r9))
    r8 = bf(r9) @ Wt[9][:, :H]
    print('dy8', fro(DY[8][:, :H], r8))
    r = (bf(r8) @ Wt[8] + bf(ga) @ bf(p['final_density.0.weight'])) * (X[7][:, :H] > 0)
    print('dy7', fro(DY[7][:, :H], r))
    for i in (6, 5, 4, 3, 2, 1, 0):
        r = (bf(r) @ Wt[i + 1][:, :H]) * (X[i][:, :H] > 0)
        print('dy%d' % i, fro(DY[i][:, :H], r))
