import sys, torch
sys.path.insert(0, '.')
from oracle import nerf_oracle as O
from nerf_amd import mlp_bf16
from nerf_amd.model import MipNeRF
from nerf_amd.utils import Rays
from tests.test_gpu_mlp_bf16 import make_cfg, make_rays
from scratch.dbg3_lib import frag_to_dense
kw = dict(hidden=64, num_samples=16, max_deg=4); S = 16; n = 37
cfg = make_cfg(**kw)
p = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=5).items()}
m = MipNeRF(cfg).cuda(); m.load_state_dict(p, strict=False)
rays = make_rays(n, 3)
g = torch.Generator().manual_seed(4)
t = O.stratified_t(rays.near, rays.far, S, torch.rand(n, S + 1, generator=g))
ga = torch.randn(n * S, 1, generator=g); gb = torch.randn(n * S, 3, generator=g)
gr = Rays(*(x.cuda() for x in (rays.origins, rays.directions, rays.viewdirs, rays.radii, rays.lossmult, rays.near, rays.far)))
viewf = mlp_bf16.view_frag(gr.viewdirs, 0, 4)
M = n * S
encf = mlp_bf16.encode_frag(t.cuda(), gr.origins, gr.directions, gr.radii, 'cone', 0, 4)
e = frag_to_dense(encf, 32, ((M + 255)//256)*256)
print('enc nan rows', torch.nonzero(torch.isnan(e).any(1)).flatten().tolist()[:20], 'cols', torch.nonzero(torch.isnan(e).any(0)).flatten().tolist())
d, rgb = mlp_bf16.query_nerf(m, t.cuda(), gr, viewf, S)
((d * ga.cuda()).sum() + (rgb * gb.cuda()).sum()).backward()
for k, v in m.named_parameters():
    if v.grad is not None and torch.isnan(v.grad).any():
        print('nan grad', k, torch.nonzero(torch.isnan(v.grad).any(0)).flatten().tolist()[:10])
