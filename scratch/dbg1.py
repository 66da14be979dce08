import sys, torch
sys.path.insert(0, '.')
from tests.conftest import load_golden, T, cfg_from, relerr
from oracle import nerf_oracle as O
from nerf_amd import ops, mlp
from nerf_amd.model import MipNeRF, SARNeRF
from nerf_amd.utils import Rays

fx = load_golden('nerf_e2e_small')
cfg = cfg_from(fx); cfg.mlp_precision='fp32'; cfg.level_samples=[16,16]
m = MipNeRF(cfg).cuda(); m.load_state_dict(O.init_params(cfg, seed=11), strict=False); m.eval()
rays = Rays(*(T(fx['ray_'+k]).cuda() for k in ('origins','directions','viewdirs','radii','lossmult','near','far')))
with torch.no_grad():
    n=48; S=16
    t = ops.stratified_t(rays.near, rays.far, S, None, False)
    t2 = ops.stratified_t(rays.near, rays.far, S, None, False)
    print('t eq', torch.equal(t,t2))
    mean,cov = ops.cast_rays(t, rays.origins, rays.directions, rays.radii, 'cone')
    mean2,cov2 = ops.cast_rays(t, rays.origins, rays.directions, rays.radii, 'cone')
    print('cast eq', torch.equal(mean,mean2), torch.equal(cov,cov2))
    enc = ops.ipe_encode(mean,cov,0,4).reshape(n*S,-1); enc2 = ops.ipe_encode(mean,cov,0,4).reshape(n*S,-1)
    print('enc eq', torch.equal(enc,enc2))
    vf = ops.pe_encode(rays.viewdirs,0,4,True)
    a = m._query(enc, vf, S); b = m._query(enc, vf, S)
    print('mlp eq', [torch.equal(x,y) for x,y in zip(a,b)], [float((x-y).abs().max()) for x,y in zip(a,b)])
    w = torch.randn(64,24).cuda(); bb=torch.randn(64).cuda()
    y1 = mlp.linear(enc, w, bb, 'relu'); y2 = mlp.linear(enc, w, bb, 'relu')
    print('lin eq', torch.equal(y1,y2), float((y1-y2).abs().max()))
    c1 = ops.composite(a[1].reshape(n,S,3), a[0].reshape(n,S), t, rays.directions, False, (True,0.001,-1.0))
    c2 = ops.composite(a[1].reshape(n,S,3), a[0].reshape(n,S), t, rays.directions, False, (True,0.001,-1.0))
    print('comp eq', [torch.equal(x,y) for x,y in zip(c1,c2)])
    r1 = ops.resample(t, c1[3], S+1, 0.01, None, True); r2 = ops.resample(t, c1[3], S+1, 0.01, None, True)
    print('resample eq', torch.equal(r1,r2))
    o1 = m(rays); o2 = m(rays)
    print('model eq', [torch.equal(x,y) for x,y in zip(o1,o2)], [float((x-y).abs().max()) for x,y in zip(o1,o2)])
    print('level diffs', [(float((o1[0][l]-o2[0][l]).abs().max())) for l in range(2)])
    for it in range(3):
        r1 = ops.resample(t, c1[3], S+1, 0.01, None, True)
        mean,cov = ops.cast_rays(r1, rays.origins, rays.directions, rays.radii, 'cone')
        enc = ops.ipe_encode(mean,cov,0,4).reshape(n*S,-1)
        a = m._query(enc, vf, S)
        cc = ops.composite(a[1].reshape(n,S,3), a[0].reshape(n,S), r1, rays.directions, False, (True,0.001,-1.0))
        print(it, float(r1.sum()), float(mean.sum()), float(enc.sum()), float(a[0].sum()), float(a[1].sum()), [float(x.sum()) for x in cc[:3]])
    for it in range(3):
        o = m(rays)
        print('model', it, [float(x.double().sum()) for x in o])
