import sys, torch
sys.path.insert(0, '.')
from oracle import nerf_oracle as O
from nerf_amd import mlp_bf16
from nerf_amd.model import MipNeRF
from tests.test_gpu_mlp_bf16 import make_cfg, make_rays
from tests.conftest import relerr

def frag_to_dense(buf, F, M):
    nblk = buf.numel() // (F * 128)
    x = buf.view(torch.bfloat16).reshape(nblk, F // 16, 2, 64, 8).float().cpu()
    out = torch.zeros(nblk * 64, F)
    for h in range(2):
        for j in range(8):
            f_in = 8 * (j >> 2) + 4 * h + (j & 3)
            # lanes h*32 .. h*32+31 -> samples ct*32 + c
            v = x[:, :, :, h * 32:(h + 1) * 32, j]            # [nblk, ks, ct, 32]
            for ks in range(F // 16):
                out[:, ks * 16 + f_in] = v[:, ks].reshape(nblk * 64)
    return out[:M]

kw = dict(hidden=64, num_samples=16, max_deg=4); S = 16; n = 37
cfg = make_cfg(**kw)
p = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=5).items()}
m = MipNeRF(cfg).cuda(); m.load_state_dict(p, strict=False)
rays = make_rays(n, 3)
g = torch.Generator().manual_seed(4)
t = O.stratified_t(rays.near, rays.far, S, torch.rand(n, S + 1, generator=g))
ga = torch.randn(n * S, 1, generator=g); gb = torch.randn(n * S, 3, generator=g)
mean, cov = O.cast_rays(t, rays.origins, rays.directions, rays.radii, 'cone')
enc = O.ipe(mean, cov, cfg.min_deg, cfg.max_deg).reshape(n * S, -1)
vf = O.view_features(rays.viewdirs, 0, 4)
vrows = vf[:, None, :].expand(-1, S, -1).reshape(n * S, -1)
# oracle with intermediates
acts_ref = []; pre = []
h = enc
for i in (0, 2, 4, 6):
    z = O.linear(h, p, 'density_net0.%d' % i); z.retain_grad(); pre.append(z); h = torch.relu(z); acts_ref.append(h)
h = torch.cat((h, enc), -1)
for i in (0, 2, 4, 6):
    z = O.linear(h, p, 'density_net1.%d' % i); z.retain_grad(); pre.append(z); h = torch.relu(z); acts_ref.append(h)
rd = O.linear(h, p, 'final_density.0')
z = O.linear(h, p, 'rgb_net0.0'); z.retain_grad(); pre.append(z); acts_ref.append(z)
z9 = O.linear(torch.cat((z, vrows), -1), p, 'rgb_net1.0'); z9.retain_grad(); pre.append(z9); x9 = torch.relu(z9); acts_ref.append(x9)
zr = O.linear(x9, p, 'final_rgb.0'); zr.retain_grad(); rgb = torch.sigmoid(zr)
((rd * ga).sum() + (rgb * gb).sum()).backward()

M = n * S
pk = mlp_bf16.packed_for(m, True)
encf = mlp_bf16.encode_frag(t.cuda(), rays.origins.cuda(), rays.directions.cuda(), rays.radii.cuda(), 'cone', 0, 4)
viewf = mlp_bf16.view_frag(rays.viewdirs.cuda(), 0, 4)
d, rgbk, _, acts, _ = mlp_bf16.mlp_forward(pk, encf, viewf, M, S, n, save=True)
print('enc frag', relerr(frag_to_dense(encf, 32, M)[:, :24], enc))
widths = [64] * 9 + [32, 32]
for i in range(10):
    a = frag_to_dense(acts[i], widths[i], M)
    print('act', i, relerr(a[:, :acts_ref[i].shape[1]], acts_ref[i].detach()))
print('view', relerr(frag_to_dense(acts[10], 32, M)[:, :27], vrows))
dys = mlp_bf16.mlp_backward(pk, acts, M, ga.cuda().reshape(-1), gb.cuda(), rgbk)
for i in range(10):
    dyk = frag_to_dense(dys[i], widths[i], M)
    print('dy', i, relerr(dyk[:, :pre[i].shape[1]], pre[i].grad), 'nan' if torch.isnan(dyk).any() else '')
dz = frag_to_dense(dys[10], 32, M); print('dz', relerr(dz[:, :3], zr.grad), float(dz[:, 3:].abs().max()))
dd = frag_to_dense(dys[11], 32, M); print('dd', relerr(dd[:, :1], ga), float(dd[:, 1:].abs().max()))
# wgrad check for layer 0 using unpacked frags
G = torch.zeros(64, 24, device='cuda'); gbias = torch.zeros(64, device='cuda')
mlp_bf16._wgrad(dys[0], 64, encf, 32, M, G, 0, 64, 24, gbias)
dy0 = frag_to_dense(dys[0], 64, M); e = frag_to_dense(encf, 32, M)
refG = dy0.t() @ e[:, :24]
print('wgrad L0 vs dense', relerr(G.cpu(), refG), 'bias', relerr(gbias.cpu(), dy0.sum(0)))
print(G.cpu()[:3, 18:24]); print(refG[:3, 18:24])
G = torch.zeros(64, 64, device='cuda'); gbias = torch.zeros(64, device='cuda')
mlp_bf16._wgrad(dys[1], 64, acts[0], 64, M, G, 0, 64, 64, gbias)
dy1 = frag_to_dense(dys[1], 64, M); x0 = frag_to_dense(acts[0], 64, M)
print('wgrad L1 vs dense', relerr(G.cpu(), dy1.t() @ x0), 'bias', relerr(gbias.cpu(), dy1.sum(0)))
print('---- dy9 analysis')
dyk9 = frag_to_dense(dys[9], 32, M)[:, :16]
ref_masked = z9.grad
Wh = p['final_rgb.0.weight'].detach()
unmasked = zr.grad @ Wh           # [M,16]
print('vs masked', relerr(dyk9, ref_masked), 'vs unmasked', relerr(dyk9, unmasked))
print(dyk9[:4, :8]); print(ref_masked[:4, :8]); print(unmasked[:4, :8])
print('ratio', (dyk9 / (ref_masked + 1e-12))[:4, :8])
err = (dyk9 - ref_masked).abs().max(dim=1)[0]
bad = torch.nonzero(err > 0.02).flatten()
print('bad rows', bad.numel(), bad[:40].tolist())
err0 = (frag_to_dense(dys[0], 64, M) - pre[0].grad).abs().max(dim=1)[0]
print('dy0 bad rows', torch.nonzero(err0 > 0.05 * pre[0].grad.abs().max()).flatten()[:60].tolist())
