import sys, torch
sys.path.insert(0, '.')
from tests.conftest import load_golden, T, cfg_from, relerr
from oracle import nerf_oracle as O
from nerf_amd.model import SARNeRF

for tag in ('det','rand'):
    fx = load_golden('sar_e2e_'+tag)
    cfg2 = cfg_from(fx); cfg2.wavelength = 40.0; cfg2.mlp_precision='fp32'
    chunks = cfg2.total_num_rays // cfg2.rays_per_chunk
    az_el = [(T(fx[f'az_{c}']), T(fx[f'el_{c}'])) for c in range(chunks)]
    t_rand = [T(fx[f't_rand_{c}']) for c in range(chunks)] if cfg2.randomized else None
    u_rand = [T(fx[f'u_rand_{c}']) for c in range(chunks)] if cfg2.randomized else None
    mfilt = torch.complex(T(fx['mfilt_re']), T(fx['mfilt_im']))
    nsam = int(fx['nsam'])
    args = (T(fx['ray_o']), None, None, mfilt, T(fx['radii']), T(fx['near']), T(fx['far']), nsam, T(fx['mpp']))
    m2 = SARNeRF(cfg2).cuda(); m2.load_state_dict(O.init_params(cfg2, sar=True, seed=31), strict=False); m2.train()
    p = {k: v.requires_grad_(True) for k, v in O.init_params(cfg2, sar=True, seed=31).items()}
    ur = [{1: u} for u in u_rand] if u_rand is not None else None
    ref_p, ref_d, ref_a, inter = O.sar_render(p, T(fx['ray_o']), az_el, mfilt, float(fx['radii'][0]), float(fx['near'][0]),
                                          float(fx['far'][0]), nsam, float(fx['mpp'][0]), cfg2, t_rand, ur, post=True)
    cu = lambda xs: None if xs is None else [x.cuda() for x in xs]
    got_p, got_d, got_a = m2(*args, az_el=az_el, t_rand=cu(t_rand), u_rand=cu(u_rand))
    print(tag, 'pulse relerr', relerr(got_p.cpu(), ref_p), 'dist', relerr(got_d.cpu(), ref_d))
    for l in range(2):
        print(' level', l, 'pulse relerr', relerr(got_p[l].cpu(), ref_p[l]))
    tgt = T(fx['target'])
    O.sar_loss(ref_p, tgt, cfg2.coarse_weight_decay)[0].backward()
    m2.loss_function(got_p, tgt.cuda())[0].backward()
    for k, v in m2.named_parameters():
        if k.endswith('scales'): continue
        print('  ', k, 'relerr %.3e' % relerr(v.grad.cpu(), p[k].grad), 'norms %.3e %.3e' % (float(v.grad.norm()), float(p[k].grad.norm())))
    # bins occupancy
    for i, it in enumerate(inter):
        b = it['bins']; print(' inter', i, 'unique bins', len(torch.unique(b)), 'of', b.numel())
