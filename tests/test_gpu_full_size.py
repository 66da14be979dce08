"""Size-independent properties at the benchmark's full sizes (the oracle cannot run these in seconds):
chunk invariance of a deterministic train step at 65 536 rays x (64+128) samples, forward invariance under
chunking at 262 144 rays, range properties of the composited outputs."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu


def cfg(randomized):
    return SimpleNamespace(use_viewdirs=True, randomized=randomized, ray_shape='cone', white_bkgd=False, num_levels=2,
                           num_samples=128, level_samples=[64, 128], min_deg=0, max_deg=10, viewdirs_min_deg=0,
                           viewdirs_max_deg=4, density_noise=0.0, rgb_padding=0.001, resample_padding=0.01,
                           density_bias=-1.0, hidden=256, coarse_weight_decay=0.1, lr_init=1e-3, lr_final=5e-5,
                           max_steps=100000, lr_delay_steps=2500, lr_delay_mult=0.1, weight_decay=1e-5,
                           mlp_precision='bf16')


def test_c3_gradients_do_not_depend_on_chunking():
    """Gradients of the 65 536-ray step add up over chunks: one chunk of 65 536 rays and four of 16 384 give the
    same loss and the same 24 gradients (deterministic sampling; fp32 atomics reorder sums)."""
    from nerf_amd.model import MipNeRF
    from nerf_amd.train import nerf_train_step, synthetic_rays
    torch.manual_seed(0)
    m = MipNeRF(cfg(False)).cuda().train()
    rays, pix = synthetic_rays(65536, 'cuda', seed=5)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    out = []
    for chunk in (65536, 16384):
        m.load_state_dict(state)
        m._weights_changed()
        opt = torch.optim.SGD(m.parameters(), lr=0.0)            # keep the weights: only gradients are compared
        loss = nerf_train_step(m, rays, pix, opt, chunk=chunk)
        out.append((float(loss), [p.grad.clone() for p in m.parameters() if p.requires_grad]))
    assert abs(out[0][0] - out[1][0]) < 1e-5 * abs(out[0][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert torch.isfinite(a).all()
        assert float((a - b).norm()) <= 2e-4 * float(a.norm()) + 1e-12


def test_c2_forward_does_not_depend_on_chunking_and_stays_in_range():
    """C2 at its full size.  Rays are independent: rendering 1 048 576 rays at once or 8 192 at a time gives identical
    pixels (deterministic sampling); acc in [0, 1], colour within the padded sigmoid range, distance within [near, far];
    and every ray of the big batch equals the same ray rendered in a batch of the oracle's size, whose pixels the
    oracle checks (test_nerf_end_to_end_bf16)."""
    from nerf_amd.model import MipNeRF
    from nerf_amd.train import synthetic_rays
    from nerf_amd.utils import namedtuple_map
    torch.manual_seed(1)
    m = MipNeRF(cfg(False)).cuda().train()
    m.randomized = False
    n = 1 << 20
    rays, _ = synthetic_rays(n, 'cuda', seed=6)
    with torch.no_grad():
        rgb, dist, acc = m(rays)
        starts = list(range(0, n, 131072)) + [n - 8192, n - 1000]
        parts = [m(namedtuple_map(lambda t: t[i:i + 8192], rays)) for i in starts]
    for i, (r, d, a) in zip(starts, parts):
        s = slice(i, min(i + 8192, n))
        assert torch.equal(r, rgb[:, s]) and torch.equal(d, dist[:, s]) and torch.equal(a, acc[:, s]), i
    assert rgb.shape == (2, n, 3) and bool(torch.isfinite(rgb).all())
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-6
    assert float(rgb.min()) >= -0.001 * 1.0001 and float(rgb.max()) <= 1.001 * 1.0001
    assert float(dist.min()) >= 2.0 - 1e-5 and float(dist.max()) <= 6.0 + 1e-5


def test_render_image_from_pose():
    """`render_image` (model.py:153-175) on rays generated on the device for a 96 x 128 camera: CPU ray tensors in,
    8-bit colours out; chunked rendering equals one forward over all rays quantised as `to8b` does; depth and
    accumulation are the fine level's; the ragged last chunk is handled."""
    import numpy as np
    from nerf_amd import ops
    from nerf_amd.model import MipNeRF, to8b
    from nerf_amd.utils import namedtuple_map
    torch.manual_seed(2)
    m = MipNeRF(cfg(False)).cuda().train()
    m.randomized = False
    h, w = 96, 128
    c2w = torch.eye(4)[None, :3].clone()
    c2w[0, 2, 3] = 4.0
    rays = ops.pinhole_rays(c2w.cuda(), h, w, 110.0, 2.0, 6.0)
    with torch.no_grad():
        rgb, dist, acc = m(rays)
    img, depth, alpha = m.render_image(namedtuple_map(lambda t: t.cpu(), rays), h, w, chunks=5000)
    assert img.dtype == np.uint8 and img.shape == (h, w, 3) and depth.shape == (h, w) and alpha.shape == (h, w)
    assert np.array_equal(img, to8b(rgb[-1].reshape(h, w, 3).cpu().numpy()))
    assert np.array_equal(depth, dist[-1].reshape(h, w).cpu().numpy())
    assert np.array_equal(alpha, acc[-1].reshape(h, w).cpu().numpy())


def test_training_fits_a_scene_from_device_generated_batches():
    """The whole loop a user of the reference runs, on the device: poses + images -> shuffled ray batches ->
    coarse/fine render -> masked loss -> backward -> AdamW -> MipLRDecay -> repacked weights.  A view-independent
    coloured pattern on three cameras is fitted for 120 steps; the fine-level PSNR must rise well above the
    untrained model's."""
    from nerf_amd.model import MipNeRF
    from nerf_amd.rays import RayBatches
    from nerf_amd.train import nerf_train_step
    from nerf_amd.util_modules import MipLRDecay, mse_to_psnr
    torch.manual_seed(3)
    c = cfg(True)
    c.lr_init, c.lr_final, c.max_steps, c.lr_delay_steps, c.lr_delay_mult = 2e-3, 2e-4, 120, 10, 0.1
    m = MipNeRF(c).cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=c.lr_init, weight_decay=c.weight_decay, fused=True)
    sched = MipLRDecay(opt, c.lr_init, c.lr_final, c.max_steps, c.lr_delay_steps, c.lr_delay_mult)
    C, h, w = 3, 64, 64
    c2w = torch.eye(4)[None, :3].repeat(C, 1, 1)
    c2w[:, 2, 3] = 4.0
    c2w[1, 0, 3], c2w[2, 1, 3] = 0.3, -0.3
    yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing='ij')
    img = torch.stack([xx, yy, 0.5 + 0.5 * torch.sin(6.28 * (xx + yy))], -1)
    images = img[None].repeat(C, 1, 1, 1).cuda()
    batches = RayBatches(c2w.cuda(), h, w, 70.0, 2.0, 6.0, images=images, batch_size=4096, seed=1)

    def fine_psnr():
        with torch.no_grad():
            rays, pix = next(iter(RayBatches(c2w.cuda(), h, w, 70.0, 2.0, 6.0, images=images, batch_size=4096,
                                             shuffle=False)))
            rgb = m(rays)[0]
            return float(mse_to_psnr(((rgb[-1] - pix) ** 2).mean()))

    before = fine_psnr()
    losses, step = [], 0
    while step < 120:
        for rays, pix in batches:
            losses.append(float(nerf_train_step(m, rays, pix, opt, chunk=4096, scheduler=sched)))
            step += 1
            if step == 120:
                break
    after = fine_psnr()
    assert all(l == l for l in losses)                   # finite
    assert sum(losses[-10:]) < 0.5 * sum(losses[:10]), (losses[:3], losses[-3:])
    assert after > before + 4.0, (before, after)


def _fit_scene(precision, operands, steps, seed=3):
    """`steps` training steps on the synthetic three-camera scene of the test above -> (fine-level PSNR over every
    pixel of the three images with the sampling jitter off, mean training loss of the last 20 steps)."""
    from nerf_amd.model import MipNeRF
    from nerf_amd.rays import RayBatches
    from nerf_amd.train import nerf_train_step
    from nerf_amd.util_modules import MipLRDecay, mse_to_psnr
    torch.manual_seed(seed)
    c = cfg(True)
    c.mlp_precision, c.wgrad_operands = precision, operands
    c.lr_init, c.lr_final, c.max_steps, c.lr_delay_steps, c.lr_delay_mult = 2e-3, 2e-4, steps, 10, 0.1
    m = MipNeRF(c).cuda().train()
    opt = torch.optim.AdamW(m.parameters(), lr=c.lr_init, weight_decay=c.weight_decay, fused=True)
    sched = MipLRDecay(opt, c.lr_init, c.lr_final, c.max_steps, c.lr_delay_steps, c.lr_delay_mult)
    C, h, w = 3, 64, 64
    c2w = torch.eye(4)[None, :3].repeat(C, 1, 1)
    c2w[:, 2, 3] = 4.0
    c2w[1, 0, 3], c2w[2, 1, 3] = 0.3, -0.3
    yy, xx = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing='ij')
    img = torch.stack([xx, yy, 0.5 + 0.5 * torch.sin(6.28 * (xx + yy))], -1)
    images = img[None].repeat(C, 1, 1, 1).cuda()
    batches = RayBatches(c2w.cuda(), h, w, 70.0, 2.0, 6.0, images=images, batch_size=4096, seed=1)
    torch.manual_seed(100 + seed)                      # the sampling jitter: the same draws for every format
    losses, step = [], 0
    while step < steps:
        for rays, pix in batches:
            losses.append(nerf_train_step(m, rays, pix, opt, chunk=4096, scheduler=sched))
            step += 1
            if step == steps:
                break
    losses = [float(l) for l in losses]
    assert all(l == l for l in losses)
    m.randomized = False
    se, n = 0.0, 0
    with torch.no_grad():
        for rays, pix in RayBatches(c2w.cuda(), h, w, 70.0, 2.0, 6.0, images=images, batch_size=4096, shuffle=False):
            rgb = m(rays)[0]
            se += float(((rgb[-1] - pix) ** 2).sum())
            n += pix.numel()
    return float(mse_to_psnr(torch.tensor(se / n))), sum(losses[-20:]) / 20


def test_mx_weight_gradient_operands_train_like_bf16_operands():
    """Convergence A/B of the saved weight-gradient operand formats: the same scene, initial weights, batches and
    sampling draws, 300 AdamW steps with the reference's learning-rate schedule, three times -- the exact-fp32 path,
    the bf16 chain with bf16 operand copies, the bf16 chain with MX copies (activations e4m3, gradients block-scaled
    e2m3).  The MX run must end within 0.2 dB (fine-level PSNR over all pixels) of the bf16-operand run, and both
    within 1 dB of the fp32 path; the three end points are printed."""
    out = {}
    for tag, prec, ops_ in (('fp32', 'fp32', None), ('bf16', 'bf16', 'bf16'), ('mx', 'bf16', 'mx')):
        out[tag] = _fit_scene(prec, ops_, 300)
    print('convergence A/B after 300 steps (fine PSNR dB, mean loss of the last 20 steps):',
          {k: (round(v[0], 3), round(v[1], 6)) for k, v in out.items()})
    assert abs(out['mx'][0] - out['bf16'][0]) < 0.2, out
    assert abs(out['bf16'][0] - out['fp32'][0]) < 1.0 and abs(out['mx'][0] - out['fp32'][0]) < 1.0, out
    assert abs(out['mx'][1] - out['bf16'][1]) < 0.05 * out['bf16'][1], out


def _sar_cfg(precision, wavelength, R=2048):
    """The reference's SAR-NeRF hyperparameters (params.yaml:67-93): R rays per chunk, one chunk per pulse."""
    return SimpleNamespace(use_viewdirs=True, randomized=True, ray_shape='cone', num_levels=2, num_samples=180,
                           hidden=156, density_noise=0.0, density_bias=-1.0, rgb_padding=1e-4, resample_padding=1e-3,
                           min_deg=0, max_deg=4, viewdirs_min_deg=0, viewdirs_max_deg=4, rays_per_chunk=R,
                           total_num_rays=R, wavelength=wavelength, coarse_weight_decay=0.01, lr_init=.5,
                           lr_final=5e-5, lr_delay_steps=5500, lr_delay_mult=0.05, weight_decay=1e-6, max_steps=2500,
                           white_bkgd=False, mlp_precision=precision)


def _sar_pulse(R, nsam=2048, nfft=8192, seed=0):
    from oracle import nerf_oracle as O
    g = torch.Generator().manual_seed(seed)
    near, mpp = torch.tensor([900.]), torch.tensor([0.5])
    far = near + mpp * (nsam - 2)
    mfilt = torch.complex(torch.randn(1, nfft, generator=g), torch.randn(1, nfft, generator=g))
    target = torch.randn(1, nsam, 2, generator=g)
    az_el = [(torch.randn(R, generator=g) * O.SAR_AZ_SIGMA, torch.randn(R, generator=g) * O.SAR_EL_SIGMA)]
    t_rand = [torch.rand(R, 181, generator=g)]
    u_rand = [torch.empty(R, 181).uniform_(0, 1 / 181 - O.EPS32, generator=g)]
    return dict(ray_o=torch.tensor([[0., -1000., 500.]]), mfilt=mfilt, radii=torch.tensor([1.12]), near=near, far=far,
                mpp=mpp, nsam=nsam, target=target, az_el=az_el, t_rand=t_rand, u_rand=u_rand)


def _sar_run(m, b):
    cu = lambda xs: [x.cuda() for x in xs]
    m.zero_grad(set_to_none=True)
    pulses, dist, acc = m(b['ray_o'], None, None, b['mfilt'], b['radii'], b['near'], b['far'], b['nsam'], b['mpp'],
                          az_el=b['az_el'], t_rand=cu(b['t_rand']), u_rand=cu(b['u_rand']))
    loss = m.loss_function(pulses, b['target'].cuda())[0]
    loss.backward()
    grads = {k: v.grad.detach().clone() for k, v in m.named_parameters() if v.requires_grad}
    return pulses.detach(), dist.detach(), acc.detach(), float(loss), grads


def test_c4_sar_step_at_reference_sizes():
    """`SARNeRF` forward + backward at the sizes of the reference's SAR configuration (params.yaml:67-93: 2048 rays,
    180 + 180 samples, hidden 156, colour width 180, L = 4/4; 2048 range bins, 8192-point matched filter).

    (1) Against the oracle at 96 rays (what it finishes in seconds), same network sizes, 40 m wavelength (a
        well-conditioned fp32 phase): pulses, distance, acc, loss within the 2e-2 gate of the bf16 MLP.
    (2) At the full 2048 rays the exact-fp32 path (1e-4 parity with the oracle) stands in for it: the bf16 chain
        must give the same pulses, distances, accs and loss to 2e-2 and gradients of the same direction (cosine > 0.995
        over all 28 tensors) and size (5 %).
    (3) At the radar's 3.1 cm wavelength: range-compressed profiles normalised to unit spread per level, acc in
        [0, 1], distances inside [near, far], all 28 gradients finite and non-zero."""
    from oracle import nerf_oracle as O
    from nerf_amd.model import SARNeRF
    from tests.conftest import relerr
    # (1) oracle, reduced ray count
    R = 96
    b = _sar_pulse(R, seed=1)
    cfg = _sar_cfg('bf16', 40.0, R)
    p = O.init_params(cfg, sar=True, seed=7)
    m = SARNeRF(cfg).cuda().train()
    m.load_state_dict(p, strict=False)
    ref_p, ref_d, ref_a, _ = O.sar_render(p, b['ray_o'], b['az_el'], b['mfilt'], 1.12, 900., float(b['far'][0]), b['nsam'],
                                          0.5, cfg, b['t_rand'], [{1: u} for u in b['u_rand']])
    ref_loss = float(O.sar_loss(ref_p, b['target'], cfg.coarse_weight_decay)[0])
    got_p, got_d, got_a, loss, _ = _sar_run(m, b)
    assert relerr(got_a.cpu(), ref_a) < 2e-2 and relerr(got_d.cpu(), ref_d) < 2e-2
    assert relerr(got_p.cpu()[0], ref_p[0]) < 2e-2
    assert abs(loss - ref_loss) < 2e-2 * abs(ref_loss)

    # (2) full size, fp32 path as the stand-in oracle
    R = 2048
    b = _sar_pulse(R, seed=2)
    outs = {}
    for prec in ('fp32', 'bf16'):
        cfg = _sar_cfg(prec, 40.0, R)
        m = SARNeRF(cfg).cuda().train()
        m.load_state_dict(O.init_params(cfg, sar=True, seed=7), strict=False)
        outs[prec] = _sar_run(m, b)
    f, h = outs['fp32'], outs['bf16']
    assert relerr(h[2], f[2]) < 2e-2 and relerr(h[1], f[1]) < 2e-2
    assert relerr(h[0][0], f[0][0]) < 2e-2
    assert abs(h[3] - f[3]) < 2e-2 * abs(f[3])
    gf = torch.cat([f[4][k].reshape(-1) for k in sorted(f[4])])
    gh = torch.cat([h[4][k].reshape(-1) for k in sorted(f[4])])
    cos = float((gf * gh).sum() / (gf.norm() * gh.norm()))
    print('c4 full size: bf16 vs fp32 gradient cosine %.4f, norm ratio %.4f' % (cos, float(gh.norm() / gf.norm())))
    assert cos > 0.995 and abs(float(gh.norm() / gf.norm()) - 1) < 0.05        # measured 0.9998 / 0.990

    # (3) the radar's wavelength
    cfg = _sar_cfg('bf16', 0.031228381041666666, R)
    m = SARNeRF(cfg).cuda().train()
    m.load_state_dict(O.init_params(cfg, sar=True, seed=7), strict=False)
    pulses, dist, acc, loss, grads = _sar_run(m, b)
    assert pulses.shape == (2, 1, 2048, 2) and dist.shape == (2, 1, R) and acc.shape == (2, 1, R)
    z = torch.view_as_complex(pulses.contiguous())
    assert torch.allclose(torch.std(z, dim=-1), torch.ones(2, 1, device='cuda'), atol=1e-4)
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-6
    assert float(dist.min()) >= 900.0 - 1e-3 and float(dist.max()) <= float(b['far'][0]) + 1e-3
    assert loss == loss and len(grads) == 28
    for k, g in grads.items():
        assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0, k


def test_sar_forward_raises_beyond_the_last_range_bin():
    """A far plane beyond the last pulse bin: the reference's indexed write raises IndexError (utils.py:379), and so
    does the model's forward; `check_range = False` drops such rays silently instead."""
    from nerf_amd.model import SARNeRF
    R = 256
    b = _sar_pulse(R, nsam=64, nfft=128, seed=3)
    b['far'], b['mpp'] = b['near'] + 400.0, torch.tensor([0.01])     # the 64 bins end 0.63 m behind the near plane
    b['t_rand'], b['u_rand'] = None, None
    cfg = _sar_cfg('bf16', 0.03, R)
    cfg.randomized = False
    m = SARNeRF(cfg).cuda().train()
    args = (b['ray_o'], None, None, b['mfilt'], b['radii'], b['near'], b['far'], b['nsam'], b['mpp'])
    with pytest.raises(IndexError):
        m(*args, az_el=b['az_el'])
    cfg.check_range = False
    out = m(*args, az_el=b['az_el'])
    assert bool(torch.isfinite(out[1]).all())


def test_fused_step_matches_the_autograd_step():
    """The fixed launch sequence (`nerf_amd.stepper`: compositing + loss + compositing backward in one kernel, one
    backward chain and one weight-gradient pass over both levels) against the autograd path on the same kernels:
    same loss and the same 24 gradients up to the order of fp32 atomic sums, with a ragged mask, white background
    on and off, 4096 rays (whole tiles) and 1000 rays (each level ends in a partly filled tile)."""
    from nerf_amd.model import MipNeRF
    from nerf_amd.train import nerf_train_step, synthetic_rays
    for n, wb in ((4096, False), (1000, True)):
        torch.manual_seed(4)
        c = cfg(False)
        c.white_bkgd = wb
        m = MipNeRF(c).cuda().train()
        rays, pix = synthetic_rays(n, 'cuda', seed=8)
        rays = rays._replace(lossmult=(torch.rand(n, 1, device='cuda') > 0.25).float())
        out = []
        for fused in (False, True):
            opt = torch.optim.SGD(m.parameters(), lr=0.0)
            loss = nerf_train_step(m, rays, pix, opt, chunk=n, fused=fused)
            out.append((float(loss), [p.grad.clone() for p in m.parameters() if p.requires_grad]))
        assert m._stepper is not None
        assert abs(out[0][0] - out[1][0]) < 1e-5 * abs(out[0][0]), (n, out[0][0], out[1][0])
        for a, b in zip(out[0][1], out[1][1]):
            assert torch.isfinite(b).all() and float(a.norm()) > 0
            assert float((a - b).norm()) <= 2e-4 * float(a.norm()), n


def test_fused_step_with_density_noise_matches_the_autograd_step():
    """`density_noise > 0` (model.py:135-136: uniform noise on the raw density between the MLP and the compositing)
    on the fixed launch sequence: with the device generator seeded alike, the fused step draws the jitter and the
    noise in the reference's order and gives the loss and the gradients of the autograd path; and the noise does
    change the result."""
    from nerf_amd.model import MipNeRF
    from nerf_amd.train import nerf_train_step, synthetic_rays
    n = 1000
    torch.manual_seed(4)
    c = cfg(True)
    c.density_noise = 0.7
    m = MipNeRF(c).cuda().train()
    rays, pix = synthetic_rays(n, 'cuda', seed=8)
    out = []
    for fused, noise in ((False, 0.7), (True, 0.7), (True, 0.0)):
        m.density_noise = noise
        opt = torch.optim.SGD(m.parameters(), lr=0.0)
        torch.manual_seed(21)
        loss = nerf_train_step(m, rays, pix, opt, chunk=n, fused=fused)
        out.append((float(loss), [p.grad.clone() for p in m.parameters() if p.requires_grad]))
    assert abs(out[0][0] - out[1][0]) < 1e-5 * abs(out[0][0]), (out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert torch.isfinite(b).all() and float(a.norm()) > 0
        assert float((a - b).norm()) <= 2e-4 * float(a.norm())
    assert abs(out[2][0] - out[1][0]) > 1e-4 * abs(out[1][0])


def test_graphed_sar_step_follows_the_learning_rate_schedule():
    """A graph replay must see a changed learning rate (the reference steps MipLRDecay after every optimizer step,
    model.py:389-393).  A float lr is turned into a device tensor ahead of the capture; three replays at lr 0,
    1e-3 and 2e-3 (through `step(lr=...)`) move the weights by 0, d and about 2 d; and a scheduler handed to
    `step` leaves the tensor in place and writes its rate into it."""
    from bench import make_sar_cfg, sar_batch
    from nerf_amd.model import SARNeRF
    from nerf_amd.train import GraphedSarStep
    from nerf_amd.util_modules import MipLRDecay
    dev = torch.device('cuda')
    cfg_ = make_sar_cfg('bf16')
    cfg_.rays_per_chunk, cfg_.total_num_rays, cfg_.wavelength = 256, 256, 40.0
    torch.manual_seed(3)
    m = SARNeRF(cfg_).to(dev).train()
    # plain SGD-like behaviour of AdamW's first steps: the update is lr * sign-ish(m / sqrt(v)), linear in lr
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.0, fused=True, capturable=True)
    g = GraphedSarStep(m, sar_batch(dev, seed=7), opt, warmup=2)
    assert all(torch.is_tensor(gr['lr']) and gr['lr'].is_cuda for gr in opt.param_groups)
    flat = lambda: torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
    deltas = []
    for lr in (0.0, 1e-3, 2e-3):
        w0 = flat()
        g.step(lr=lr)
        torch.cuda.synchronize()
        deltas.append(float((flat() - w0).norm()))
    assert deltas[0] == 0.0 and deltas[1] > 0, deltas
    assert 1.6 < deltas[2] / deltas[1] < 2.4, deltas
    sched = MipLRDecay(opt, lr_init=1e-2, lr_final=1e-4, max_steps=10, lr_delay_steps=0, lr_delay_mult=1)
    t = opt.param_groups[0]['lr']
    g.step(scheduler=sched)
    assert opt.param_groups[0]['lr'] is t and abs(float(t) - sched.get_last_lr()[0]) < 1e-9
    assert 1e-4 < float(t) < 1e-2
    with pytest.raises(RuntimeError):
        opt.param_groups[0]['lr'] = 1e-3
        g.step(lr=5e-4)


def test_graphed_sar_step_replays_the_eager_step():
    """The SAR pulse step captured as a HIP graph (train.GraphedSarStep) against the same step issued launch by
    launch, from the same weights and the same device-generator seed: the replayed steps give the eager losses
    and leave the same weights (up to the order of fp32 atomic sums)."""
    import copy
    from bench import make_sar_cfg, sar_batch
    from nerf_amd.model import SAR_AZ_SIGMA, SAR_EL_SIGMA, SARNeRF
    from nerf_amd.train import GraphedSarStep, sar_train_step
    dev = torch.device('cuda')
    cfg = make_sar_cfg('bf16')
    cfg.rays_per_chunk, cfg.total_num_rays = 512, 1024                     # two chunks per pulse
    # a long wavelength: at the radar's 3 cm the phase turns by 4e5 rad over the scene and the loss after an
    # optimizer step depends on the last bits of the weights (on the order of the fp32 atomic sums), in either mode
    cfg.wavelength = 40.0
    torch.manual_seed(3)
    m0 = SARNeRF(cfg).to(dev).train()
    batch = sar_batch(dev, seed=7)
    warm, steps = 2, 3

    def run(graphed):
        m = copy.deepcopy(m0)
        opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=cfg.weight_decay, fused=True, capturable=True)
        torch.manual_seed(11)
        losses = []
        if graphed:
            g = GraphedSarStep(m, batch, opt, warmup=warm)
            for _ in range(steps):
                losses.append(float(g.step()))
        else:
            for i in range(warm + 1 + steps):                              # the capture pass itself runs nothing
                if i == warm:
                    continue
                az_el = [(torch.randn(512, device=dev) * SAR_AZ_SIGMA, torch.randn(512, device=dev) * SAR_EL_SIGMA)
                         for _ in range(2)]
                loss = float(sar_train_step(m, batch, opt, az_el=az_el))
                if i > warm:
                    losses.append(loss)
        return losses, torch.cat([p.detach().reshape(-1) for p in m.parameters()])

    le, we = run(False)
    lg, wg = run(True)
    print('eager', le, 'graphed', lg)
    assert all(l == l and l > 0 for l in lg)
    # both runs draw their beams and sampling offsets from the device generator seeded alike, and a replay consumes
    # the generator as the eager step does: the same pulses, up to the order of fp32 atomic sums in the gradients
    for a, b in zip(le, lg):
        assert abs(a - b) < 2e-3 * abs(a), (le, lg)
    w0 = torch.cat([p.detach().reshape(-1) for p in m0.parameters()])
    assert float((wg - we).norm()) < 0.2 * float((we - w0).norm())
