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
    """Rays are independent: rendering 262 144 rays at once or 8 192 at a time gives identical pixels
    (deterministic sampling); acc in [0, 1], colour within the padded sigmoid range, distance within [near, far]."""
    from nerf_amd.model import MipNeRF
    from nerf_amd.train import synthetic_rays
    from nerf_amd.utils import namedtuple_map
    torch.manual_seed(1)
    m = MipNeRF(cfg(False)).cuda().train()
    m.randomized = False
    rays, _ = synthetic_rays(262144, 'cuda', seed=6)
    with torch.no_grad():
        rgb, dist, acc = m(rays)
        parts = [m(namedtuple_map(lambda t: t[i:i + 8192], rays)) for i in range(0, 262144, 65536)]
    for i, (r, d, a) in enumerate(parts):
        s = slice(i * 65536, i * 65536 + 8192)
        assert torch.equal(r, rgb[:, s]) and torch.equal(d, dist[:, s]) and torch.equal(a, acc[:, s])
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
