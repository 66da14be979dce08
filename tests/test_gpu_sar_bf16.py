"""SAR head on the fused bf16 MFMA chain (LayerNorm density head, rgb + angle heads) against the CPU oracle.
Tolerance 2e-2 (bf16 MLP gate)."""
import pytest
import torch

from oracle import nerf_oracle as O
from tests.conftest import relerr
from tests.test_gpu_mlp_bf16 import make_cfg

pytestmark = pytest.mark.gpu

SAR_CASES = [('small_metres', dict(hidden=64, num_samples=32, max_deg=4), 32, 128, True),
             ('small', dict(hidden=64, num_samples=32, max_deg=4), 32, 96, False),
             ('c4', dict(hidden=156, num_samples=180, max_deg=4), 180, 24, False),
             # use_viewdirs=False (model.py:283-302): density, rgb and angle heads all on the trunk output
             ('small_noview', dict(hidden=64, num_samples=32, max_deg=4, use_viewdirs=False), 32, 96, False),
             ('c4_noview', dict(hidden=156, num_samples=180, max_deg=4, use_viewdirs=False), 180, 24, False)]


def sar_inputs(cfg, S, R, seed=3, metres=True):
    """metres: the survey's slant-range geometry (the integrated encoding is then almost fully damped and the
    network sees a near-constant input); otherwise a unit-scale scene whose encoding varies from sample to sample."""
    g = torch.Generator().manual_seed(seed)
    az, el = torch.randn(R, generator=g) * O.SAR_AZ_SIGMA, torch.randn(R, generator=g) * O.SAR_EL_SIGMA
    if not metres:
        az, el = az * 40, el * 40
    ray_d, ray_p = O.sar_rays(az, el)
    origins = torch.tensor([0., -1000., 500.] if metres else [0.1, -4., 0.3]).expand(R, 3).contiguous()
    near, far = torch.full((R, 1), 900. if metres else 2.), torch.full((R, 1), 1900. if metres else 6.)
    radii = torch.full((R, 1), 1.12 if metres else 0.005)
    t = O.stratified_t(near, far, S, torch.rand(R, S + 1, generator=g))
    return origins, ray_d, ray_p, radii, t


def sar_model(cfg, seed=5):
    from nerf_amd.model import SARNeRF
    cfg.wavelength, cfg.total_num_rays, cfg.rays_per_chunk = 0.03, 128, 128
    p = O.init_params(cfg, sar=True, seed=seed)
    m = SARNeRF(cfg).cuda()
    m.load_state_dict(p, strict=False)
    return m, p


@pytest.mark.parametrize('tag,kw,S,R,metres', SAR_CASES)
def test_sar_mlp_forward_vs_oracle(tag, kw, S, R, metres):
    from nerf_amd import mlp_bf16
    cfg = make_cfg(**kw)
    m, p = sar_model(cfg)
    origins, ray_d, ray_p, radii, t = sar_inputs(cfg, S, R, metres=metres)
    mean, cov = O.cast_rays(t, origins, ray_d, radii, 'cone')
    enc = O.ipe(mean, cov, cfg.min_deg, cfg.max_deg).reshape(R * S, -1)
    rows = O.sar_view_rows(ray_d, S, cfg) if cfg.use_viewdirs else None      # row i takes ray i mod R (model.py:351-354)
    ref_d, ref_rgb, ref_ang = O.mlp_sar(p, enc, rows)

    pk = mlp_bf16.packed_for(m, need_bwd=False)
    assert pk.sar and pk.view == cfg.use_viewdirs
    encf = mlp_bf16.encode_frag(t.cuda(), origins.cuda(), ray_d.cuda(), radii.cuda(), 'cone', cfg.min_deg, cfg.max_deg)
    viewf = mlp_bf16.view_frag(ray_d.cuda(), 0, 4) if cfg.use_viewdirs else None
    for save in (False, True):
        d, rgb, ang, acts, ln = mlp_bf16.mlp_forward(pk, encf, viewf, R * S, R, R, save=save)
        assert relerr(d.cpu(), ref_d[:, 0]) < 2e-2, (tag, save)
        assert float((rgb.cpu() - ref_rgb).abs().max()) < 2e-2, (tag, save)
        assert float((ang.cpu() - ref_ang).abs().max()) < 2e-2, (tag, save)


def fro(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize('tag,kw,S,R,metres', SAR_CASES[1:])
def test_sar_mlp_gradients(tag, kw, S, R, metres):
    """All 28 parameter gradients of the SAR MLP on the bf16 chain against oracle autograd.  The upstream
    gradient is smooth (a fixed random projection of the outputs), as it is behind the scatter integral; the
    LayerNorm pair and the density weight are checked tightest, since they are assembled from three reductions."""
    from nerf_amd import mlp_bf16
    cfg = make_cfg(**kw)
    m, p = sar_model(cfg)
    p = {k: v.requires_grad_(True) for k, v in p.items()}
    origins, ray_d, ray_p, radii, t = sar_inputs(cfg, S, R, metres=metres)
    mean, cov = O.cast_rays(t, origins, ray_d, radii, 'cone')
    enc = O.ipe(mean, cov, cfg.min_deg, cfg.max_deg).reshape(R * S, -1)
    ref = O.mlp_sar(p, enc, O.sar_view_rows(ray_d, S, cfg) if cfg.use_viewdirs else None)
    g = torch.Generator().manual_seed(9)
    # smooth upstream gradients: low-frequency functions of the sample index, not white noise (white noise makes
    # every ReLU flip of the bf16 forward visible and says nothing about the kernels)
    x = torch.linspace(0, 1, R * S)[:, None]
    ups = [torch.cos(3 * x + i) * (0.5 + torch.rand(1, c, generator=g)) for i, c in enumerate((1, 3, 2))]
    sum((o * u).sum() for o, u in zip(ref, ups)).backward()

    viewf = mlp_bf16.view_frag(ray_d.cuda(), 0, 4) if cfg.use_viewdirs else None
    out = mlp_bf16.query_sar(m, t.cuda(), origins.cuda(), ray_d.cuda(), radii.cuda(), viewf, R, S)
    sum((o * u.cuda()).sum() for o, u in zip(out, ups)).backward()
    worst, cos = {}, {}
    for k, v in m.named_parameters():
        if k.endswith('scales'):
            continue
        a, b = v.grad.cpu(), p[k].grad
        worst[k] = fro(a, b)
        cos[k] = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
    print({k: (round(worst[k], 4), round(cos[k], 4)) for k in worst})
    # trunk layers: a bf16 forward flips some ReLU decisions against the fp32 oracle, which moves single
    # elements by several per cent whatever the kernels do -> direction and size of each tensor
    for k in worst:
        assert cos[k] > 0.98 and worst[k] < 0.15, (k, worst[k], cos[k])
    # heads and LayerNorm pair: straight reductions of the upstream gradient against saved activations
    # (the operands of these reductions are 6-bit copies with three mantissa bits, and this test's upstream gradient is
    # smooth: the 32 samples of a scale block hold nearly the same value and round the same way, so the rounding does
    # not average out the way it does behind a render; measured up to 3.3 %)
    for k in ('final_density.0.weight', 'final_density.0.bias', 'final_density.1.weight', 'final_density.1.bias',
              'final_rgb.0.weight', 'final_rgb.0.bias', 'final_angle.0.weight', 'final_angle.0.bias'):
        assert worst[k] < 4e-2, (k, worst[k])


@pytest.mark.parametrize('tag', ['rand', 'det_noview'])
def test_sar_end_to_end_bf16(golden, tag):
    """SARNeRF.forward on the bf16 chain against the oracle at a long wavelength (well-conditioned fp32 phase):
    pulses, distance, acc and the loss within the bf16 gate; gradients finite and aligned with the oracle's."""
    from tests.conftest import T, cfg_from
    fx = golden('sar_e2e_' + tag)
    cfg = cfg_from(fx)
    cfg.wavelength = 40.0
    cfg.mlp_precision = 'bf16'
    from nerf_amd.model import SARNeRF
    m = SARNeRF(cfg).cuda().train()
    p = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, sar=True, seed=int(fx['wseed'])).items()}
    m.load_state_dict({k: v.detach() for k, v in p.items()}, strict=False)
    chunks = cfg.total_num_rays // cfg.rays_per_chunk
    az_el = [(T(fx[f'az_{c}']), T(fx[f'el_{c}'])) for c in range(chunks)]
    t_rand = [T(fx[f't_rand_{c}']) for c in range(chunks)] if cfg.randomized else None
    u_rand = [T(fx[f'u_rand_{c}']) for c in range(chunks)] if cfg.randomized else None
    mfilt = torch.complex(T(fx['mfilt_re']), T(fx['mfilt_im']))
    nsam = int(fx['nsam'])
    ref_p, ref_d, ref_a, _ = O.sar_render(p, T(fx['ray_o']), az_el, mfilt, float(fx['radii'][0]), float(fx['near'][0]),
                                          float(fx['far'][0]), nsam, float(fx['mpp'][0]), cfg, t_rand,
                                          [{1: u} for u in u_rand] if u_rand is not None else None)
    cu = lambda xs: None if xs is None else [x.cuda() for x in xs]
    got_p, got_d, got_a = m(T(fx['ray_o']), None, None, mfilt, T(fx['radii']), T(fx['near']), T(fx['far']), nsam,
                            T(fx['mpp']), az_el=az_el, t_rand=cu(t_rand), u_rand=cu(u_rand))
    assert relerr(got_a.cpu(), ref_a) < 2e-2
    # the fine level's fenceposts come from the coarse weights, so its distance moves with them
    assert relerr(got_d.cpu()[0], ref_d[0]) < 1e-3 and relerr(got_d.cpu(), ref_d) < 2e-2
    assert relerr(got_p.cpu()[0], ref_p[0]) < 2e-2
    tgt = T(fx['target'])
    O.sar_loss(ref_p, tgt, cfg.coarse_weight_decay)[0].backward()
    loss = m.loss_function(got_p, tgt.cuda())[0]
    loss.backward()
    gk = torch.cat([v.grad.reshape(-1).cpu() for k, v in m.named_parameters() if not k.endswith('scales')])
    go = torch.cat([p[k].grad.reshape(-1) for k, v in m.named_parameters() if not k.endswith('scales')])
    assert bool(torch.isfinite(gk).all())
    assert float((gk * go).sum() / (gk.norm() * go.norm())) > 0.9


def _sar_rank_worker(rank, world, port, q):
    """One rank of the two-process SAR step (both ranks share cuda:0; gloo carries the CUDA tensors)."""
    import os
    import numpy as np
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    out = _sar_step(rank, world)
    q.put((rank,) + tuple(np.ascontiguousarray(x) for x in out))
    dist.barrier()
    dist.destroy_process_group()


def _sar_step(rank, world):
    import os
    import numpy as np
    from tests.conftest import T, cfg_from
    from nerf_amd.model import SARNeRF
    from nerf_amd.train import sar_train_step
    fx = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'sar_e2e_rand.npz'))
    cfg = cfg_from(fx)
    cfg.wavelength = 40.0
    cfg.mlp_precision = 'bf16'
    m = SARNeRF(cfg).cuda().train()
    m.load_state_dict(O.init_params(cfg, sar=True, seed=int(fx['wseed'])), strict=False)
    chunks = cfg.total_num_rays // cfg.rays_per_chunk
    draws = dict(az_el=[(T(fx[f'az_{c}']), T(fx[f'el_{c}'])) for c in range(chunks)],
                 t_rand=[T(fx[f't_rand_{c}']).cuda() for c in range(chunks)],
                 u_rand=[T(fx[f'u_rand_{c}']).cuda() for c in range(chunks)])
    mfilt = torch.complex(T(fx['mfilt_re']), T(fx['mfilt_im']))
    batch = (T(fx['ray_o']), None, None, mfilt, T(fx['radii']), T(fx['near']), T(fx['far']), T(fx['mpp']),
             T(fx['target']))
    opt = torch.optim.SGD(m.parameters(), lr=0.0)            # keep the weights: gradients are what is compared
    loss = sar_train_step(m, batch, opt, rank=rank, world_size=world, **draws)
    grads = torch.cat([p.grad.reshape(-1) for k, p in m.named_parameters() if p.requires_grad])
    return np.array([float(loss), chunks]), grads.cpu().numpy()


def test_sar_chunks_dealt_to_two_ranks_match_one_device():
    """The SAR step with its ray chunks dealt to two processes (range profiles and gradient bucket all-reduced)
    gives the loss and the weight gradients of the one-device loop over the same draws."""
    import socket
    import torch.multiprocessing as mp
    one_loss, one_grads = _sar_step(0, 1)
    assert one_loss[1] >= 2, 'fixture must hold at least two ray chunks'
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sar_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = []
    try:
        for _ in range(2):
            got.append(q.get(timeout=150))
    finally:
        for pr in procs:
            pr.join(30)
            if pr.is_alive():
                pr.kill()
    assert all(pr.exitcode == 0 for pr in procs)
    import numpy as np
    for _, loss, grads in got:
        assert abs(loss[0] - one_loss[0]) <= 1e-4 * abs(one_loss[0])
        assert np.isfinite(grads).all()
        # chunk order changes only the order of fp32 sums (profiles, split-K atomics)
        assert np.linalg.norm(grads - one_grads) <= 2e-3 * np.linalg.norm(one_grads)
    assert np.array_equal(got[0][2], got[1][2])


@pytest.mark.parametrize('tag', ['rand', 'det_noview'])
def test_fused_sar_step_matches_the_autograd_step(golden, tag):
    """The SAR pulse as a fixed launch sequence (`nerf_amd.stepper.SarStepper`: one backward chain and one
    weight-gradient pass over all chunks and levels) against the autograd path on the same kernels and draws: the
    same loss and the same gradients up to the order of fp32 atomic sums."""
    from tests.conftest import T, cfg_from
    from nerf_amd.model import SARNeRF
    from nerf_amd.train import sar_train_step
    fx = golden('sar_e2e_' + tag)
    cfg = cfg_from(fx)
    cfg.wavelength = 40.0
    cfg.mlp_precision = 'bf16'
    m = SARNeRF(cfg).cuda().train()
    m.load_state_dict(O.init_params(cfg, sar=True, seed=int(fx['wseed'])), strict=False)
    chunks = cfg.total_num_rays // cfg.rays_per_chunk
    draws = dict(az_el=[(T(fx[f'az_{c}']), T(fx[f'el_{c}'])) for c in range(chunks)])
    if cfg.randomized:
        draws.update(t_rand=[T(fx[f't_rand_{c}']).cuda() for c in range(chunks)],
                     u_rand=[T(fx[f'u_rand_{c}']).cuda() for c in range(chunks)])
    mfilt = torch.complex(T(fx['mfilt_re']), T(fx['mfilt_im']))
    batch = (T(fx['ray_o']), None, None, mfilt, T(fx['radii']), T(fx['near']), T(fx['far']), T(fx['mpp']), T(fx['target']))
    out = []
    for fused in (False, True):
        opt = torch.optim.SGD(m.parameters(), lr=0.0)
        loss = sar_train_step(m, batch, opt, fused=fused, **draws)
        out.append((float(loss), {k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad}))
    assert getattr(m, '_sar_stepper', None) is not None
    assert abs(out[0][0] - out[1][0]) < 1e-5 * abs(out[0][0]), (out[0][0], out[1][0])
    scale = max(float(g.norm()) for g in out[0][1].values())
    for k, a in out[0][1].items():
        b = out[1][1][k]
        assert torch.isfinite(b).all(), k
        assert float((a - b).norm()) <= 2e-4 * float(a.norm()) + 1e-6 * scale, (k, float((a - b).norm()), float(a.norm()))
