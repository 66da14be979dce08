"""The `torch.library` custom ops (nerf_amd.torch_ops): registration checks, equality with the autograd.Function
wrappers the models use, and a fullgraph `torch.compile` of a small render (no graph break at the kernels)."""
import pytest
import torch

from oracle import nerf_oracle as O


def _inputs(n=37, s=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    rgb = torch.rand(n, s, 3, generator=g)
    density = torch.rand(n, s, generator=g) * 3
    near, far = torch.full((n, 1), 2.0), torch.full((n, 1), 6.0)
    t = O.stratified_t(near, far, s, torch.rand(n, s + 1, generator=g))
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    return rgb, density, t, d


def test_ops_are_registered_and_have_fake_kernels():
    """CPU: shapes through the fake (meta) implementations, as a tracing compiler sees the ops."""
    import nerf_amd.torch_ops  # noqa: F401
    n, s = 5, 8
    m = lambda *shape: torch.empty(*shape, device='meta')
    comp, dist, acc, w = torch.ops.nerf_amd.composite(m(n, s, 3), m(n, s), m(n, s + 1), m(n, 3), False, False, 0.0, 0.0)
    assert comp.shape == (n, 3) and dist.shape == (n,) and acc.shape == (n,) and w.shape == (n, s)
    assert torch.ops.nerf_amd.stratified_t(m(n, 1), m(n, 1), s, None, False).shape == (n, s + 1)
    assert torch.ops.nerf_amd.cast_ipe(m(n, s + 1), m(n, 3), m(n, 3), m(n, 1), 0, 0, 10).shape == (n, s, 60)
    assert torch.ops.nerf_amd.resample(m(n, s + 1), m(n, s), 2 * s + 1, 0.01, None, True).shape == (n, 2 * s + 1)
    g_rgb, g_den = torch.ops.nerf_amd.composite_bwd(m(n, s, 3), m(n, s), m(n, s + 1), m(n, 3), m(n, 3), m(n), m(n), False,
                                                    False, 0.0, 0.0)
    assert g_rgb.shape == (n, s, 3) and g_den.shape == (n, s)
    # the fused MLP: fragment encoders, forward / backward chains, weight gradients (8x256 network, MX operand copies)
    from nerf_amd import mlp_bf16
    geom = mlp_bf16.Geom(60, 27, 256, 128, False, 1)
    names = geom.param_names()
    assert len(names) == 24 and names[0] == 'density_net0.0.weight' and 'rgb_net1.0.bias' in names
    assert len(mlp_bf16.Geom(60, 0, 256, 0, False, 0).param_names()) == 20
    assert len(mlp_bf16.Geom(24, 27, 156, 180, True, 1).param_names()) == 28
    M = n * s
    nblk = mlp_bf16.n_blocks(M)
    enc = torch.ops.nerf_amd.encode_frag(m(n, s + 1), m(n, 3), m(n, 3), m(n), 0, 0, 10)
    assert enc.dtype == torch.uint8 and enc.numel() == nblk * 4 * 2 * 1024
    view = torch.ops.nerf_amd.view_frag(m(n, 3), 0, 4)
    assert view.numel() == n * 2 * 2 * 16
    w = [m(3, 3) for _ in names]
    d, rgb, ang, ln, acts = torch.ops.nerf_amd.mlp_fwd(w, enc, view, geom.ints(), M, s, n, True)
    assert d.shape == (M, 1) and rgb.shape == (M, 3) and ang.shape[0] == 0 and ln.shape[0] == 0
    want = [nblk * b if b is not None else 0 for b in mlp_bf16.act_block_bytes(geom)]
    assert [a.numel() for a in acts] == want and want[8] == 0 and want[0] == nblk * 8 * 2048
    assert torch.ops.nerf_amd.mlp_fwd(w, enc, view, geom.ints(), M, s, n, False)[4] == []
    dys = torch.ops.nerf_amd.mlp_bwd(w, geom.ints(), M, m(M, 1), m(M, 3), ang, d, rgb, ang, ln, acts)
    assert [x.numel() for x in dys] == [nblk * b if b is not None else 0 for b in mlp_bf16.dy_block_bytes(geom)]
    grads = torch.ops.nerf_amd.mlp_wgrad(w, geom.ints(), M, acts, dys)
    assert len(grads) == 24 and all(g.shape == (3, 3) for g in grads)


@pytest.mark.gpu
def test_composite_op_equals_the_autograd_wrapper_and_passes_opcheck():
    from nerf_amd import ops, torch_ops
    rgb, density, t, d = (x.cuda() for x in _inputs())
    a = [rgb.clone().requires_grad_(True), density.clone().requires_grad_(True)]
    b = [rgb.clone().requires_grad_(True), density.clone().requires_grad_(True)]
    c1, d1, a1, w1 = torch_ops.composite(a[0], a[1], t, d, False, False, 0.0, 0.0)
    c2, d2, a2, w2, _ = ops.composite(b[0], b[1], t, d, False)
    for x, y in ((c1, c2), (d1, d2), (a1, a2), (w1, w2)):
        assert torch.equal(x, y)
    gc, gd, ga = torch.randn_like(c1), torch.randn_like(d1), torch.randn_like(a1)
    ((c1 * gc).sum() + (d1 * gd).sum() + (a1 * ga).sum()).backward()
    ((c2 * gc).sum() + (d2 * gd).sum() + (a2 * ga).sum()).backward()
    assert torch.equal(a[0].grad, b[0].grad) and torch.equal(a[1].grad, b[1].grad)
    args = (rgb.clone().requires_grad_(True), density.clone().requires_grad_(True), t, d, False, False, 0.0, 0.0)
    torch.library.opcheck(torch.ops.nerf_amd.composite.default, args,
                          test_utils=('test_schema', 'test_autograd_registration', 'test_faketensor'))
    torch.library.opcheck(torch.ops.nerf_amd.stratified_t.default, (t[:, :1].contiguous(), t[:, -1:].contiguous(), 16, None, False),
                          test_utils=('test_schema', 'test_faketensor'))


def _small_model(**kw):
    from types import SimpleNamespace
    from nerf_amd.model import MipNeRF
    c = dict(use_viewdirs=True, randomized=True, ray_shape='cone', white_bkgd=False, num_levels=2, num_samples=16,
             level_samples=[16, 16], min_deg=0, max_deg=4, viewdirs_min_deg=0, viewdirs_max_deg=4, density_noise=0.0,
             rgb_padding=0.001, resample_padding=0.01, density_bias=-1.0, hidden=64, coarse_weight_decay=0.1,
             lr_init=1e-3, lr_final=5e-5, max_steps=100, lr_delay_steps=0, lr_delay_mult=1, weight_decay=1e-5,
             mlp_precision='bf16', wgrad_operands='bf16')
    c.update(kw)
    torch.manual_seed(7)
    return MipNeRF(SimpleNamespace(**c)).cuda().train()


def _small_rays(n, seed=3):
    from nerf_amd.utils import Rays
    g = torch.Generator().manual_seed(seed)
    o = 0.1 * torch.randn(n, 3, generator=g) + torch.tensor([0., 0., 4.])
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    rays = Rays(o, d, d.clone(), torch.full((n, 1), 0.005), torch.ones(n, 1), torch.full((n, 1), 2.), torch.full((n, 1), 6.))
    draws = dict(t_rand=torch.rand(n, 17, generator=g).cuda(),
                 u_rand={1: (torch.rand(n, 17, generator=g) * (1 / 17 - 1.2e-7)).cuda()})
    return Rays(*(x.cuda() for x in rays)), torch.rand(n, 3, generator=g).cuda(), draws


@pytest.mark.gpu
@pytest.mark.parametrize('kw', [dict(), dict(use_viewdirs=False), dict(wgrad_operands='mx', hidden=256, num_samples=128, max_deg=10,
                                                                        level_samples=[16, 16])])
def test_mlp_ops_equal_the_eager_wrappers_and_pass_opcheck(kw):
    """The MLP of model.py:117-132 as custom ops (encode_frag, view_frag, mlp_fwd with its registered backward through
    mlp_bwd and mlp_wgrad): a two-level render + loss through `MipNeRF._forward_custom_ops` gives the outputs of the
    eager wrappers bit for bit and the same 24 (20) gradients up to the order of fp32 atomic sums; opcheck passes."""
    from nerf_amd import mlp_bf16, torch_ops  # noqa: F401
    m = _small_model(**kw)
    rays, target, draws = _small_rays(53)
    out = []
    for custom in (False, True):
        m.config.use_custom_ops = custom
        m.zero_grad(set_to_none=True)
        rgb, dist, acc = m(rays, **draws)
        loss, _ = m.loss_function(rgb, target, rays.lossmult)
        loss.backward()
        out.append((rgb.detach(), dist.detach(), acc.detach(), {k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad}))
    for a, b in zip(out[0][:3], out[1][:3]):
        assert torch.equal(a, b)
    assert len(out[1][3]) == (24 if m.use_viewdirs else 20)
    for k, g in out[0][3].items():
        assert float((g - out[1][3][k]).norm()) <= 2e-4 * float(g.norm()) + 1e-9, k
    # registration checks on one level's call
    geom = mlp_bf16.Geom.of_model(m)
    params = dict(m.named_parameters())
    weights = [params[k].detach().clone().requires_grad_(True) for k in geom.param_names()]
    S, n = 16, 53
    t = torch.ops.nerf_amd.stratified_t(rays.near, rays.far, S, draws['t_rand'], False)
    enc = torch.ops.nerf_amd.encode_frag(t, rays.origins, rays.directions, rays.radii.reshape(-1), 0, m.config.min_deg, m.config.max_deg)
    view = torch.ops.nerf_amd.view_frag(rays.viewdirs, 0, 4) if m.use_viewdirs else None
    torch.library.opcheck(torch.ops.nerf_amd.mlp_fwd.default, (weights, enc, view, geom.ints(), n * S, S, n, True),
                          test_utils=('test_schema', 'test_autograd_registration', 'test_faketensor'))
    torch.library.opcheck(torch.ops.nerf_amd.encode_frag.default,
                          (t, rays.origins, rays.directions, rays.radii.reshape(-1), 0, m.config.min_deg, m.config.max_deg),
                          test_utils=('test_schema', 'test_faketensor'))
    d, rgb, ang, ln, acts = torch.ops.nerf_amd.mlp_fwd(weights, enc, view, geom.ints(), n * S, S, n, True)
    gd, gc = torch.randn_like(d), torch.randn_like(rgb)
    torch.library.opcheck(torch.ops.nerf_amd.mlp_bwd.default, (weights, geom.ints(), n * S, gd, gc, ang, d.detach(), rgb.detach(), ang, ln, acts),
                          test_utils=('test_schema', 'test_faketensor'))
    dys = torch.ops.nerf_amd.mlp_bwd(weights, geom.ints(), n * S, gd, gc, ang, d.detach(), rgb.detach(), ang, ln, acts)
    torch.library.opcheck(torch.ops.nerf_amd.mlp_wgrad.default, (weights, geom.ints(), n * S, acts, dys),
                          test_utils=('test_schema', 'test_faketensor'))


@pytest.mark.gpu
def test_two_level_model_forward_and_backward_compile_as_one_graph():
    """The whole coarse + fine `MipNeRF.forward` (model.py:100-151) with its loss under
    `torch.compile(fullgraph=True)`: sampling, encoders, the MLP chain, resampling and compositing are opaque nodes
    of ONE graph (no graph break at any kernel), and the compiled forward and backward equal eager mode."""
    from nerf_amd import torch_ops  # noqa: F401
    m = _small_model()
    rays, target, draws = _small_rays(41, seed=5)

    def step(rays, target, t_rand, u_rand1):
        rgb, dist, acc = m(rays, t_rand=t_rand, u_rand={1: u_rand1})
        return m.loss_function(rgb, target, rays.lossmult)[0], rgb

    loss_e, rgb_e = step(rays, target, draws['t_rand'], draws['u_rand'][1])
    loss_e.backward()
    eager = {k: p.grad.clone() for k, p in m.named_parameters() if p.requires_grad}
    m.zero_grad(set_to_none=True)
    loss_c, rgb_c = torch.compile(step, backend='aot_eager', fullgraph=True)(rays, target, draws['t_rand'], draws['u_rand'][1])
    loss_c.backward()
    assert torch.equal(rgb_c, rgb_e) and torch.allclose(loss_c, loss_e, rtol=1e-6)
    for k, p in m.named_parameters():
        if p.requires_grad:
            assert float((p.grad - eager[k]).norm()) <= 2e-4 * float(eager[k].norm()) + 1e-9, k
    # without supplied draws the jitter comes from torch.rand inside the graph: still one graph
    out = torch.compile(lambda r: m(r)[0], backend='aot_eager', fullgraph=True)(rays)
    assert out.shape == (2, 41, 3) and bool(torch.isfinite(out).all())


@pytest.mark.gpu
def test_composite_weights_carry_no_gradient():
    """The weights returned by the composite op are marked non-differentiable (the reference consumes them under
    no_grad, model.py:104-111; no gradient is implemented for them)."""
    from nerf_amd import torch_ops
    rgb, density, t, d = (x.cuda() for x in _inputs())
    out = torch_ops.composite(rgb.requires_grad_(True), density.requires_grad_(True), t, d, False, False, 0.0, 0.0)
    assert out[0].requires_grad and not out[3].requires_grad


@pytest.mark.gpu
def test_small_render_compiles_as_one_graph():
    """sample -> cast + IPE -> a linear stand-in for the MLP -> composite -> loss under torch.compile(fullgraph=True):
    the kernels are opaque nodes of ONE graph, forward and backward equal eager mode."""
    from nerf_amd import torch_ops  # noqa: F401
    n, s = 64, 16
    g = torch.Generator().manual_seed(3)
    o = (0.1 * torch.randn(n, 3, generator=g) + torch.tensor([0., 0., 4.])).cuda()
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).cuda()
    radii, near, far = torch.full((n, 1), 0.005).cuda(), torch.full((n, 1), 2.0).cuda(), torch.full((n, 1), 6.0).cuda()
    t_rand = torch.rand(n, s + 1, generator=g).cuda()
    target = torch.rand(n, 3, generator=g).cuda()

    def render(w):
        t = torch.ops.nerf_amd.stratified_t(near, far, s, t_rand, False)
        enc = torch.ops.nerf_amd.cast_ipe(t, o, d, radii, 0, 0, 4)
        raw = enc @ w                                               # [n, s, 4]
        comp, dist, acc, wts = torch.ops.nerf_amd.composite(torch.sigmoid(raw[..., :3]).contiguous(),
                                                            torch.nn.functional.softplus(raw[..., 3]).contiguous(),
                                                            t, d, False, False, 0.0, 0.0)
        return ((comp - target) ** 2).mean() + 0.01 * dist.mean()

    w1 = (0.3 * torch.randn(24, 4, generator=g)).cuda().requires_grad_(True)
    w2 = w1.detach().clone().requires_grad_(True)
    eager = render(w1)
    eager.backward()
    compiled = torch.compile(render, backend='aot_eager', fullgraph=True)(w2)
    compiled.backward()
    assert torch.allclose(compiled, eager, rtol=1e-6, atol=1e-7)
    assert torch.allclose(w2.grad, w1.grad, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_scatter_op_equals_the_autograd_wrapper():
    """The SAR range-bin scatter (utils.py:321-380) through the custom op: outputs and gradients equal ops.scatter's."""
    from nerf_amd import ops, torch_ops
    r, s, nsam = 96, 32, 256
    g = torch.Generator().manual_seed(5)
    rgb, ang = torch.rand(r, s, 3, generator=g).cuda(), torch.rand(r, s, 2, generator=g).cuda()
    den = (torch.rand(r, s, generator=g) * 2).cuda()
    power = torch.rand(r, generator=g).cuda()
    near, far = torch.full((r, 1), 900.0), torch.full((r, 1), 1000.0)
    t = O.stratified_t(near, far, s, torch.rand(r, s + 1, generator=g)).cuda()
    d = torch.nn.functional.normalize(torch.randn(r, 3, generator=g), dim=-1).cuda()
    bins = (900.0 + 0.5 * torch.arange(nsam)).cuda()
    a = [x.clone().requires_grad_(True) for x in (rgb, ang, den)]
    b = [x.clone().requires_grad_(True) for x in (rgb, ang, den)]
    p1, d1, a1, w1, rb1, _, st1 = torch_ops.scatter(a[0], a[1], a[2], power, t, d, bins, 40.0, False, 0.0, 0.0)
    p2, d2, a2, w2, _, rb2, st2 = ops.scatter(b[0], b[1], b[2], power, t, d, bins, 40.0)
    for x, y in ((p1, p2), (d1, d2), (a1, a2), (w1, w2), (rb1, rb2), (st1, st2)):
        assert torch.equal(x, y)
    gp = torch.randn_like(p1)
    (p1 * gp).sum().backward()
    (p2 * gp).sum().backward()
    for x, y in zip(a, b):
        assert torch.equal(x.grad, y.grad)
    torch.library.opcheck(torch.ops.nerf_amd.scatter.default,
                          (rgb.clone().requires_grad_(True), ang, den, power, t, d, bins, 40.0, False, 0.0, 0.0),
                          test_utils=('test_schema', 'test_autograd_registration', 'test_faketensor'))


@pytest.mark.gpu
def test_matmul_small_fixed_order_product():
    """nerf_matmul_small_f32 (the weight-fold products): strided / transposed operands, bias, accumulation; fp64 torch as
    the reference (1e-6), and the same bits on repeated calls."""
    from nerf_amd.mlp import matmul_small
    g = torch.Generator().manual_seed(11)
    m, n, k = 37, 53, 129
    A = torch.randn(m, k + 5, generator=g).cuda()            # row stride k + 5
    Bt = torch.randn(n, k, generator=g).cuda()               # B given transposed
    bias = torch.randn(n, generator=g).cuda()
    want = (A[:, :k].double() @ Bt.double().t() + bias.double()).float()
    out = matmul_small(A, (k + 5, 1), Bt, (1, k), m, n, k, bias=bias)
    assert float((out - want).abs().max() / want.abs().max()) < 1e-6
    assert torch.equal(out, matmul_small(A, (k + 5, 1), Bt, (1, k), m, n, k, bias=bias))
    big = torch.ones(m, n + 3, device='cuda')                # accumulate into a wider matrix's first n columns
    matmul_small(A, (k + 5, 1), Bt, (1, k), m, n, k, out=big, accumulate=True)
    assert float((big[:, :n] - 1 - (want - bias)).abs().max() / want.abs().max()) < 1e-6 and bool((big[:, n:] == 1).all())
