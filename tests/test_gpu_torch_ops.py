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
