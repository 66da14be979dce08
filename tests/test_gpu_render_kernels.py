"""HIP sampling / resampling / compositing kernels (through the C ABI) against the golden
fixtures captured from the reference and against the CPU oracle on seeded inputs.

Tolerances: elementwise fp32 stages 1e-6 relative (libm differences only); sums 1e-5;
interval indices bit-exact on identical cdf/u inputs.
"""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from tests.conftest import T, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    from nerf_amd import ops as _ops
    return _ops


def G(a):
    return T(a).cuda()


def close(a, b, rtol=1e-6, atol=1e-7):
    a = a.detach().cpu()
    b = torch.as_tensor(b).detach().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, rtol=rtol, atol=atol), float((a - b).abs().max())


def test_stratified_and_cast_golden(ops, golden):
    fx = golden('sample')
    S = int(fx['S'])
    o, d, r, near, far = (G(fx[k]) for k in ('origins', 'directions', 'radii', 'near', 'far'))
    for shape in ('cone', 'cylinder'):
        for rnd in (0, 1):
            for lin in (0, 1):
                key = f'{shape}_{rnd}_{lin}'
                t = ops.stratified_t(near, far, S, G(fx['trand_' + key]) if rnd else None, bool(lin))
                close(t, fx['t_' + key], 1e-6, 0)
                mean, cov = ops.cast_rays(G(fx['t_' + key]), o, d, r, shape)
                close(mean, fx['mean_' + key], 1e-6, 1e-6)
                close(cov, fx['cov_' + key], 1e-5, 1e-9)


def test_stratified_bit_exact_vs_oracle(ops):
    g = torch.Generator().manual_seed(1)
    n, S = 300, 64
    near = torch.rand(n, 1, generator=g) * 2 + 1
    far = near + torch.rand(n, 1, generator=g) * 5 + 0.5
    tr = torch.rand(n, S + 1, generator=g)
    for rnd in (False, True):
        ref = O.stratified_t(near, far, S, tr if rnd else None)
        got = ops.stratified_t(near.cuda(), far.cuda(), S, tr.cuda() if rnd else None).cpu()
        # identical up to torch.linspace's vector-width dependent last bit
        assert float((got - ref).abs().max()) <= 1e-6
        assert (got[:, 1:] >= got[:, :-1]).all()


def test_posenc_golden(ops, golden):
    fx = golden('posenc')
    mean, cov = G(fx['mean']), G(fx['cov'])
    for L in (4, 10, 16):
        close(ops.ipe_encode(mean, cov, 0, L), fx[f'ipe_{L}'], 0, 2e-6)
    close(ops.ipe_encode(mean, cov, 1, 5), fx['ipe_1_5'], 0, 2e-6)
    # the module returns (encoding, variance) like the reference's (util_modules.py:44-46)
    from nerf_amd.util_modules import PositionalEncoding
    for lo, hi, key in ((0, 4, '4'), (0, 10, '10'), (0, 16, '16'), (1, 5, '1_5')):
        enc, var = PositionalEncoding(lo, hi).cuda()(mean, cov)
        close(enc, fx['ipe_' + key], 0, 2e-6)
        close(var, fx['ipe_var_' + key], 0, 2e-6)
        assert float(var.min()) >= 0.0
    v = G(fx['view'])
    close(ops.pe_encode(v, 0, 4), fx['pe_4'], 0, 1e-6)
    full = ops.pe_encode(v, 0, 4, append_input=True)
    close(full[:, :24], fx['pe_4'], 0, 1e-6)
    assert torch.equal(full[:, 24:].cpu(), T(fx['view']))


def test_sar_raygen_golden(ops, golden):
    fx = golden('misc')
    ray_d, ray_p = ops.sar_raygen(G(fx['az']), G(fx['el']))
    close(ray_d, fx['ray_d'], 1e-6, 1e-7)
    close(ray_p, fx['ray_p'], 1e-5, 1e-2)


@pytest.mark.parametrize('tag,n_out', [('65_65', 65), ('65_129', 129), ('181_181', 181), ('9_5', 5)])
def test_invert_cdf_bit_exact(ops, golden, tag, n_out):
    """Same cdf and u in, the interval indices and samples must equal the reference's bit for bit."""
    fx = golden('resample')
    bins, w = T(fx['bins_' + tag]), T(fx['w_' + tag])
    cdf = O.pdf_to_cdf(w)
    for rnd in (0, 1):
        u = O.resample_u(bins.shape[0], n_out, T(fx[f'urand_{tag}_{rnd}']) if rnd else None).contiguous()
        ref, ref_idx = O.invert_cdf(bins, cdf, u)
        got, idx = ops.invert_cdf(bins.cuda(), cdf.cuda(), u.cuda())
        assert torch.equal(idx.cpu().long(), ref_idx)
        assert torch.equal(got.cpu(), ref)
        assert torch.equal(got.cpu(), T(fx[f'spdf_{tag}_{rnd}']))


@pytest.mark.parametrize('tag,n_out', [('65_65', 65), ('65_129', 129), ('181_181', 181), ('9_5', 5)])
def test_resample_fused_golden(ops, golden, tag, n_out):
    """The fused kernel (pdf, cdf, u, search, lerp) on the reference's own fixtures."""
    fx = golden('resample')
    bins, w = G(fx['bins_' + tag]), G(fx['w_' + tag])
    for rnd in (0, 1):
        ur = G(fx[f'urand_{tag}_{rnd}']) if rnd else None
        got, idx, cdf = ops.resample(bins, w, n_out, 0.0, ur, blur=False, return_idx=True, return_cdf=True)
        ref_cdf = O.pdf_to_cdf(T(fx['w_' + tag]))
        close(cdf, ref_cdf, 0, 1.2e-7)
        # search is exact on the kernel's own cdf
        u = O.resample_u(bins.shape[0], n_out, T(fx[f'urand_{tag}_{rnd}']) if rnd else None).contiguous()
        _, self_idx = O.invert_cdf(bins.cpu(), cdf.cpu(), u)
        diff = (idx.cpu().long() - self_idx).abs()
        if rnd:
            assert int(diff.max()) == 0
        else:
            # deterministic u comes from an in-kernel linspace (scalar form, as torch evaluates it on
            # a GPU); torch's vectorised CPU linspace differs in the last bit, which only matters on the
            # all-zero-weight row where u_j and cdf_j are both ~ j/n.  Neighbouring interval, same sample.
            assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 0.05
        close(got, fx[f'spdf_{tag}_{rnd}'], 2e-6, 2e-6)


def test_resample_along_rays_golden(ops, golden):
    fx = golden('resample')
    for rnd in (0, 1):
        ur = G(fx[f'rar_urand_{rnd}']) if rnd else None
        nt = ops.resample(G(fx['rar_tin']), G(fx['rar_w']), 17, 0.01, ur, blur=True)
        close(nt, fx[f'rar_t_{rnd}'], 2e-6, 2e-6)
        mean, cov = ops.cast_rays(nt, G(fx['rar_origins']), G(fx['rar_directions']), G(fx['rar_radii']), 'cone')
        close(mean, fx[f'rar_mean_{rnd}'], 1e-5, 1e-5)
        close(cov, fx[f'rar_cov_{rnd}'], 1e-4, 1e-8)


def test_resample_vs_oracle_large(ops):
    """Full sizes.  Everything after the cdf must be bit-exact given the kernel's own cdf; the cdf
    itself is within one fp32 ulp of the oracle's (row sum accumulated in double vs torch's float sum)."""
    g = torch.Generator().manual_seed(7)
    for n, s0, n_out in ((513, 64, 129), (100, 180, 181), (64, 64, 65), (3, 500, 77)):
        t = torch.sort(torch.rand(n, s0 + 1, generator=g) * 4 + 2, dim=-1)[0]
        w = torch.rand(n, s0, generator=g) ** 6
        w[::7] = 0
        ur = torch.empty(n, n_out).uniform_(0, 1 / n_out - O.EPS32, generator=g)
        got, idx, cdf = ops.resample(t.cuda(), w.cuda(), n_out, 0.01, ur.cuda(), blur=True, return_idx=True,
                                     return_cdf=True)
        ref_cdf = O.pdf_to_cdf(O.blur_weights(w, 0.01))
        assert float((cdf.cpu() - ref_cdf).abs().max()) <= 2.4e-7
        u = O.resample_u(n, n_out, ur)
        self_samples, self_idx = O.invert_cdf(t, cdf.cpu(), u)
        assert torch.equal(idx.cpu().long(), self_idx)
        assert torch.equal(got.cpu(), self_samples)
        ref, ref_idx = O.resample_t(t, w, n_out, 0.01, ur)
        assert float((idx.cpu().long() != ref_idx).float().mean()) < 1e-3
        assert float((got.cpu() - ref).abs().max()) < 1e-3                # ill-conditioned where the pdf ~ 0
        assert float((got.cpu() - ref).abs().mean()) < 1e-6
        assert (got[:, 1:] >= got[:, :-1]).all()                          # sortedness
        assert float(got.min()) >= float(t.min()) and float(got.max()) <= float(t.max())


def test_composite_golden(ops, golden):
    fx = golden('composite')
    t, dirs = G(fx['t_vals']), G(fx['dirs'])
    for wb in (0, 1):
        rgb = G(fx['rgb']).requires_grad_(True)
        den = G(fx['density']).requires_grad_(True)
        comp, dist, acc, w, alpha = ops.composite(rgb, den, t, dirs, bool(wb))
        close(comp, fx[f'comp_{wb}'], 1e-5, 1e-6)
        close(dist, fx[f'dist_{wb}'], 1e-5, 1e-6)
        close(acc, fx[f'acc_{wb}'], 1e-5, 1e-6)
        close(w, fx[f'w_{wb}'], 1e-5, 1e-7)
        close(alpha, fx[f'alpha_{wb}'], 1e-5, 1e-7)
        loss = (comp * G(fx[f'gc_{wb}'])).sum() + (acc * G(fx[f'ga_{wb}'])).sum() + (dist * G(fx[f'gd_{wb}'])).sum()
        loss.backward()
        ref_drgb, ref_dden = T(fx[f'drgb_{wb}']), T(fx[f'dden_{wb}'])
        close(rgb.grad, ref_drgb, 1e-4, 1e-6)
        ok = ~torch.isnan(ref_dden)                  # the empty ray is NaN upstream (0 * inf); the kernel gives 0
        assert relerr(den.grad.cpu()[ok], ref_dden[ok]) < 1e-4
        assert torch.isfinite(den.grad).all()
    assert float(acc[0]) == 0.0 and float(dist[0]) == float(t[0, 0])


@pytest.mark.parametrize('n,S', [(257, 64), (130, 128), (65, 180), (9, 300)])
def test_composite_vs_oracle(ops, n, S):
    """Chunked scan (S > 64), fused activation, gradients through comp_rgb/acc/distance."""
    g = torch.Generator().manual_seed(S)
    t = torch.sort(torch.rand(n, S + 1, generator=g) * 4 + 2, dim=-1)[0]
    dirs = torch.randn(n, 3, generator=g)
    raw_rgb = torch.rand(n, S, 3, generator=g)
    raw_den = torch.randn(n, S, 1, generator=g) * 3
    raw_den[0] = 30.0                      # softplus linear branch (threshold 20)
    gc, ga, gd = torch.randn(n, 3, generator=g), torch.randn(n, generator=g), torch.randn(n, generator=g)
    for wb in (False, True):
        rr = raw_rgb.clone().requires_grad_(True)
        rd = raw_den.clone().requires_grad_(True)
        rgb, den = O.activate(rd, rr, -1.0, 0.001)
        ref = O.composite(rgb, den, t, dirs, wb)
        ((ref[0] * gc).sum() + (ref[2] * ga).sum() + (ref[1] * gd).sum()).backward()
        xr = raw_rgb.cuda().requires_grad_(True)
        xd = raw_den.cuda().requires_grad_(True)
        got = ops.composite(xr, xd, t.cuda(), dirs.cuda(), wb, act=(True, 0.001, -1.0))
        ((got[0] * gc.cuda()).sum() + (got[2] * ga.cuda()).sum() + (got[1] * gd.cuda()).sum()).backward()
        for a, b in zip(got, ref):
            assert relerr(a.cpu(), b) < 1e-5
        assert relerr(xr.grad.cpu(), rr.grad) < 1e-4
        assert relerr(xd.grad.cpu(), rd.grad) < 1e-4
        assert float(got[2].max()) <= 1.0 + 1e-6 and float(got[3].min()) >= 0.0


def test_scatter_golden(ops, golden):
    fx = golden('scatter')
    rgb, ang, den = G(fx['rgb']), G(fx['angle']), G(fx['density'][..., 0])
    wl = float(fx['wavelength'])
    pulse, dist, acc, w, alpha, ray_bin, status = ops.scatter(rgb, ang, den, G(fx['power']), G(fx['t_vals']),
                                                              G(fx['dirs']), G(fx['pulse_bins']), wl)
    assert int(status) == 0
    close(acc, fx['acc'], 1e-5, 1e-7)
    close(w, fx['w'], 1e-5, 1e-7)
    close(dist, fx['dist'], 1e-6, 0)
    # The phase is -4 pi d / lambda ~ 4e5 rad evaluated in fp32: one ulp of `distance` (6e-5) turns the
    # return by ~2.4e-2 rad.  So the pulse is compared (a) exactly through the reference's reflectance
    # formula fed with the kernel's own distance, (b) in magnitude against the reference fixture.
    comp_rgb = (w[..., None] * rgb).sum(-2).cpu()
    comp_ang = ((w[..., None] * ang).sum(-2) * 3.14).cpu()
    tail = O.scatter_reflect(comp_rgb, comp_ang, dist.cpu(), T(fx['power']), T(fx['dirs']), wl)
    bins = torch.bucketize(dist.cpu(), T(fx['pulse_bins']))
    assert torch.equal(bins, ray_bin.cpu().long())
    expect = torch.zeros_like(T(fx['pulse']))
    for r in range(tail.shape[0]):                        # ascending ray order: the last writer wins
        expect[bins[r]] = tail[r]
    assert relerr(pulse.cpu(), expect) < 2e-4
    mag = lambda p: torch.linalg.norm(torch.as_tensor(p), dim=-1)
    assert relerr(mag(pulse.cpu()), mag(T(fx['pulse']))) < 1e-5
    assert len(set(ray_bin[3:6].tolist())) == 1           # the duplicate-bin rays share a bin ...
    win = int(torch.nonzero(ray_bin == ray_bin[5]).max())            # ... and the highest index in it wins
    assert win >= 5 and relerr(pulse[ray_bin[5].long()].cpu(), tail[win]) < 2e-4


@pytest.mark.parametrize('wavelength', [300.0, 25.0])
def test_scatter_backward_vs_oracle(ops, golden, wavelength):
    """Hand-written SAR backward (reflectance, phase, bin gather, compositing) against oracle autograd.
    A long wavelength keeps the fp32 phase well conditioned so the comparison is elementwise tight; the
    kernels take the wavelength as a plain parameter, so the same code runs at 3 cm."""
    fx = golden('scatter')
    g = torch.Generator().manual_seed(3)
    gp = torch.randn(T(fx['pulse']).shape, generator=g)
    raw_rgb = torch.rand(T(fx['rgb']).shape, generator=g)
    raw_den = torch.randn(T(fx['density']).shape, generator=g)
    args = [T(fx[k]) for k in ('power', 't_vals', 'dirs', 'pulse_bins')]
    a = [x.clone().requires_grad_(True) for x in (raw_rgb, T(fx['angle']), raw_den)]
    rgb_a, den_a = O.activate(a[2], a[0], -1.0, 1e-4)
    ref = O.scatter(rgb_a, a[1], den_a, *args, wavelength)
    # the loss also reads the returned distance and acc (utils.py:380 returns both with their graph)
    gd, ga = torch.randn(ref[1].shape, generator=g), torch.randn(ref[2].shape, generator=g)
    ((ref[0] * gp).sum() + (ref[1] * gd).sum() + (ref[2] * ga).sum()).backward()
    b = [x.clone().cuda().requires_grad_(True) for x in (raw_rgb, T(fx['angle']), raw_den[..., 0])]
    got = ops.scatter(b[0], b[1], b[2], *[x.cuda() for x in args], wavelength, act=(True, 1e-4, -1.0))
    ((got[0] * gp.cuda()).sum() + (got[1] * gd.cuda()).sum() + (got[2] * ga.cuda()).sum()).backward()
    assert torch.equal(got[5].cpu().long(), ref[5])
    assert relerr(got[0].cpu(), ref[0]) < 1e-4
    assert relerr(b[0].grad.cpu(), a[0].grad) < 5e-4
    assert relerr(b[1].grad.cpu(), a[1].grad) < 5e-4
    assert relerr(b[2].grad.cpu(), a[2].grad[..., 0]) < 5e-4
    # every ray of the shared bin receives gradient, not only the one whose value was written
    assert all(float(b[0].grad[r].abs().sum()) > 0 for r in (3, 4, 5))


def test_scatter_out_of_range_flag(ops, golden):
    fx = golden('scatter')
    bins = G(fx['pulse_bins'])[:8].contiguous()            # distances beyond the last bin -> reference raises
    out = ops.scatter(G(fx['rgb']), G(fx['angle']), G(fx['density'][..., 0]), G(fx['power']), G(fx['t_vals']),
                      G(fx['dirs']), bins, float(fx['wavelength']))
    assert int(out[-1]) == -4


@pytest.mark.parametrize('levels,nsam,nfft', [(2, 2048, 8192), (2, 256, 512), (1, 37, 64), (3, 64, 64), (2, 2, 2)])
def test_sar_pulse_loss_vs_oracle(ops, levels, nsam, nfft):
    """Matched filter, normalisation, loss and gradient in one launch against the oracle's torch.fft chain and its
    autograd, both evaluated in double precision on the same fp32 inputs (the oracle's chain itself is pinned by the
    reference's end-to-end SAR fixtures): profiles, per-level errors, loss and gradient within fp32 rounding of a
    13-stage transform."""
    g = torch.Generator().manual_seed(levels * 1000 + nsam)
    pulse = torch.randn(levels, nsam, 2, generator=g) * 3e4           # the scale of summed ray returns
    mf = torch.complex(torch.randn(1, nfft, generator=g), torch.randn(1, nfft, generator=g))
    target = torch.randn(1, nsam, 2, generator=g)
    cwd = 0.1
    leaf = pulse.double().requires_grad_(True)
    want = torch.stack([O.pulse_postprocess(torch.view_as_complex(leaf[l][None]), mf.to(torch.complex128), nsam)[0]
                        for l in range(levels)])
    want_loss, _ = O.sar_loss(want[:, None], target.double(), cwd)
    want_g, = torch.autograd.grad(want_loss, leaf)
    loss, mse, prof, grad = ops.sar_pulse_loss(pulse.cuda(), mf.cuda(), target.cuda(), cwd, need_profile=True)
    assert relerr(prof.cpu(), want.detach().float()) < 2e-5
    assert abs(float(loss) - float(want_loss.detach())) < 1e-5 * abs(float(want_loss.detach()))
    want_mse = ((want.detach() - target.double()) ** 2).mean(dim=(1, 2))
    assert relerr(mse.cpu(), want_mse.float()) < 1e-5
    assert relerr(grad.cpu(), want_g.float()) < 5e-5
    # forward only: no gradient buffer, the same errors
    loss2, mse2, _, none = ops.sar_pulse_loss(pulse.cuda(), mf.cuda(), target.cuda(), cwd, need_grad=False)
    assert none is None and torch.equal(mse2, mse)


def test_sar_pulse_loss_refuses_other_sizes(ops):
    from nerf_amd import _lib
    z = torch.zeros(1, 8, 2, device='cuda')
    for nsam, nfft in ((8, 12), (8, 16384), (8, 4), (1, 8)):
        assert not ops.sar_pulse_supported(nsam, nfft)
    import ctypes
    m = torch.zeros(1, device='cuda')
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    L = _lib.lib()
    assert L.nerf_sar_pulse_loss(P(z), P(z), P(z), 1, 8, 12, 0.1, None, P(m), None, None) == -3      # NERF_ERR_SIZE
    assert L.nerf_sar_pulse_loss(P(z), P(z), P(z), 1, 8, 4, 0.1, None, P(m), None, None) == -3
    assert L.nerf_sar_pulse_loss(P(z), P(z), P(z), 1, 1, 8, 0.1, None, P(m), None, None) == -1       # NERF_ERR_ARG
    assert L.nerf_sar_pulse_loss(P(z), None, P(z), 1, 8, 8, 0.1, None, P(m), None, None) == -1


def test_empty_inputs(ops):
    z = torch.empty(0, 5, device='cuda')
    assert ops.stratified_t(z[:, :1], z[:, :1], 4).shape == (0, 5)
    assert ops.resample(z, z[:, :4], 5, 0.01).shape == (0, 5)
    with pytest.raises(RuntimeError):
        ops.stratified_t(torch.zeros(3, 1), torch.ones(3, 1), 4)         # CPU tensors are refused


def test_pinhole_rays_match_oracle():
    """On-device ray generation against the NumPy restatement: every field bit-exact, for the whole pixel grid of
    several cameras and for a ragged id subset (including first/last rows, where the radius rule changes)."""
    from nerf_amd import ops
    g = torch.Generator().manual_seed(11)
    C, h, w, focal = 3, 37, 53, 61.25
    rot, _ = torch.linalg.qr(torch.randn(C, 3, 3, generator=g))
    c2w = torch.cat([rot, torch.randn(C, 3, 1, generator=g) * 3], -1)
    c2w = torch.cat([c2w, torch.tensor([0., 0., 0., 1.]).expand(C, 1, 4)], 1).float()
    ref = O.pinhole_rays(c2w.numpy(), h, w, focal, 2.0, 6.0)
    got = ops.pinhole_rays(c2w.cuda(), h, w, focal, 2.0, 6.0)
    for name, a, b in zip(got._fields, got, ref):
        assert torch.equal(a.cpu(), b), name
    ids = torch.cat([torch.randint(0, C * h * w, (1001,), generator=g), torch.tensor([0, w - 1, (h - 1) * w, C * h * w - 1])])
    sub = ops.pinhole_rays(c2w.cuda(), h, w, focal, 2.0, 6.0, ids=ids.cuda())
    for name, a, b in zip(sub._fields, sub, ref):
        assert torch.equal(a.cpu(), b[ids]), name
    assert ops.pinhole_rays(c2w.cuda(), h, w, focal, 2.0, 6.0, ids=ids[:0].cuda()).origins.shape == (0, 3)


@pytest.mark.parametrize('tag', ['pinhole', 'pinhole_tele', 'llff'])
def test_ray_generation_matches_reference(golden, tag):
    """nerf_pinhole_rays / nerf_ndc_rays against the reference's own rays (`NeRFDataset.generate_rays`,
    dataloader.py:103-138, and `LLFF.generate_rays`, dataloader.py:437-460, run on bare instances and flattened by
    `flatten_to_pytorch`): pinhole fields bit for bit, NDC fields to fp32 rounding of the projection; then a
    shuffled id subset, which must equal the same rows."""
    from nerf_amd import ops
    g = golden('rays')
    h, w, focal = g[tag + '_hwf']
    near, far = g[tag + '_nearfar']
    c2w = torch.tensor(g[tag + '_c2w']).cuda()
    ndc = tag == 'llff'
    got = ops.pinhole_rays(c2w, int(h), int(w), float(focal), float(near), float(far), ndc=ndc)
    for name, a in zip(got._fields, got):
        ref = g[f'{tag}_{name}']
        if ndc:
            np.testing.assert_allclose(a.cpu().numpy(), ref, rtol=3e-5, atol=2e-6, err_msg=name)
        else:
            assert torch.equal(a.cpu(), T(ref)), (tag, name)
    ids = torch.randperm(got.origins.shape[0], generator=torch.Generator().manual_seed(1))[:77].cuda()
    sub = ops.pinhole_rays(c2w, int(h), int(w), float(focal), float(near), float(far), ids=ids, ndc=ndc)
    for a, b in zip(sub, got):
        assert torch.equal(a, b[ids])


def test_convert_to_ndc_matches_golden():
    """utils.convert_to_ndc against the reference's output on float32 rays (bit-exact: same fp32 expression order)."""
    import os
    from nerf_amd import utils as U
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ndc.npz'))
    o = torch.tensor(g['origins']).cuda()
    d = torch.tensor(g['directions']).cuda()
    for near, ko, kd in ((1., 'o_ndc', 'd_ndc'), (0.5, 'o_ndc_near05', 'd_ndc_near05')):
        oo, dd = U.convert_to_ndc(o, d, float(g['focal']), int(g['w']), int(g['h']), near)
        assert oo.shape == o.shape
        np.testing.assert_allclose(oo.cpu().numpy(), g[ko], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(dd.cpu().numpy(), g[kd], rtol=2e-6, atol=1e-7)
    from nerf_amd.model import to8b
    assert np.array_equal(to8b(g['img']), g['img8'])


def test_ndc_rays_match_oracle():
    """nerf_ndc_rays against the NumPy restatement of the forward-facing loader (parity unpinned beyond
    convert_to_ndc): all pixels, then a shuffled id subset, then ids outside the stack."""
    from nerf_amd import ops
    rng = np.random.default_rng(5)
    C, h, w, focal = 3, 9, 11, 13.7
    c2w = np.zeros((C, 3, 4), np.float32)
    for c in range(C):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        c2w[c, :, :3] = q
        c2w[c, :, 3] = rng.standard_normal(3) * 0.3 + [0, 0, 1.5]
    want = O.ndc_rays(c2w, h, w, focal, 0., 1.)
    got = ops.ndc_rays(torch.tensor(c2w).cuda(), h, w, focal, 0., 1.)
    for name, a, b in zip(got._fields, got, want):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=3e-5, atol=2e-6, err_msg=name)
    ids = torch.tensor(rng.permutation(C * h * w)[:100]).cuda()
    sub = ops.ndc_rays(torch.tensor(c2w).cuda(), h, w, focal, 0., 1., ids)
    for a, b in zip(sub, got):
        assert torch.equal(a, b[ids])
    bad = ops.ndc_rays(torch.tensor(c2w).cuda(), h, w, focal, 0., 1., torch.tensor([C * h * w, -1, 5]).cuda())
    assert torch.isnan(bad.origins[:2]).all() and torch.equal(bad.origins[2], got.origins[5])
    bad = ops.pinhole_rays(torch.tensor(c2w).cuda(), h, w, focal, 2., 6., torch.tensor([C * h * w, 7]).cuda())
    assert torch.isnan(bad.radii[0]).all() and torch.isfinite(bad.radii[1]).all()


def test_ray_batches_epoch():
    """One shuffled epoch visits every pixel once; rays and pixels of a batch belong to the same ids; two ranks
    split each batch without overlap."""
    from nerf_amd import ops
    from nerf_amd.rays import RayBatches
    C, h, w = 2, 6, 7
    c2w = torch.eye(4)[None, :3].repeat(C, 1, 1).cuda()
    c2w[1, 0, 3] = 0.5
    imgs = torch.arange(C * h * w * 3, dtype=torch.float32).reshape(C, h, w, 3).cuda()
    full = ops.pinhole_rays(c2w, h, w, 20., 2., 6.)
    seen = []
    batches = RayBatches(c2w, h, w, 20., 2., 6., images=imgs, batch_size=32, seed=3)
    assert len(batches) == 3
    for rays, pix in batches:
        ids = (pix[:, 0] / 3).long()
        seen.append(ids)
        assert torch.equal(rays.directions, full.directions[ids])
        assert torch.equal(rays.origins, full.origins[ids])
    seen = torch.cat(seen)
    assert seen.numel() == C * h * w and torch.equal(seen.sort().values, torch.arange(C * h * w, device='cuda'))
    halves = [list(RayBatches(c2w, h, w, 20., 2., 6., images=imgs, batch_size=32, seed=3, rank=r, world_size=2))
              for r in range(2)]
    both = torch.cat([p[:, 0] for r in range(2) for _, p in halves[r]])
    assert both.numel() == C * h * w and both.unique().numel() == C * h * w
    ordered = list(RayBatches(c2w, h, w, 20., 2., 6., batch_size=50, shuffle=False))
    assert torch.equal(torch.cat([r.directions for r in ordered]), full.directions)


def test_c_abi_rejects_bad_arguments():
    """Entry points called straight through ctypes with arguments a careless binding could pass: null pointers,
    negative or inconsistent sizes, unsupported geometry.  Each returns a negative code and launches nothing."""
    import ctypes
    from nerf_amd import _lib, mlp_bf16  # noqa: F401  (mlp_bf16 registers its bindings)
    L = _lib.lib()
    buf = torch.zeros(4096, device='cuda')
    p = ctypes.c_void_p(buf.data_ptr())
    null = ctypes.c_void_p(None)
    ERR_ARG, ERR_SIZE = -1, -3
    assert L.nerf_stratified_t(null, p, null, 8, 4, 0, p, None) == ERR_ARG
    assert L.nerf_stratified_t(p, p, null, -1, 4, 0, p, None) == ERR_ARG
    assert L.nerf_stratified_t(p, p, null, 8, 0, 0, p, None) == ERR_ARG           # no samples
    assert L.nerf_pe_encode(p, 8, 4, 4, 0, p, None) == ERR_ARG                  # max_deg must exceed min_deg
    assert L.nerf_pe_encode(null, 8, 0, 4, 0, p, None) == ERR_ARG
    assert L.nerf_pinhole_rays(p, 1, 2, 8, 10.0, 2.0, 6.0, null, 4, p, p, p, p, p, p, p, None) == ERR_ARG     # h < 3
    assert L.nerf_ndc_rays(p, 1, 8, 2, 10.0, 0.0, 1.0, null, 4, p, p, p, p, p, p, p, None) == ERR_ARG         # w < 3
    assert L.nerf_pinhole_rays(p, 1, 8, 8, 0.0, 2.0, 6.0, null, 4, p, p, p, p, p, p, p, None) == ERR_ARG      # focal
    assert L.nerf_convert_to_ndc(p, null, 4, 10.0, 8, 8, 1.0, p, p, None) == ERR_ARG
    assert L.nerf_sar_raygen(p, null, 4, p, p, None) == ERR_ARG
    # zero-sized calls are valid no-ops whatever the pointers
    assert L.nerf_pe_encode(null, 0, 0, 4, 0, null, None) == 0
    assert L.nerf_convert_to_ndc(null, null, 0, 10.0, 8, 8, 1.0, null, null, None) == 0
    nw, tn, tk, gt = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    out = [ctypes.byref(v) for v in (nw, tn, tk, gt)]
    assert L.nerf_mlp_wgrad_config(256, 256, *out) == 0
    assert (nw.value, tn.value, tk.value, gt.value) == (4, 4, 4, 4)
    assert L.nerf_mlp_wgrad_config(31, 256, *out) < 0                             # not a multiple of 32
    assert L.nerf_mlp_wgrad_config(4096, 4096, *out) == ERR_SIZE
    assert float(buf.abs().sum()) == 0.0                                        # nothing was written
    for code in (-1, -2, -3, -4):
        assert len(L.nerf_error_string(code)) > 0


def test_integration_snippet_runs(ops):
    """The hand-written ctypes binding quoted in the integration guide (examples/composite_ctypes.py) calls
    nerf_composite_fwd with the prototype's 14 arguments and returns what `ops.composite` returns; the document
    quotes the file verbatim."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, 'examples', 'composite_ctypes.py')
    text = open(path).read()
    body = text[text.index('# --- begin snippet\n') + len('# --- begin snippet\n'):text.index('# --- end snippet')]
    assert body in open(os.path.join(root, 'INTEGRATION.md')).read()
    spec = importlib.util.spec_from_file_location('composite_ctypes', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = torch.Generator().manual_seed(12)
    n, S = 33, 40
    rgb = torch.rand(n, S, 3, generator=g).cuda()
    den = torch.randn(n, S, generator=g).cuda()
    t = torch.sort(torch.rand(n, S + 1, generator=g) * 4 + 2, dim=-1)[0].cuda()
    dirs = torch.randn(n, 3, generator=g).cuda()
    for wb in (False, True):
        got = mod.volumetric_rendering_fwd(rgb, den, t, dirs, wb, 0.001, -1.0)
        want = ops.composite(rgb, den, t, dirs, wb, act=(True, 0.001, -1.0))
        for a, b in zip(got, want):
            assert torch.equal(a, b)
