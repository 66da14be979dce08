"""Ray-sharded data parallelism on CPU (gloo, world_size 2): the host logic that the GPU path uses --
contiguous shards, per-shard loss normalised by the GLOBAL mask sum, one flat-bucket SUM all-reduce --
must give every rank the gradients of the unsharded batch.  Compute is the CPU oracle here (the HIP
kernels need a GPU); only the sharding/collective code of nerf_amd.train is under test."""
import os
import socket
from types import SimpleNamespace

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nerf_oracle as O


def _cfg():
    return SimpleNamespace(use_viewdirs=True, randomized=False, ray_shape='cone', white_bkgd=False, num_levels=2,
                           num_samples=16, min_deg=0, max_deg=4, viewdirs_min_deg=0, viewdirs_max_deg=4,
                           density_noise=0., rgb_padding=0.001, resample_padding=0.01, density_bias=-1., hidden=32,
                           coarse_weight_decay=0.1)


def _batch(n):
    g = torch.Generator().manual_seed(7)
    o = 0.1 * torch.randn(n, 3, generator=g) + torch.tensor([0., 0., 4.])
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    lossmult = (torch.rand(n, 1, generator=g) > 0.3).float()          # ragged mask: shards differ in weight
    rays = O.Rays(o, d, d.clone(), torch.full((n, 1), 0.005), lossmult, torch.full((n, 1), 2.), torch.full((n, 1), 6.))
    return rays, torch.rand(n, 3, generator=g)


def _shard_loss(p, rays, tgt, cfg, lo, hi, mask_sum):
    r = O.Rays(*(getattr(rays, f)[lo:hi] for f in ('origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'near', 'far')))
    out = O.nerf_render(p, r, cfg, [16, 16], None, None)
    sq = (r.lossmult * (out['comp_rgb_stack'] - tgt[lo:hi][None]) ** 2).sum(dim=(1, 2)) / mask_sum
    return cfg.coarse_weight_decay * sq[:-1].sum() + sq[-1]


def _worker(rank, world, port, n, q):
    from nerf_amd.train import GradBucket, _all_reduce_sum, shard_range
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = _cfg()
    rays, tgt = _batch(n)
    params = [torch.nn.Parameter(v) for v in O.init_params(cfg, seed=3).values()]
    p = dict(zip(O.init_params(cfg, seed=3).keys(), params))
    lo, hi = shard_range(n, rank, world)
    # the step's own recipe: gradients are views of one flat bucket; the loss denominator is the mask sum of the
    # whole batch, obtained by adding up the (unequal) shards' sums
    bucket = GradBucket(params)
    bucket.zero()
    mask_sum = rays.lossmult[lo:hi].sum().clone()
    _all_reduce_sum(mask_sum)
    _shard_loss(p, rays, tgt, cfg, lo, hi, mask_sum).backward()
    bucket.attach()
    assert all(x.grad.data_ptr() == v.data_ptr() for x, v in zip(params, bucket.views))
    # as the fused step reduces it: the trunk layers' leading part of the bucket first (started while the rest of the
    # weight-gradient pass is still running), then the remainder; together the whole bucket, once
    split = bucket.prefix_end(list(p.items()), 'density_net')
    assert 0 < split < bucket.flat.numel() and split == sum(-(-v.numel() // 64) * 64 for k, v in p.items() if k.startswith('density_net'))
    works = [bucket.all_reduce(world, async_op=True, lo=0, hi=split),
             bucket.all_reduce(world, async_op=True, lo=split, hi=bucket.flat.numel())]
    for w in works:
        w.wait()
    q.put((rank, lo, hi, [x.grad.numpy().copy() for x in params]))     # by value: the worker may exit first
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from nerf_amd.train import shard_range
    for n in (0, 1, 7, 64, 65537):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gradients_equal_unsharded():
    n, world = 37, 2                                                   # odd: unequal shards
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = []
    try:
        for _ in range(world):
            got.append(q.get(timeout=90))
    finally:
        for pr in procs:
            pr.join(30)
            if pr.is_alive():
                pr.kill()
    assert all(pr.exitcode == 0 for pr in procs)
    got.sort(key=lambda t: t[0])
    assert got[0][1:3] == (0, 19) and got[1][1:3] == (19, 37)

    cfg = _cfg()
    rays, tgt = _batch(n)
    p = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=3).items()}
    out = O.nerf_render(p, rays, cfg, [16, 16], None, None)
    loss, _ = O.nerf_loss(out['comp_rgb_stack'], tgt, rays.lossmult, cfg.coarse_weight_decay)
    loss.backward()
    for g0, g1, ref in zip(got[0][3], got[1][3], p.values()):
        g0, g1 = torch.from_numpy(g0), torch.from_numpy(g1)
        assert torch.equal(g0, g1)                                      # every rank holds the same reduced bucket
        assert float((g0 - ref.grad).abs().max()) <= 1e-5 * float(ref.grad.abs().max()) + 1e-9


def _shared_loss_worker(rank, world, port, q):
    from nerf_amd.train import allreduce_grads, allreduce_sum_shared_loss
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    theta = torch.nn.Parameter(torch.linspace(-1, 1, 12, dtype=torch.float64).reshape(3, 4))
    parts = _parts(theta)
    mine = sum((parts[c] for c in range(len(parts)) if c % world == rank), torch.zeros(5, dtype=torch.float64))
    y = allreduce_sum_shared_loss(mine, world)
    loss = _shared_loss(y)
    if loss.requires_grad:
        loss.backward()
    else:
        theta.grad = torch.zeros_like(theta)
    allreduce_grads([theta], world)
    q.put((rank, y.detach().numpy().copy(), theta.grad.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def _parts(theta):
    """Three 'chunks': different differentiable maps of the shared parameters to a 5-vector."""
    w = torch.cos(torch.arange(20, dtype=torch.float64).reshape(4, 5))
    return [torch.tanh(theta[c] @ w) * (c + 1) for c in range(3)]


def _shared_loss(y):
    return ((y / y.std()) ** 2 * torch.arange(1, 6)).sum()        # a normalisation over the whole sum, like the pulse std


def test_chunk_dealing_with_shared_loss_matches_one_process():
    """SAR sharding: chunks dealt round-robin, their sum all-reduced ahead of a loss every rank evaluates; the
    SUM-all-reduced parameter gradients equal those of one process looping over all chunks (3 chunks on 2 ranks,
    so the shares are unequal)."""
    world = 2
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_shared_loss_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = []
    try:
        for _ in range(world):
            got.append(q.get(timeout=90))
    finally:
        for pr in procs:
            pr.join(30)
            if pr.is_alive():
                pr.kill()
    assert all(pr.exitcode == 0 for pr in procs)
    theta = torch.nn.Parameter(torch.linspace(-1, 1, 12, dtype=torch.float64).reshape(3, 4))
    y = sum(_parts(theta))
    _shared_loss(y).backward()
    for _, yy, g in got:
        assert float((torch.from_numpy(yy) - y.detach()).abs().max()) < 1e-12
        assert float((torch.from_numpy(g) - theta.grad).abs().max()) <= 1e-9 * float(theta.grad.abs().max())
