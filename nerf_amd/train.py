"""Training-step driver for the ray-sharded render path: chunked forward/backward over a batch of
rays with gradient accumulation, one flat-bucket all-reduce of the weight gradients across ranks,
one optimizer step.  (The reference's `training_step`, model.py:185-191, processes the whole batch
at once on one device.)"""
import torch

from .utils import Rays, namedtuple_map


def synthetic_rays(n, device, seed=0, normalize=True):
    """Synthetic Blender-like rays: origins near (0,0,4), unit directions, near=2, far=6."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    o = 0.1 * torch.randn(n, 3, generator=g) + torch.tensor([0., 0., 4.])
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    if not normalize:
        d = d * (1.0 + 0.2 * torch.rand(n, 1, generator=g))
    v = d / d.norm(dim=-1, keepdim=True)
    rays = Rays(o, d, v, torch.full((n, 1), 0.005), torch.ones(n, 1), torch.full((n, 1), 2.), torch.full((n, 1), 6.))
    pixels = torch.rand(n, 3, generator=g)
    return namedtuple_map(lambda t: t.to(device), rays), pixels.to(device)


def flat_grads(params):
    """One contiguous fp32 bucket holding every gradient (a single collective per step)."""
    return torch.cat([p.grad.reshape(-1) for p in params])


def scatter_flat_grads(params, flat):
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p))
        off += n


def allreduce_grads(params, world_size):
    """SUM-all-reduce of the flattened gradient bucket (RCCL on GPUs, gloo on CPU)."""
    if world_size <= 1:
        return
    flat = flat_grads(params)
    _all_reduce_sum(flat)
    scatter_flat_grads(params, flat)


def _all_reduce_sum(t):
    """In-place SUM all-reduce.  RCCL reduces device buffers directly; under gloo (CPU rehearsals of the multi-rank
    path) a device tensor is staged through host memory: gloo's own device path was seen to hang on a 2.4 MB
    bucket with two ranks on one GPU."""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend() == 'gloo':
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def shard_range(n, rank, world_size):
    """Contiguous, near-equal ray shard of rank `rank`."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def nerf_train_step(model, rays, pixels, optimizer, chunk=8192, world_size=1, global_mask_sum=None, scheduler=None):
    """fwd + bwd (+ all-reduce) + optimizer step on this rank's rays.

    The loss is sum(mask * err^2) / sum(mask) over the GLOBAL batch (util_modules.py:65), so each
    chunk contributes its masked squared-error sum divided by the global mask sum; gradients then
    add up across chunks and ranks."""
    params = [p for p in model.parameters() if p.requires_grad]
    n = rays.origins.shape[0]
    if global_mask_sum is None:
        global_mask_sum = rays.lossmult.sum() * world_size
    optimizer.zero_grad(set_to_none=False)
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    total = torch.zeros((), device=rays.origins.device)
    cwd = model.loss_function.coarse_weight_decay
    import contextlib
    in_place = contextlib.nullcontext()
    if getattr(model.config, 'mlp_precision', 'fp32') == 'bf16':
        from .mlp_bf16 import accumulate_into_grad
        in_place = accumulate_into_grad()           # the .grad buffers exist (above) and are read from below
    with in_place:
        for i in range(0, n, chunk):
            r = namedtuple_map(lambda t: t[i:i + chunk], rays)
            rgb = model(r)[0]
            sq = (r.lossmult * (rgb - pixels[i:i + chunk][None]) ** 2).sum(dim=(1, 2)) / global_mask_sum
            loss = cwd * sq[:-1].sum() + sq[-1]
            loss.backward()
            total += loss.detach()
    allreduce_grads(params, world_size)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    model._weights_changed()
    return total


class _AllReduceSumSharedLoss(torch.autograd.Function):
    """y = sum over ranks of x, for a y that every rank then feeds to the SAME loss: each rank's gradient with
    respect to its own x is the loss gradient with respect to y, unchanged (no collective in backward); the
    parameter gradients of the ranks are partial sums and are SUM-all-reduced afterwards."""

    @staticmethod
    def forward(ctx, x):
        y = x.detach().clone()
        _all_reduce_sum(y)
        return y

    @staticmethod
    def backward(ctx, g):
        return g


def allreduce_sum_shared_loss(x, world_size):
    return x if world_size <= 1 else _AllReduceSumSharedLoss.apply(x)


def sar_train_step(model, batch, optimizer, rank=0, world_size=1, scheduler=None, **draws):
    """One SAR pulse, ray chunks dealt round-robin to the ranks (model.py:326-375 loops over them on one device).

    The reference ADDS the chunks' range profiles before the matched filter (model.py:371), and within a chunk
    the last ray written to a range bin wins; dealing whole chunks keeps both rules intact.  Each rank renders
    its chunks, the per-level profiles are SUM-all-reduced (nsam x 2 floats per level) ahead of the FFT and the
    per-pulse std, every rank evaluates the same loss, and the weight gradients -- partial sums over each
    rank's chunks -- are SUM-all-reduced in one flat bucket."""
    ray_o, pan, tilt, mfilt, radii, near, far, mpp, pulse_data = batch
    params = [p for p in model.parameters() if p.requires_grad]
    optimizer.zero_grad(set_to_none=False)
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    pulses = model(ray_o, pan, tilt, mfilt, radii, near, far, pulse_data.shape[1], mpp, rank=rank,
                   world_size=world_size, **draws)[0]
    loss, _ = model.loss_function(pulses, pulse_data.to(pulses.device))
    if loss.requires_grad:              # a rank that was dealt no chunk has nothing to differentiate
        loss.backward()
    allreduce_grads(params, world_size)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    model._weights_changed()
    return loss.detach()
