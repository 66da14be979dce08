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


class GradBucket:
    """One persistent, contiguous buffer (the parameters' dtype, fp32 for the render models) that holds every gradient: each parameter's `.grad` is a view into
    it, so the weight-gradient kernels accumulate straight into the bucket and the step needs exactly one
    collective on it (or two on its two parts, when the first part is reduced while the second is still being
    computed), with no gather before and no scatter after.  Views start on 256-byte boundaries (the
    padding stays zero and rides along in the collective)."""

    ALIGN = 64          # floats

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += -(-p.numel() // self.ALIGN) * self.ALIGN
        dev = self.params[0].device if self.params else torch.device('cpu')
        dtype = self.params[0].dtype if self.params else torch.float32
        self.flat = torch.zeros(n, dtype=dtype, device=dev)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, self.params)]
        self.attach()

    def attach(self):
        """(Re)install the views as the parameters' `.grad` (an optimizer's `zero_grad(set_to_none=True)` or a
        backward pass into a parameter without a gradient replaces them)."""
        for p, v in zip(self.params, self.views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v

    def zero(self):
        self.attach()
        self.flat.zero_()

    def all_reduce(self, world_size, async_op=False, lo=0, hi=None):
        """SUM over ranks, in place (RCCL on GPUs, gloo on CPU), of the whole bucket or of its floats lo .. hi; returns
        the work handle when `async_op`."""
        if world_size <= 1:
            return None
        return _all_reduce_sum(self.flat if (lo == 0 and hi is None) else self.flat[lo:hi], async_op=async_op)

    def prefix_end(self, named_params, prefix):
        """End (in floats) of the leading run of parameters whose names start with `prefix` (0 if the first does not):
        the part of the bucket that is complete once those parameters' gradients are."""
        names = {id(p): k for k, p in named_params}
        end = 0
        for p, v in zip(self.params, self.views):
            if not names.get(id(p), '').startswith(prefix):
                break
            end = v.storage_offset() + -(-p.numel() // self.ALIGN) * self.ALIGN
        return min(end, self.flat.numel())


def grad_bucket(model_or_params):
    """The gradient bucket of a model (cached on it) or of a parameter list."""
    if isinstance(model_or_params, torch.nn.Module):
        params = [p for p in model_or_params.parameters() if p.requires_grad]
        b = getattr(model_or_params, '_grad_bucket', None)
        if (b is None or len(b.params) != len(params) or any(x is not y for x, y in zip(b.params, params))
                or (params and b.flat.device != params[0].device)):
            b = GradBucket(params)
            model_or_params._grad_bucket = b
        return b
    return GradBucket(list(model_or_params))


def allreduce_grads(params, world_size):
    """SUM-all-reduce of the gradients of `params` as one flat bucket.  Gradients that already are views of a
    `GradBucket` are reduced in place; otherwise they are moved into a fresh bucket first."""
    if world_size <= 1:
        return
    GradBucket([p for p in params if p.grad is not None or p.requires_grad]).all_reduce(world_size)


def _all_reduce_sum(t, async_op=False):
    """In-place SUM all-reduce.  RCCL reduces device buffers directly; under gloo (rehearsals of the multi-rank
    path with both ranks on one GPU) a device tensor is staged through host memory, because gloo's own device path
    blocks there: `tools/diag_gloo_device.py` shows `dist.all_reduce` on a CUDA tensor completing on an idle device
    and never returning (both ranks, already for 1 KiB) as soon as kernels are still queued on the tensor's stream
    when the collective is called -- the hand-over from the compute stream to gloo's copy stream does not resolve
    on this ROCm build; size does not matter.  `.cpu()` waits for the stream first, so the staged form is safe."""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend() == 'gloo':
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        t.copy_(host)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


def shard_range(n, rank, world_size):
    """Contiguous, near-equal ray shard of rank `rank`."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def nerf_train_step(model, rays, pixels, optimizer, chunk=8192, world_size=1, global_mask_sum=None, scheduler=None,
                    fused=True):
    """fwd + bwd (+ all-reduce) + optimizer step on this rank's rays.

    The loss is sum(mask * err^2) / sum(mask) over the GLOBAL batch (util_modules.py:65), so each
    chunk contributes its masked squared-error sum divided by the global mask sum; gradients then
    add up across chunks and ranks.

    When the batch is one chunk and the model runs the bf16 chain, the step is the fixed launch sequence of
    `nerf_amd.stepper` (no autograd, persistent buffers, one backward chain and one weight-gradient pass for all
    levels); otherwise (`fused=False`, several chunks, the fp32 path, density noise) it goes through autograd
    chunk by chunk.  Both accumulate into the same gradient bucket."""
    bucket = grad_bucket(model)
    n = rays.origins.shape[0]
    if global_mask_sum is None:             # the denominator of the reference loss is the mask sum of the WHOLE batch
        global_mask_sum = rays.lossmult.sum()
        if world_size > 1:                  # shards may be unequal (short last batch): add them up, do not scale
            global_mask_sum = global_mask_sum.clone()
            _all_reduce_sum(global_mask_sum)
    bucket.zero()
    from . import stepper
    if fused and chunk >= n and n > 0 and stepper.supported(model):
        # Several ranks: the trunk layers' gradients (the leading nine tenths of the bucket) are reduced as soon as
        # their weight-gradient launches are queued, under the remaining launches; the collective runs on the
        # backend's own stream behind what the compute stream holds at that point (the mechanism bucketed data
        # parallelism overlaps with), and the rest of the bucket follows at the end of the pass.
        early = {}

        def reduce_trunk():
            hi = bucket.prefix_end(model.named_parameters(), 'density_net')
            if world_size > 1 and hi > 0:
                early['hi'], early['work'] = hi, bucket.all_reduce(world_size, async_op=True, lo=0, hi=hi)
        total, _ = stepper.stepper_for(model, n).step(rays, pixels, global_mask_sum,
                                                      after_trunk=reduce_trunk if world_size > 1 else None)
        return _finish_step(model, bucket, optimizer, scheduler, world_size, total, early)
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
    return _finish_step(model, bucket, optimizer, scheduler, world_size, total)


def _finish_step(model, bucket, optimizer, scheduler, world_size, total, early=None):
    bucket.attach()
    lo = early.get('hi', 0) if early else 0           # floats 0 .. lo are already being reduced (nerf_train_step)
    work = bucket.all_reduce(world_size, async_op=True, lo=lo, hi=None if lo == 0 else bucket.flat.numel())
    for w in ((early or {}).get('work'), work):
        if w is not None:
            w.wait()            # orders the collective ahead of the optimizer kernels on the current stream
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


def sar_train_step(model, batch, optimizer, rank=0, world_size=1, scheduler=None, fused=True, **draws):
    """One SAR pulse, ray chunks dealt round-robin to the ranks (model.py:326-375 loops over them on one device).

    The reference ADDS the chunks' range profiles before the matched filter (model.py:371), and within a chunk
    the last ray written to a range bin wins; dealing whole chunks keeps both rules intact.  Each rank renders
    its chunks, the per-level profiles are SUM-all-reduced (nsam x 2 floats per level) ahead of the FFT and the
    per-pulse std, every rank evaluates the same loss, and the weight gradients -- partial sums over each
    rank's chunks -- are SUM-all-reduced in one flat bucket."""
    ray_o, pan, tilt, mfilt, radii, near, far, mpp, pulse_data = batch
    bucket = grad_bucket(model)
    bucket.zero()
    from . import stepper
    if fused and stepper.sar_supported(model):
        # the pulse as a fixed launch sequence: one backward chain and one weight-gradient pass for all passes
        # (several ranks: each over its own chunks, the range profiles all-reduced inside the step)
        loss = stepper.sar_stepper_for(model, rank, world_size).step(ray_o, mfilt, radii, near, far, pulse_data.shape[1],
                                                                     mpp, pulse_data, **draws)
        bucket.attach()
        bucket.all_reduce(world_size)
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
        model._weights_changed()
        return loss
    pulses = model(ray_o, pan, tilt, mfilt, radii, near, far, pulse_data.shape[1], mpp, rank=rank,
                   world_size=world_size, **draws)[0]
    loss, _ = model.loss_function.level_losses(pulses, pulse_data.to(pulses.device))      # (the PSNRs go to the host)
    if loss.requires_grad:              # a rank that was dealt no chunk has nothing to differentiate
        loss.backward()
    bucket.attach()
    bucket.all_reduce(world_size)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    model._weights_changed()
    return loss.detach()


class GraphedSarStep:
    """`sar_train_step` captured once as a HIP graph and replayed.

    At the reference's sizes (params.yaml:67-93: 2048 rays x 180 samples per chunk, a 156-wide network) one pulse is
    about a hundred short launches -- samplers, encoders, chain kernels, scatter, FFTs, loss, the backward of all of
    them, AdamW, weight repack -- and a step issued launch by launch is bound by the host; replaying the captured
    graph leaves only the kernels.  Needs one rank, `config.check_range = False` (the range check reads a status
    word back, which a capture cannot do) and an optimizer built with `capturable=True`.  The learning rate must be
    a device tensor for a replay to see a new value (a Python number is baked into the captured kernels' arguments):
    a float `lr` of a parameter group is replaced by a one-element device tensor here, before the warm-up steps, and
    `step(lr=...)` / `set_lr` write into it (the reference steps MipLRDecay after every optimizer step,
    model.py:389-393).  The beam offsets are
    drawn on the device (the reference draws them on the host, model.py:319-320: same distribution, another
    stream of numbers); every replay draws afresh.  The `warmup` eager runs that precede the capture are real
    optimizer steps."""

    def __init__(self, model, batch, optimizer, warmup=3):
        if getattr(model.config, 'check_range', True):
            raise ValueError('GraphedSarStep: set config.check_range = False (the check synchronises the stream)')
        if not all(g.get('capturable', False) for g in optimizer.param_groups):
            raise ValueError('GraphedSarStep: build the optimizer with capturable=True')
        self.model, self.optimizer = model, optimizer
        dev = model.device
        for g in optimizer.param_groups:        # see the class comment: only a tensor lr reaches a replay
            if not torch.is_tensor(g['lr']):
                g['lr'] = torch.tensor(float(g['lr']), dtype=torch.float32, device=dev)
            elif g['lr'].device != dev:
                raise ValueError('GraphedSarStep: the optimizer\'s tensor lr must live on the model\'s device')
        self.batch = tuple(x.to(dev).clone() if torch.is_tensor(x) else x for x in batch)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):           # one-time work (LDS attributes, FFT plans, optimizer state) happens here
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._body()

    def _body(self):
        from .model import SAR_AZ_SIGMA, SAR_EL_SIGMA
        m = self.model
        az_el = [(torch.randn(m.rays_per_chunk, device=m.device) * SAR_AZ_SIGMA,
                  torch.randn(m.rays_per_chunk, device=m.device) * SAR_EL_SIGMA) for _ in range(m.num_ray_iters)]
        return sar_train_step(m, self.batch, self.optimizer, az_el=az_el)

    def set_lr(self, lr):
        """Write `lr` (a number, or one per parameter group) into the optimizer's tensor learning rates."""
        groups = self.optimizer.param_groups
        lrs = list(lr) if isinstance(lr, (list, tuple)) else [lr] * len(groups)
        for g, v in zip(groups, lrs):
            if not torch.is_tensor(g['lr']):
                raise RuntimeError('GraphedSarStep: a parameter group\'s lr was replaced by a Python number after the '
                                   'capture; a replay would not see it')
            g['lr'].fill_(float(v))

    def step(self, batch=None, lr=None, scheduler=None):
        """Replay on `batch` (copied into the captured tensors; None = the same pulse again) -> the loss tensor of
        the capture (overwritten by every replay).  `lr` is written into the tensor learning rates ahead of the
        replay; `scheduler` (e.g. MipLRDecay on this optimizer) is stepped after it, and the rate it computes is
        written back into the tensors (schedulers assign Python numbers to `param_groups[i]['lr']`)."""
        if batch is not None:
            for dst, src in zip(self.batch, batch):
                if torch.is_tensor(dst):
                    dst.copy_(src)
        if lr is not None:
            self.set_lr(lr)
        self.graph.replay()
        if scheduler is not None:
            tensors = [g['lr'] for g in self.optimizer.param_groups]
            scheduler.step()
            for g, t in zip(self.optimizer.param_groups, tensors):
                if g['lr'] is not t:            # the scheduler put a number there: keep the tensor, take its value
                    t.fill_(float(g['lr']))
                    g['lr'] = t
        self.model._weights_changed()           # the packed streams outside the graph are one step behind
        return self.loss
