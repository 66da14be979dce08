#!/usr/bin/env python
"""Benchmark of the render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4] [--wgrad-operands bf16|mx]
                    [--scaling weak|strong] [--rays R]

With --gpus N > 1 and no launcher in the environment, bench.py starts the N ranks itself (one process per GPU
through torch.distributed.run, RCCL backend) before it touches a GPU, relays rank 0's JSON line and exits
with the job's status; under an external launcher (WORLD_SIZE set) it is one of the ranks.

Workload ("step"): one full mip-NeRF training step on a batch of synthetic rays already resident in
HBM -- coarse + fine render at 64+128 samples per ray (stratified sampling, cast + integrated positional
encoding, 8x256 MLP with the bf16 MFMA chain, hierarchical resampling, alpha compositing), the masked-MSE
loss over both levels, the hand-written backward of compositing and MLP, the gradient all-reduce across
ranks (RCCL, N > 1), fused AdamW and the weight repack.  65,536 rays per step per GPU (weak scaling: the
global batch grows with N; `extra.strong` carries the same step with the 65,536 rays split over the ranks),
L=10/4 positional-encoding degrees, hidden 256, colour-layer width 128.

Precision.  The MLP computes in bf16 with fp32 accumulation throughout (forward chain, input-gradient chain).  The
weight-gradient pass reads saved copies of its operands, and the format of those copies is what `--wgrad-operands`
selects: `bf16` (16-bit copies: bf16 x bf16 products, the arithmetic the reference's 'medium' matmul precision
allows, train.py:36) or `mx` (activations e4m3, gradients block-scaled e2m3: half the bytes and less).  Without the
flag BOTH are timed in this process from the same seeds: `value` / `ms_per_step` / `dtype` / `roofline` are the bf16
step, `extra.mx` holds the MX step.  With the flag only that format runs and `dtype` names it.

Prints ONE JSON line on rank 0: whole-job rays/s, plus a roofline object for the dominant kernel family
(HIP-event timing on the launching stream inside the timed region) and a CPU baseline (the numpy/torch
oracle timed on a bounded ray sample on the host cores).
"""
import argparse
import hashlib
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 matrix peak
MFMA_FP8_PEAK_TFLOPS = 5000.0      # dense fp8 peak: the rate of the block-scaled product with one fp8 operand
HBM_PEAK_GBS = 8000.0              # MI355X HBM3E peak
DTYPE_NAME = {'bf16': 'bf16', 'mx': 'bf16 chain + mx(e4m3 x e2m3) wgrad', 'fp32': 'fp32'}


def make_cfg(precision='bf16'):
    return SimpleNamespace(use_viewdirs=True, randomized=True, ray_shape='cone', white_bkgd=False, num_levels=2,
                           num_samples=128, level_samples=[64, 128], min_deg=0, max_deg=10, viewdirs_min_deg=0,
                           viewdirs_max_deg=4, density_noise=0.0, rgb_padding=0.001, resample_padding=0.01,
                           density_bias=-1.0, hidden=256, coarse_weight_decay=0.1, lr_init=1e-3, lr_final=5e-5,
                           max_steps=100000, lr_delay_steps=2500, lr_delay_mult=0.1, weight_decay=1e-5,
                           mlp_precision=precision)


PMC_TRAFFIC = 'r03_pmc_traffic.json'
KERNEL_SOURCES = ('nerf_amd/csrc', 'nerf_amd/stepper.py', 'nerf_amd/mlp_bf16.py')


def kernel_source_hash():
    """sha256 over the kernel sources and the launch sequence: what a committed counter pass must have been taken
    from for its byte counts to describe this tree."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        path = os.path.join(ROOT, rel)
        files = sorted(os.path.join(path, f) for f in os.listdir(path)) if os.path.isdir(path) else [path]
        for f in files:
            if os.path.isfile(f):
                h.update(os.path.basename(f).encode())
                h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, rays, operands):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same workload and format
    (profiles/<PMC_TRAFFIC>, made by tools/pmc_summary.py), with where they come from; (None, reason) if there is no
    such pass, if it was taken at another batch size or format, or from other kernel sources than this tree's."""
    path = os.path.join(ROOT, 'profiles', PMC_TRAFFIC)
    if not os.path.exists(path):
        return None, 'no committed counter pass'
    doc = json.load(open(path))
    sec = doc.get('formats', {}).get(operands)
    if sec is None or rays != doc.get('rays', 65536):
        return None, 'no committed counter pass for this batch size / format'
    if doc.get('source_hash') != kernel_source_hash():
        return None, f'profiles/{PMC_TRAFFIC} was taken from other kernel sources than this tree'
    for name, rec in sec.items():
        if name == kernel or name.startswith(kernel + '<'):
            return rec['traffic_bytes_per_launch'], (f'profiles/{PMC_TRAFFIC}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes '
                                                     f'of this workload (committed, same kernel sources {doc["source_hash"]}; '
                                                     'not collected in this run)')
    return None, 'kernel not in the committed counter pass'


def make_sar_cfg(precision='bf16'):
    """The reference's SAR-NeRF hyperparameters (params.yaml:67-93)."""
    return SimpleNamespace(use_viewdirs=True, randomized=True, ray_shape='cone', num_levels=2, num_samples=180,
                           hidden=156, density_noise=0.0, density_bias=-1.0, rgb_padding=1e-4, resample_padding=1e-3,
                           min_deg=0, max_deg=4, viewdirs_min_deg=0, viewdirs_max_deg=4, rays_per_chunk=2048,
                           total_num_rays=2048, wavelength=0.031228381041666666, coarse_weight_decay=0.01,
                           lr_init=.5, lr_final=5e-5, lr_delay_steps=5500, lr_delay_mult=0.05, weight_decay=1e-6,
                           max_steps=2500, white_bkgd=False, mlp_precision=precision, check_range=False)


def sar_batch(dev, nsam=2048, nfft=8192, seed=0):
    """One synthetic pulse: platform at (0, -1000, 500) m, 0.5 m range bins from 900 m, random matched filter and
    target profile; far = near + mpp (nsam - 2) keeps every range-bin index inside the profile."""
    g = torch.Generator().manual_seed(seed)
    near, mpp = torch.tensor([900.]), torch.tensor([0.5])
    far = near + mpp * (nsam - 2)
    mfilt = torch.complex(torch.randn(1, nfft, generator=g), torch.randn(1, nfft, generator=g))
    target = torch.randn(1, nsam, 2, generator=g)
    ray_o = torch.tensor([[0., -1000., 500.]])
    return tuple(x.to(dev) if torch.is_tensor(x) else x for x in
                 (ray_o, None, None, mfilt, torch.tensor([1.12]), near, far, mpp, target))


def sar_pulse_step(model, batch, opt, world):
    """One SAR training step on this rank's own pulse; with several ranks the gradient bucket is averaged (a batch
    of `world` pulses, where the reference's loader delivers one, params.yaml:99)."""
    from nerf_amd import stepper
    from nerf_amd.train import grad_bucket, sar_train_step
    if world == 1:
        return sar_train_step(model, batch, opt)
    ray_o, pan, tilt, mfilt, radii, near, far, mpp, target = batch
    bucket = grad_bucket(model)
    bucket.zero()
    if stepper.sar_supported(model):
        loss = stepper.sar_stepper_for(model).step(ray_o, mfilt, radii, near, far, target.shape[1], mpp, target)
        bucket.flat.mul_(1.0 / world)
    else:
        pulses = model(ray_o, pan, tilt, mfilt, radii, near, far, target.shape[1], mpp)[0]
        loss, _ = model.loss_function(pulses, target)
        (loss / world).backward()
    bucket.attach()
    bucket.all_reduce(world)
    opt.step()
    model._weights_changed()
    return loss.detach()


def mlp_default_operands():
    from nerf_amd import mlp_bf16
    return mlp_bf16.DEFAULT_WGRAD_OPERANDS


def launch_ranks(n, argv):
    """Start n ranks of this script (one per GPU) and relay their output; nothing here touches a GPU."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and '--rehearse-one-gpu' not in argv:
        raise SystemExit(f'bench.py --gpus {n}: only {have} HIP device(s) visible')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(budget_rays=2048, threads=None):
    """The CPU oracle (fp32 restatement of the reference path) on a bounded sample of the same workload: the full
    step (fwd + loss + bwd + AdamW) at 64+128 samples per ray, all host cores torch gives us."""
    from oracle import nerf_oracle as O
    if threads:
        torch.set_num_threads(threads)
    cfg = make_cfg('fp32')
    p = {k: v.requires_grad_(True) for k, v in O.init_params(cfg, seed=0).items()}
    g = torch.Generator().manual_seed(0)
    n = budget_rays
    o = 0.1 * torch.randn(n, 3, generator=g) + torch.tensor([0., 0., 4.])
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    rays = O.Rays(o, d, d.clone(), torch.full((n, 1), 0.005), torch.ones(n, 1), torch.full((n, 1), 2.),
                  torch.full((n, 1), 6.))
    tgt = torch.rand(n, 3, generator=g)
    eps = float(torch.finfo(torch.float32).eps)
    opt = torch.optim.AdamW(list(p.values()), lr=cfg.lr_init, weight_decay=cfg.weight_decay)

    def step():
        t_rand = torch.rand(n, 65, generator=g)
        u_rand = {1: torch.empty(n, 129).uniform_(0, 1 / 129 - eps, generator=g)}
        out = O.nerf_render(p, rays, cfg, [64, 128], t_rand, u_rand)
        loss, _ = O.nerf_loss(out['comp_rgb_stack'], tgt, rays.lossmult, cfg.coarse_weight_decay)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    step()
    t0 = time.perf_counter()
    reps = 0
    while reps < 3 or time.perf_counter() - t0 < 10.0:
        step()
        reps += 1
        if time.perf_counter() - t0 > 30.0:
            break
    dt = (time.perf_counter() - t0) / reps
    return dict(value=n / dt, unit='rays/s', cores=torch.get_num_threads(), kind='port',
                sample=f'{n} rays x (64+128) samples per step, fwd+loss+bwd+AdamW, {reps} steps timed, fp32 torch-CPU oracle')


def roofline_of(prof, steps, rays, operands, chunked):
    """-> (kernel_ms_per_step, roofline of the kernel symbol with the most time per step, reference-equivalent and
    executed matrix-core utilisation of the MLP kernels)."""
    per = {k: v['ms'] / steps for k, v in prof.items()}
    dom = max(per, key=per.get)
    r = prof[dom]
    avg_ms = r['ms'] / r['calls']
    if dom.startswith('mlp_wgrad'):      # streams its operands once from HBM: bytes / time
        achieved = r['bytes'] / (r['ms'] * 1e-3) / 1e9
        roof = {'kernel': dom, 'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': achieved / HBM_PEAK_GBS, 'algorithmic_bytes_per_launch': r['bytes'] / r['calls']}
    else:                                 # weights in LDS, activations in registers: FLOPs / time
        achieved = r['flops'] / (r['ms'] * 1e-3) / 1e12
        roof = {'kernel': dom, 'bound': 'mfma', 'achieved': achieved, 'peak': MFMA_BF16_PEAK_TFLOPS,
                'unit': 'TFLOP/s', 'frac': achieved / MFMA_BF16_PEAK_TFLOPS,
                'algorithmic_flops_per_launch': r['flops'] / r['calls'],
                'executed_flops_per_launch': r['flops_exec'] / r['calls'],
                'frac_executed': r['flops_exec'] / (r['ms'] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS}
        if r['bytes'] > 0:                # the training chain kernels also stream saved activations to HBM
            roof['stores'] = {'achieved': r['bytes'] / (r['ms'] * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                              'frac': r['bytes'] / (r['ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              'bytes_saved_per_launch': r['bytes'] / r['calls']}
    traffic, source = (None, 'chunked step') if chunked else pmc_traffic(dom, rays, operands)
    roof.update(avg_launch_ms=avg_ms, launches_per_step=r['calls'] / steps, traffic=traffic, traffic_source=source)
    mlp = [k for k in prof if k.startswith('mlp_')]
    mlp_ms = sum(prof[k]['ms'] for k in mlp)
    util = util_exec = None
    if mlp_ms > 0:
        # reference-equivalent: the reference's MLP FLOPs (rgb_net0 counted although it is folded away) per dense
        # bf16 peak FLOP; executed: the FLOPs the kernels issue for true layer sizes, each kernel priced at the dense
        # peak of the instruction it issues (bf16 products 2.5 PFLOP/s, block-scaled fp6 x fp8 products 5 PFLOP/s)
        util = sum(prof[k]['flops'] for k in mlp) / (mlp_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS
        pipe_s = sum(prof[k]['flops_exec'] / ((MFMA_FP8_PEAK_TFLOPS if k.startswith('mlp_wgrad8') else MFMA_BF16_PEAK_TFLOPS) * 1e12)
                     for k in mlp)
        util_exec = pipe_s / (mlp_ms * 1e-3)
    return {k: round(v, 3) for k, v in per.items()}, roof, util, util_exec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--rays', type=int, default=65536, help='rays per step per GPU (weak scaling) / per step (strong)')
    ap.add_argument('--chunk', type=int, default=65536, help='rays per forward/backward chunk')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--wgrad-operands', default=None, choices=['bf16', 'mx', 'e4m3'],
                    help='format of the saved weight-gradient operands: bf16, or mx = activations e4m3 + gradients '
                         'block-scaled e2m3 (e4m3 is an older name of mx).  Default for c3: both are timed, the bf16 step '
                         'is the headline and the mx step goes to extra.mx; the MLP itself computes in bf16 either way')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                    help='c3 with N ranks: weak = --rays per rank (default; the strong figure rides in extra.strong), '
                         'strong = --rays in total, split over the ranks')
    ap.add_argument('--rehearse-one-gpu', action='store_true',
                    help='N>1 ranks all on cuda:0 over gloo: a rehearsal of the multi-rank code path on a one-GPU box '
                         '(its numbers mean nothing)')
    ap.add_argument('--graph', default='auto', choices=['auto', 'on', 'off'],
                    help='c4 only: replay the pulse step as one captured HIP graph (auto: on for one rank; the per-kernel '
                         'timings then come from a few launch-by-launch steps ahead of the capture)')
    ap.add_argument('--workload', default='c3', choices=['c3', 'c2', 'c4'],
                    help="c3 (default, the headline metric): full train step; c2: forward-only render (use --rays 1048576); "
                         "c4: SAR-NeRF pulse step at the reference's params.yaml sizes (2048 rays x 180+180 samples)")
    args = ap.parse_args()
    if args.wgrad_operands == 'e4m3':
        args.wgrad_operands = 'mx'

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:       # no launcher around us: be the launcher
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (the render path has no CPU fallback)')
    if args.rehearse_one_gpu:
        import faulthandler
        faulthandler.dump_traceback_later(300, exit=True)        # a rehearsal that hangs says where
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.rehearse_one_gpu:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from nerf_amd import profiling
    from nerf_amd.model import MipNeRF, SARNeRF
    from nerf_amd.train import nerf_train_step, synthetic_rays

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(step, profile=True):
        """W untimed warm-up steps, then exactly K steps between two fences; the maximum over the ranks."""
        for _ in range(args.warmup):
            step()
        fence()
        profiling.enable(profile)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        fence()
        dt = time.perf_counter() - t0
        prof = profiling.summary() if profile else {}
        profiling.enable(False)
        if world > 1:                               # (gloo rehearsal: host tensors, see train._all_reduce_sum)
            tmax = torch.tensor([dt], device='cpu' if args.rehearse_one_gpu else dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax)
        return dt, prof, float(loss)

    def result(dt, prof, loss, rays_per_rank, operands):
        out = {'value': rays_per_rank * world * args.steps / dt, 'ms_per_step': dt / args.steps * 1e3,
               'rays_per_step_per_gpu': rays_per_rank, 'global_rays_per_step': rays_per_rank * world, 'loss': loss}
        if prof:
            per, roof, util, util_exec = roofline_of(prof, args.steps, rays_per_rank, operands,
                                                     args.workload == 'c3' and args.chunk < rays_per_rank)
            out.update(kernel_ms_per_step=per, roofline=roof)
            if util is not None:
                out.update(mlp_mfma_util=util, mlp_mfma_util_executed=util_exec)
        return out

    def new_model(operands):
        """The same initial weights for every format / scaling: seeded construction, rank 0's broadcast."""
        torch.manual_seed(0)
        sar = args.workload == 'c4'
        cfg = make_sar_cfg(args.precision) if sar else make_cfg(args.precision)
        cfg.wgrad_operands = operands
        model = (SARNeRF(cfg) if sar else MipNeRF(cfg)).to(dev).train()
        if world > 1:
            for p in model.parameters():
                if args.rehearse_one_gpu:           # gloo's device path blocks behind queued kernels: stage through the host
                    t = p.data.cpu()
                    dist.broadcast(t, 0)
                    p.data.copy_(t)
                else:
                    dist.broadcast(p.data, 0)
        return model, cfg

    def run_c3(operands, rays_per_rank):
        model, cfg = new_model(operands)
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(params, lr=cfg.lr_init, weight_decay=cfg.weight_decay, fused=True)
        rays, pixels = synthetic_rays(rays_per_rank, dev, seed=1234 + rank)
        mask_sum = rays.lossmult.sum() * world
        torch.manual_seed(1 + rank)                 # the sampling jitter: the same draws for every format
        res = result(*timed_steps(lambda: nerf_train_step(model, rays, pixels, opt, chunk=args.chunk, world_size=world,
                                                          global_mask_sum=mask_sum)), rays_per_rank, operands)
        model._stepper = model._packed = None
        del opt, model
        torch.cuda.empty_cache()                    # the next format's persistent buffers take this one's place
        return res

    head = {'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': True,
            'scaling': args.scaling if args.workload == 'c3' else 'weak', 'vs_baseline': None, 'data': 'synthetic'}
    parallelism = f'ray-sharded dp{world}, flat-bucket grad all-reduce'
    out = None

    if args.workload == 'c3':
        formats = [args.wgrad_operands] if args.wgrad_operands or args.precision == 'fp32' else ['bf16', 'mx']
        if args.precision == 'fp32':
            formats = [None]
        weak_rays = args.rays
        strong_rays = max(1, args.rays // world)
        lead_rays, other = (weak_rays, strong_rays) if args.scaling == 'weak' else (strong_rays, weak_rays)
        res = {}
        for f in formats:
            res[f] = run_c3(f, lead_rays)
            if world > 1:                           # the other scaling mode as a second figure
                o = run_c3(f, other)
                res[f]['strong' if args.scaling == 'weak' else 'weak'] = {
                    k: o[k] for k in ('value', 'ms_per_step', 'rays_per_step_per_gpu', 'global_rays_per_step', 'kernel_ms_per_step',
                                      'mlp_mfma_util') if k in o}
        if rank == 0:
            lead = res[formats[0]]
            out = dict(metric='rays/sec fwd+bwd (64+128 samples/ray)', value=lead['value'], **head,
                       ms_per_step=lead['ms_per_step'],
                       dtype=DTYPE_NAME[formats[0] or 'fp32'],
                       config={'workload': 'C3: full NeRF train step (fwd+bwd+AdamW), coarse 64 + fine 128 samples/ray, '
                                           '8x256 MLP (colour width 128), L=10/4 posenc',
                               'rays_per_step_per_gpu': lead['rays_per_step_per_gpu'],
                               'global_rays_per_step': lead['global_rays_per_step'], 'chunk_rays': args.chunk,
                               'wgrad_operands': formats[0], 'parallelism': parallelism, 'loss': lead['loss']})
            for k in ('kernel_ms_per_step', 'roofline', 'mlp_mfma_util', 'mlp_mfma_util_executed'):
                if k in lead:
                    out[k] = lead[k]
            extra = {}
            for k in ('strong', 'weak'):
                if k in lead:
                    extra[k] = lead[k]
            for f in formats[1:]:
                extra[f] = dict(res[f], dtype=DTYPE_NAME[f], wgrad_operands=f,
                                note='same process, seeds, rays and step count as the headline; only the format of the '
                                     'saved weight-gradient operands differs')
            if extra:
                out['extra'] = extra
            if world == 1 and not args.no_cpu_baseline:
                out['cpu_baseline'] = cpu_baseline()

    elif args.workload == 'c2':
        model, cfg = new_model(args.wgrad_operands)
        rays, pixels = synthetic_rays(args.rays, dev, seed=1234 + rank)

        def step():                                 # coarse+fine render only, nothing saved for a backward pass
            with torch.no_grad():
                return model(rays)[0].sum()
        r = result(*timed_steps(step), args.rays, None)
        if rank == 0:
            out = dict(metric='rays/sec fwd (64+128 samples/ray)', value=r['value'], **head, ms_per_step=r['ms_per_step'],
                       dtype=args.precision,
                       config={'workload': 'C2: forward-only hierarchical render, coarse 64 + fine 128 samples/ray, 8x256 MLP '
                                           '(colour width 128), L=10/4 posenc',
                               'rays_per_step_per_gpu': args.rays, 'global_rays_per_step': args.rays * world,
                               'parallelism': f'ray-sharded dp{world}, no collective'})
            for k in ('kernel_ms_per_step', 'roofline', 'mlp_mfma_util', 'mlp_mfma_util_executed'):
                if k in r:
                    out[k] = r[k]

    else:                                           # c4
        operands = args.wgrad_operands or mlp_default_operands()
        model, cfg = new_model(operands)
        params = [p for p in model.parameters() if p.requires_grad]
        # one pulse per step and rank (ranks take different pulses: weak scaling); the reference's lr_init of 0.5
        # diverges on synthetic targets within a few steps, so the timed steps use a small fixed rate
        graphed = args.graph == 'on' or (args.graph == 'auto' and world == 1)
        if graphed and world > 1:
            raise SystemExit('--graph on needs one rank (the captured step holds no collective)')
        lr = torch.tensor(1e-4, device=dev) if graphed else 1e-4
        opt = torch.optim.AdamW(params, lr=lr, weight_decay=cfg.weight_decay, fused=True, capturable=graphed)
        batch = sar_batch(dev, seed=100 + rank)
        R = cfg.total_num_rays
        eager = result(*timed_steps(lambda: sar_pulse_step(model, batch, opt, world)), R, operands)
        r = eager
        if graphed:     # (HIP events around single launches mean nothing inside a replay: the kernel times are the eager pass's)
            from nerf_amd.train import GraphedSarStep
            gstep = GraphedSarStep(model, batch, opt)
            r = result(*timed_steps(gstep.step, profile=False), R, operands)
        if rank == 0:
            out = dict(metric='rays/sec fwd+bwd (SAR pulse, 180+180 samples/ray)', value=r['value'], **head,
                       ms_per_step=r['ms_per_step'], dtype=DTYPE_NAME[operands],
                       config={'workload': 'C4: SAR-NeRF pulse step (fwd+bwd+AdamW), 2048 rays x (180+180) samples, H=156, '
                                           'W=180, L=4/4, nsam 2048, Nfft 8192',
                               'rays_per_step_per_gpu': R, 'global_rays_per_step': R * world, 'wgrad_operands': operands,
                               'parallelism': parallelism, 'hip_graph': bool(graphed), 'loss': r['loss']})
            for k in ('kernel_ms_per_step', 'roofline', 'mlp_mfma_util', 'mlp_mfma_util_executed'):
                if k in eager:
                    out[k] = eager[k]
            if graphed:
                out['extra'] = {'launch_by_launch': {'value': eager['value'], 'ms_per_step': eager['ms_per_step'],
                                                     'note': 'kernel_ms_per_step, roofline and mlp_mfma_util are from this '
                                                             'pass (HIP events per launch); value is the graph replay'}}
                # utilisation of the replayed step: the same kernels and work, on the replay's clock
                if 'mlp_mfma_util' in eager:
                    mlp_ms = sum(v for k, v in eager['kernel_ms_per_step'].items() if k.startswith('mlp_'))
                    out['extra']['mlp_share_of_replay'] = mlp_ms / r['ms_per_step']

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
