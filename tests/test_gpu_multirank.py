"""Multi-rank paths on GPUs: the RCCL train step on two devices (skipped on a one-GPU box) and bench.py's own
rank launcher (rehearsed with both ranks on cuda:0 over gloo where only one GPU exists)."""
import json
import os
import socket
import subprocess
import sys
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg():
    return SimpleNamespace(use_viewdirs=True, randomized=False, ray_shape='cone', white_bkgd=False, num_levels=2,
                           num_samples=128, level_samples=[64, 128], min_deg=0, max_deg=10, viewdirs_min_deg=0,
                           viewdirs_max_deg=4, density_noise=0.0, rgb_padding=0.001, resample_padding=0.01,
                           density_bias=-1.0, hidden=256, coarse_weight_decay=0.1, lr_init=1e-3, lr_final=5e-5,
                           max_steps=100000, lr_delay_steps=2500, lr_delay_mult=0.1, weight_decay=1e-5,
                           mlp_precision='bf16')


def _step(rank, world, n=4096):
    """One deterministic train step (weights kept: lr 0) on this rank's shard of an n-ray batch -> loss, gradients."""
    from nerf_amd.model import MipNeRF
    from nerf_amd.train import nerf_train_step, shard_range, synthetic_rays
    from nerf_amd.utils import namedtuple_map
    dev = torch.device('cuda', rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    m = MipNeRF(_cfg()).to(dev).train()
    rays, pix = synthetic_rays(n, dev, seed=9)
    lo, hi = shard_range(n, rank, world)
    shard = namedtuple_map(lambda t: t[lo:hi], rays)
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    loss = nerf_train_step(m, shard, pix[lo:hi], opt, chunk=n, world_size=world)       # mask sum all-reduced inside
    grads = torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.requires_grad])
    return float(loss), grads.cpu().numpy()


def _nccl_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    loss, grads = _step(rank, world)
    t = torch.tensor([loss], device='cuda', dtype=torch.float64)
    dist.all_reduce(t)                                   # the ranks' shares of the loss add up to the batch loss
    q.put((rank, float(t), grads))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (RCCL all-reduce between devices)')
def test_two_rank_rccl_step_matches_one_rank():
    """Ray-sharded step on two GPUs over RCCL: the all-reduced gradient bucket and the summed loss equal those of
    one rank stepping on the whole batch (fp32 atomics reorder sums)."""
    import numpy as np
    import torch.multiprocessing as mp
    one_loss, one_grads = _step(0, 1)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = []
    try:
        for _ in range(2):
            got.append(q.get(timeout=240))
    finally:
        for pr in procs:
            pr.join(60)
            if pr.is_alive():
                pr.kill()
    assert all(pr.exitcode == 0 for pr in procs)
    for _, loss, grads in got:
        assert abs(loss - one_loss) <= 1e-4 * abs(one_loss)
        assert np.linalg.norm(grads - one_grads) <= 2e-3 * np.linalg.norm(one_grads)
    assert np.array_equal(got[0][2], got[1][2])


def _nccl_single_worker(port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    assert dist.get_backend() == 'nccl'
    from nerf_amd.model import MipNeRF
    from nerf_amd.train import nerf_train_step, synthetic_rays
    dev = torch.device('cuda', 0)
    out = []
    for world in (1, 2):            # 2: every collective of the multi-rank step is issued (over the group's one rank)
        torch.manual_seed(0)
        m = MipNeRF(_cfg()).to(dev).train()
        rays, pix = synthetic_rays(4096, dev, seed=9)
        opt = torch.optim.SGD(m.parameters(), lr=0.0)
        loss = nerf_train_step(m, rays, pix, opt, chunk=4096, world_size=world)
        out.append((float(loss), torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.requires_grad]).cpu().numpy()))
    # the other collectives bench.py issues around the step: parameter broadcast, max of the ranks' times, barrier
    for p in m.parameters():
        dist.broadcast(p.data, 0)
    tmax = torch.tensor([1.25], device=dev, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    assert float(tmax) == 1.25
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    q.put(out)


def test_one_rank_rccl_group_runs_the_collective_path():
    """What a one-GPU box can check of the RCCL path: in a one-rank `nccl` group the multi-rank step (mask-sum
    all-reduce, the trunk gradients' asynchronous all-reduce issued from inside the weight-gradient pass, the rest of
    the bucket, both waited for ahead of the optimizer, a barrier) runs on the real backend and, the sums being over
    one rank, leaves the loss and the gradients of the plain step."""
    import numpy as np
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    pr = ctx.Process(target=_nccl_single_worker, args=(port, q))
    pr.start()
    try:
        (loss1, g1), (loss2, g2) = q.get(timeout=240)
    finally:
        pr.join(60)
        if pr.is_alive():
            pr.kill()
    assert pr.exitcode == 0
    assert abs(loss1 - loss2) <= 1e-6 * abs(loss1)
    assert np.isfinite(g2).all() and np.linalg.norm(g2 - g1) <= 1e-4 * np.linalg.norm(g1)


def test_bench_launches_its_own_ranks():
    """`bench.py --gpus 2` with no launcher around it starts two ranks itself and rank 0 prints one JSON line with
    n_gpus 2.  With two GPUs this is the RCCL path; on a one-GPU box both ranks share cuda:0 over gloo (a rehearsal
    of the launcher and of the multi-rank step, not a measurement)."""
    two = torch.cuda.device_count() >= 2
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--rays',
           '4096', '--chunk', '4096', '--no-cpu-baseline'] + ([] if two else ['--rehearse-one-gpu'])
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['global_rays_per_step'] == 8192
    assert rec['value'] > 0 and rec['config']['loss'] == rec['config']['loss']
