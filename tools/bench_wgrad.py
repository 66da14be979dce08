"""Time every weight-gradient launch of one backward pass separately (8.39 M samples, C3 network)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nerf_amd._lib as _L
if os.environ.get('NERF_LIB'):
    _L.LIB_PATH = os.environ['NERF_LIB']
from bench import make_cfg
from nerf_amd import mlp_bf16, ops
from nerf_amd.model import MipNeRF
from nerf_amd.train import synthetic_rays

cfg = make_cfg('bf16')
if os.environ.get('NERF_FMT'):
    cfg.wgrad_operands = os.environ['NERF_FMT']
m = MipNeRF(cfg).cuda()
n, S = int(os.environ.get('NERF_RAYS', 65536)), 128
rays, _ = synthetic_rays(n, 'cuda')
t = ops.stratified_t(rays.near, rays.far, S, torch.rand(n, S + 1, device='cuda'))
pk = mlp_bf16.packed_for(m, True)
encf = mlp_bf16.encode_frag(t, rays.origins, rays.directions, rays.radii, 'cone', cfg.min_deg, cfg.max_deg)
viewf = mlp_bf16.view_frag(rays.viewdirs, 0, 4)
M = n * S
d, rgb, _, acts, _ = mlp_bf16.mlp_forward(pk, encf, viewf, M, S, n, save=True)
dys = mlp_bf16.mlp_backward(pk, acts, M, torch.randn(M, device='cuda') * 1e-4, torch.randn(M, 3, device='cuda') * 1e-4, rgb)
grads = {k: torch.zeros_like(v) for k, v in m.named_parameters()}
grads['_g98.weight'], grads['_g98.bias'] = torch.zeros(pk.W, pk.H, device='cuda'), torch.zeros(pk.W, device='cuda')
plan = mlp_bf16._wgrad_plan(pk)
tot_ms = tot_b = 0.0
for k, item in enumerate(plan):
    mlp_bf16._wgrad_plan = lambda _pk, item=item: [item]
    f = lambda: mlp_bf16._wgrad_launches(pk, acts, dys, M, grads)
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    nbytes = mlp_bf16._wgrad_launch_bytes(pk, item[0], item[1]) * M
    tot_ms, tot_b = tot_ms + ms, tot_b + nbytes
    nf, kf = sum(s[1] for s in item[0]), sum(s[1] for s in item[1])
    segs = (tuple(s[1] for s in item[0]), tuple(s[1] for s in item[1])) if pk.fmt else (nf, kf)
    print(f'launch {k} ({nf:3d} x {kf:3d}) {mlp_bf16.wgrad_kernel_name(*segs, pk.fmt):32s} {ms:7.3f} ms {nbytes / ms / 1e6:7.0f} GB/s', flush=True)
print(f'all launches: {tot_ms:.3f} ms, {tot_b / tot_ms / 1e6:.0f} GB/s')
