"""Time the fused MLP forward (inference / training), backward chain and weight-gradient pass on C3-shaped input.
NERF_FMT=bf16|mx picks the format of the saved operands, NERF_AMD_LIB another build of the library, NERF_ZERO=1 zero weights."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_cfg
from nerf_amd.model import MipNeRF
from nerf_amd import mlp_bf16
from nerf_amd.train import synthetic_rays

def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

cfg = make_cfg('bf16')
if os.environ.get('NERF_FMT'): cfg.wgrad_operands = os.environ['NERF_FMT']
m = MipNeRF(cfg).cuda()
if os.environ.get('NERF_ZERO'):
    for p_ in m.parameters():
        if p_.requires_grad: p_.data.zero_()
n, S = 32768, 128
rays, _ = synthetic_rays(n, 'cuda')
from nerf_amd import ops
t = ops.stratified_t(rays.near, rays.far, S, torch.rand(n, S + 1, device='cuda'))
pk = mlp_bf16.packed_for(m, True)
encf = mlp_bf16.encode_frag(t, rays.origins, rays.directions, rays.radii, 'cone', cfg.min_deg, cfg.max_deg)
viewf = mlp_bf16.view_frag(rays.viewdirs, 0, 4)
M = n * S
for save in (False, True):
    ms = timeit(lambda: mlp_bf16.mlp_forward(pk, encf, viewf, M, S, n, save=save))
    print(f"fwd save={save}: {ms:.3f} ms  {pk.fwd_flops * M / ms / 1e9:.0f} TFLOP/s", flush=True)
d, rgb, _, acts, _ = mlp_bf16.mlp_forward(pk, encf, viewf, M, S, n, save=True)
gd, gr = torch.randn(M, device='cuda'), torch.randn(M, 3, device='cuda')
ms = timeit(lambda: mlp_bf16.mlp_backward(pk, acts, M, gd, gr, rgb))
print(f"bwd: {ms:.3f} ms  {pk.bwd_dx_flops * M / ms / 1e9:.0f} TFLOP/s", flush=True)
dys = mlp_bf16.mlp_backward(pk, acts, M, gd, gr, rgb)
grads = {k: torch.zeros_like(v) for k, v in m.named_parameters()}
ms = timeit(lambda: mlp_bf16.mlp_weight_grads(pk, acts, dys, M, grads))
print(f"wgrad: {ms:.3f} ms  {pk.wgrad_bytes * M / ms / 1e6:.0f} GB/s", flush=True)

