"""Does the HBM-bound weight-gradient pass of one half of a batch run faster side by side with the compute-bound
backward chain of the other half?  Two halves A, B of 32768 rays x 128 samples each."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_cfg
from nerf_amd.model import MipNeRF
from nerf_amd import mlp_bf16, ops
from nerf_amd.train import synthetic_rays

fmt = os.environ.get('NERF_FMT', 'mx')
cfg = make_cfg('bf16'); cfg.wgrad_operands = fmt
m = MipNeRF(cfg).cuda()
n, S = 32768, 128
M = n * S
pk = mlp_bf16.packed_for(m, True)
halves = []
for h in range(2):
    rays, _ = synthetic_rays(n, 'cuda')
    t = ops.stratified_t(rays.near, rays.far, S, torch.rand(n, S + 1, device='cuda'))
    encf = mlp_bf16.encode_frag(t, rays.origins, rays.directions, rays.radii, 'cone', cfg.min_deg, cfg.max_deg)
    viewf = mlp_bf16.view_frag(rays.viewdirs, 0, 4)
    d, rgb, _, acts, _ = mlp_bf16.mlp_forward(pk, encf, viewf, M, S, n, save=True)
    gd, gr = torch.randn(M, device='cuda'), torch.randn(M, 3, device='cuda')
    dys = mlp_bf16.mlp_backward(pk, acts, M, gd, gr, rgb)
    halves.append(dict(acts=acts, rgb=rgb, gd=gd, gr=gr, dys=dys))
grads = {k: torch.zeros_like(v) for k, v in m.named_parameters()}
ws2 = torch.empty(mlp_bf16.WGRAD_WORKSPACE_BYTES, dtype=torch.uint8, device='cuda')
side = torch.cuda.Stream()

def bwd(h): mlp_bf16.mlp_backward(pk, h['acts'], M, h['gd'], h['gr'], h['rgb'], dys=h['dys'])
def wg(h, grid=0, ws=None): mlp_bf16.mlp_weight_grads(pk, h['acts'], h['dys'], M, grads, grid_hint=grid, workspace=ws)

def serial():
    bwd(halves[0]); bwd(halves[1]); wg(halves[0]); wg(halves[1])

def overlapped(grid_a, grid_b):
    def f():
        main = torch.cuda.current_stream()
        bwd(halves[0])
        ev = torch.cuda.Event(); ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            wg(halves[0], grid_a, ws2)
        bwd(halves[1])
        wg(halves[1], grid_b)
        ev2 = torch.cuda.Event(); ev2.record(side)
        main.wait_event(ev2)
    return f

def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

print(fmt, 'bwd one half %.2f ms' % timeit(lambda: bwd(halves[0])), 'wgrad one half %.2f ms' % timeit(lambda: wg(halves[0])), flush=True)
print('serial (bwd A, bwd B, wgrad A, wgrad B): %.2f ms' % timeit(serial), flush=True)
for ga, gb in ((128, 0), (96, 0), (64, 0), (160, 0), (128, 128), (192, 0), (256, 0)):
    print('overlap, wgrad A on %3d workgroups beside bwd B, wgrad B on %3d: %.2f ms' % (ga, gb or 256, timeit(overlapped(ga, gb))), flush=True)
print('serial again: %.2f ms' % timeit(serial), flush=True)
