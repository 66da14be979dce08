import sys, time, torch
sys.path.insert(0, '.')
from types import SimpleNamespace
from nerf_amd import mlp_bf16
from nerf_amd.model import MipNeRF
from tests.test_gpu_mlp_bf16 import make_cfg
cfg = make_cfg()
m = MipNeRF(cfg).cuda()
pk = mlp_bf16.packed_for(m, need_bwd=False)
n, S = 32768, 128
M = n * S
t = torch.sort(torch.rand(n, S + 1, device='cuda') * 4 + 2, -1)[0]
o = torch.randn(n, 3, device='cuda') * 0.1; d = torch.nn.functional.normalize(torch.randn(n, 3, device='cuda'), dim=-1)
r = torch.full((n,), 0.005, device='cuda')
def timeit(f, it=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
t_enc = timeit(lambda: mlp_bf16.encode_frag(t, o, d, r, 'cone', 0, 10))
encf = mlp_bf16.encode_frag(t, o, d, r, 'cone', 0, 10)
viewf = mlp_bf16.view_frag(d, 0, 4)
flop = 1.1837e6 * M
for save in (False, True):
    ms = timeit(lambda: mlp_bf16.mlp_forward(pk, encf, viewf, M, S, n, save=save))
    print(f'fwd save={save}: {ms:.2f} ms  {flop/ms/1e9:.1f} TFLOP/s  ({flop/ms/1e9/2500*100:.1f}% of 2.5PF)  {M/ms/1e3:.1f} Msamples/s')
print(f'encode_frag: {t_enc:.2f} ms  ({M*128/t_enc/1e6:.1f} GB/s written)')
