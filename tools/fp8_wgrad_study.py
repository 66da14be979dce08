"""What 8- and 6-bit copies of the weight-gradient operands cost in accuracy, measured on the kernels' own buffers.

Runs one fused training step on the 1024-ray fixture (tests/golden/nerf_e2e_c3_1k.npz: 196 608 samples, 8x256
network), unpacks the saved activations X and the gradients dY from their transposed-fragment buffers and forms
dW = dY^T X per trunk layer in fp32 from: the bf16 operands as stored (what the kernels compute), X rounded to OCP
e4m3, dY rounded to e4m3 with one power-of-two scale per (feature, 32-sample block) (the MX block-scale layout
along the reduction axis), both; dY rounded to block-scaled e2m3 (what the training step stores), with X in e4m3
(what it stores) and with X in block-scaled e2m3 as well.  Prints the relative Frobenius error of each variant against
the bf16-operand dW and the error of the first three against the reference's fp32 gradient sample stored in the
fixture.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.conftest import T, cfg_from, load_golden     # noqa: E402
from oracle import nerf_oracle as O                       # noqa: E402  (initial weights only)


def tfrag_dense(buf, F, M):
    """[16-sample k-step][F/32 tiles][2 lane halves][32 features][8] bf16 -> [M, F] fp32 (on the device)."""
    NT = F // 32
    x = buf.view(torch.bfloat16).reshape(-1, NT, 2, 32, 2, 4)          # q, tile, hh, n, jh, jl
    x = x.permute(0, 4, 2, 5, 1, 3).reshape(-1, F)                      # sample = 16 q + 8 jh + 4 hh + jl
    return x[:M].float()


def e4m3(x):
    return x.to(torch.float8_e4m3fn).float()


def e4m3_block_scaled(x, block=32):
    """Round to e4m3 after scaling each (column, block of `block` rows) by the power of two that puts its largest
    magnitude just below 448 (E8M0 scales along the reduction axis)."""
    M, F = x.shape
    pad = (-M) % block
    xp = torch.cat([x, x.new_zeros(pad, F)]) if pad else x
    xb = xp.reshape(-1, block, F)
    amax = xb.abs().amax(dim=1, keepdim=True)
    scale = torch.exp2(torch.floor(torch.log2(448.0 / amax.clamp_min(1e-30)))).clamp(max=2.0 ** 100)
    scale = torch.where(amax > 0, scale, torch.ones_like(scale))
    q = (xb * scale).to(torch.float8_e4m3fn).float() / scale
    return q.reshape(-1, F)[:M]


def e2m3_block_scaled(x, block=32):
    """Round to OCP e2m3 (fp6: 1.0 .. 7.5 in three binades of eight steps, multiples of 0.125 below 1) after scaling
    each (column, block of `block` rows) by the power of two that puts its largest magnitude into [4, 8) -- the MX
    convention; magnitudes between 7.5 and 8 saturate at 7.5."""
    M, F = x.shape
    pad = (-M) % block
    xp = torch.cat([x, x.new_zeros(pad, F)]) if pad else x
    xb = xp.reshape(-1, block, F)
    amax = xb.abs().amax(dim=1, keepdim=True)
    scale = torch.exp2(2.0 - torch.floor(torch.log2(amax.clamp_min(1e-30))))
    scale = torch.where(amax > 0, scale, torch.ones_like(scale))
    v = xb * scale
    mag = v.abs()
    step = torch.exp2(torch.floor(torch.log2(mag.clamp_min(1.0))) - 3.0)        # 0.125 below 2, 0.25 below 4, 0.5 above
    q = (torch.round(mag / step) * step).clamp(max=7.5) * torch.sign(v) / scale
    return q.reshape(-1, F)[:M]


def fro(a, b):
    return float((a - b).norm() / b.norm())


def main():
    from nerf_amd import mlp_bf16
    from nerf_amd.model import MipNeRF
    from nerf_amd.stepper import stepper_for
    from nerf_amd.train import grad_bucket
    from nerf_amd.utils import Rays
    fx = load_golden('nerf_e2e_c3_1k')
    cfg = cfg_from(fx)
    cfg.mlp_precision = 'bf16'
    cfg.wgrad_operands = 'bf16'          # the study decodes 16-bit copies
    cfg.level_samples = [int(v) for v in fx['levels']]
    m = MipNeRF(cfg).cuda().train()
    m.load_state_dict(O.init_params(cfg, seed=int(fx['wseed'])), strict=False)
    rays = Rays(*(T(fx['ray_' + k]).cuda() for k in ('origins', 'directions', 'viewdirs', 'radii', 'lossmult', 'near', 'far')))
    n = rays.origins.shape[0]
    grad_bucket(m).zero()
    st = stepper_for(m, n)
    st.step(rays, T(fx['target']).cuda(), t_rand=T(fx['t_rand']).cuda(), u_rand={1: T(fx['u_rand_1']).cuda()})
    torch.cuda.synchronize()
    pk = mlp_bf16.packed_for(m)
    M = 64 * st.nblk_total
    H, D = pk.H, pk.D
    X = {i: tfrag_dense(st.acts[i], 32 * pk.HT, M) for i in range(8)}
    E = tfrag_dense(st.acts[20], 32 * pk.DT, M)[:, :D]
    DY = {i: tfrag_dense(st.dys[i], 32 * pk.HT, M) for i in range(8)}
    names = mlp_bf16.STREAMED[:8]
    ins = [E, X[0], X[1], X[2], torch.cat([X[3], E], 1), X[4], X[5], X[6]]
    print('%-16s %9s | %9s %9s %9s | vs bf16-operand dW: %8s %8s %8s %8s %8s %8s' % (
        'layer', 'kernel', 'ref:bf16', 'ref:X8', 'ref:X8dY8', 'X e4m3', 'dY mx8', 'both', 'dY mx6', 'X8 dY6', 'X6 dY6'))
    for i, nm in enumerate(names):
        dy, x = DY[i], ins[i]
        base = dy.t() @ x
        kern = m.get_parameter(nm + '.weight').grad
        vx = dy.t() @ e4m3(x)
        vy = e4m3_block_scaled(dy).t() @ x
        vb = e4m3_block_scaled(dy).t() @ e4m3(x)
        ref = T(fx['gsample_' + nm + '.weight']).cuda()
        samp = lambda g: g.reshape(-1)[::max(1, g.numel() // 4096)][:4096]
        y6 = e2m3_block_scaled(dy)
        print('%-16s %9.2e | %9.4f %9.4f %9.4f | %27.4f %8.4f %8.4f %8.4f %8.4f %8.4f' % (
            nm, fro(kern, base), fro(samp(base), ref), fro(samp(vx), ref), fro(samp(vb), ref),
            fro(vx, base), fro(vy, base), fro(vb, base), fro(y6.t() @ x, base), fro(y6.t() @ e4m3(x), base),
            fro(y6.t() @ e2m3_block_scaled(x), base)))


if __name__ == '__main__':
    main()
