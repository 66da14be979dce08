"""The density/colour MLP on HIP kernels.

Two execution paths share one parameter layout (the reference's nn.Linear tensors,
`[out, in]` weights, model.py:54-93 / 258-302):

* `precision='fp32'` -- layer-by-layer on the fp32-input matrix cores (`nerf_gemm_f32`),
  exact fp32 accumulation; parity path (1e-4) and the path for arbitrary layer widths.
* `precision='bf16'` -- the fused register-chained bf16 MFMA kernels of `mlp_bf16.hip`
  (see `nerf_amd.mlp_bf16`); throughput path (2e-2).

Everything here runs through the C ABI; there is no torch.matmul / F.linear fallback.
"""
import torch

from ._lib import check, f32c, lib, ptr, stream

ACT = {None: 0, 'none': 0, 'relu': 1, 'sigmoid': 2}


def _split_k(m_out, n_out, k):
    tiles = ((m_out + 127) // 128) * ((n_out + 127) // 128)
    want = max(1, 1024 // tiles)
    return int(max(1, min(want, (k + 2047) // 2048)))


def gemm(a, a_strides, b, b_strides, m, n, k, bias=None, act=0, split_k=1, out=None):
    """out[m,n] = act(A[m,k] B[k,n] + bias); operands addressed by element strides."""
    if out is None:
        out = (torch.zeros if split_k > 1 else torch.empty)(m, n, device=a.device, dtype=torch.float32)
    check(lib().nerf_gemm_f32(ptr(a), a_strides[0], a_strides[1], ptr(b), b_strides[0], b_strides[1], ptr(bias),
                              ptr(out), out.stride(0), m, n, k, act, split_k, stream()))
    return out


def matmul_small(a, a_strides, b, b_strides, m, n, k, bias=None, out=None, accumulate=False):
    """out[m,n] (+)= A[m,k] B[k,n] (+ bias) for small products, in a fixed summation order (no atomics)."""
    if out is None:
        out = torch.empty(m, n, device=a.device, dtype=torch.float32)
    check(lib().nerf_matmul_small_f32(ptr(a), a_strides[0], a_strides[1], ptr(b), b_strides[0], b_strides[1], ptr(bias),
                                      ptr(out), out.stride(0), m, n, k, int(bool(accumulate)), stream()))
    return out


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) with W [out,in]; backward = act', dX = dY W, dW = dY^T X, db = colsum(dY)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x, weight = f32c(x), f32c(weight)
        bias = f32c(bias) if bias is not None else None
        m, k = x.shape
        n = weight.shape[0]
        y = gemm(x, (k, 1), weight, (1, k), m, n, k, bias, act)
        ctx.save_for_backward(x, weight, y if act else None)
        ctx.act = act
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        m, k = x.shape
        n = weight.shape[0]
        g = gy.detach().to(torch.float32).contiguous()
        if ctx.act:
            g = g.clone() if g.data_ptr() == gy.data_ptr() else g
            check(lib().nerf_act_bwd_f32(ptr(g), n, ptr(y), n, m, n, ctx.act, stream()))
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = gemm(g, (n, 1), weight, (k, 1), m, k, n)
        if ctx.needs_input_grad[1]:
            gw = gemm(g, (1, n), x, (k, 1), n, k, m, split_k=_split_k(n, k, m))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = torch.zeros(n, device=g.device, dtype=torch.float32)
            check(lib().nerf_colsum_f32(ptr(g), n, m, n, ptr(gb), stream()))
        return gx, gw, gb, None


def linear(x, weight, bias=None, act=None):
    return _Linear.apply(x, weight, bias, ACT[act])


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x, gamma, beta = f32c(x), f32c(gamma), f32c(beta)
        m, h = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(m, device=x.device)
        rstd = torch.empty(m, device=x.device)
        check(lib().nerf_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), m, h, float(eps), ptr(y), ptr(mean), ptr(rstd),
                                       stream()))
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, mean, rstd = ctx.saved_tensors
        m, h = x.shape
        gy = f32c(gy)
        gx = torch.empty_like(x)
        gg = torch.zeros(h, device=x.device)
        gb = torch.zeros(h, device=x.device)
        check(lib().nerf_layernorm_bwd(ptr(x), ptr(gamma), ptr(gy), ptr(mean), ptr(rstd), m, h, ptr(gx), ptr(gg),
                                       ptr(gb), stream()))
        return gx, gg, gb, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, eps)


def _wb(p, name):
    return p[name + '.weight'], p[name + '.bias']


def mlp_forward_fp32(p, enc, view_rows, sar=False):
    """Reference layer order (model.py:117-132 / 343-359) on the fp32 kernels.
    p: name -> tensor (state_dict naming).  enc [M,D], view_rows [M,V].
    -> raw_density [M,1], raw_rgb [M,3] (post-sigmoid) [, angle [M,2]]"""
    h = enc
    for i in (0, 2, 4, 6):
        h = linear(h, *_wb(p, 'density_net0.%d' % i), act='relu')
    h = torch.cat((h, enc), -1)
    for i in (0, 2, 4, 6):
        h = linear(h, *_wb(p, 'density_net1.%d' % i), act='relu')
    if sar:
        hn = layer_norm(h, p['final_density.0.weight'], p['final_density.0.bias'])
        raw_density = linear(hn, *_wb(p, 'final_density.1'))
    else:
        raw_density = linear(h, *_wb(p, 'final_density.0'))
    if 'rgb_net0.0.weight' not in p:            # use_viewdirs=False: heads straight on the trunk (model.py:78-92,123,283-302)
        g = h
    else:
        g = linear(h, *_wb(p, 'rgb_net0.0'))
        g = linear(torch.cat((g, view_rows), -1), *_wb(p, 'rgb_net1.0'), act='relu')
    raw_rgb = linear(g, *_wb(p, 'final_rgb.0'), act='sigmoid')
    if sar:
        return raw_density, raw_rgb, linear(g, *_wb(p, 'final_angle.0'), act='sigmoid')
    return raw_density, raw_rgb
