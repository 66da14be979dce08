"""Python side of the fused bf16 MFMA MLP (`csrc/mlp_bf16.hip`): weight packing, buffer
management and the autograd hook-up of the forward / backward / weight-gradient kernels.

Data never takes a detour through torch math here: torch only allocates device buffers and owns
the fp32 master parameters (the reference's nn.Linear tensors, model.py:54-93)."""
import ctypes
from ctypes import POINTER, c_float, c_int, c_int64, c_void_p

import torch

from . import _lib, profiling
from ._lib import MlpDims, check, f32c, lib, ptr, stream
from .mlp import matmul_small

_P = c_void_p
_lib.register('nerf_mlp_sizes', c_int, [POINTER(MlpDims), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)])
_lib.register('nerf_mlp_pack', c_int, [POINTER(MlpDims), POINTER(c_void_p), _P, _P, _P])
_lib.register('nerf_encode_frag', c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, _P, _P])
_lib.register('nerf_view_frag', c_int, [_P, c_int64, c_int, c_int, _P, _P])
_lib.register('nerf_mlp_fwd', c_int, [POINTER(MlpDims), _P, _P, _P, _P, c_int64, c_int, c_int64, POINTER(c_void_p),
                                      c_int, _P, _P, _P, _P, _P])

_lib.register('nerf_mlp_fwd_composite', c_int, [POINTER(MlpDims), _P, _P, _P, _P, c_int64, c_int, _P, _P, c_int,
                                                POINTER(_lib.NerfActivation), _P, _P, _P, _P, _P])
_lib.register('nerf_mlp_bwd', c_int, [POINTER(MlpDims), _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, POINTER(c_void_p),
                                      POINTER(c_void_p), c_int, _P])
class WgradDesc(ctypes.Structure):
    """nerf_wgrad_desc of include/nerf_amd.h."""
    _fields_ = [('dy', c_void_p * 2), ('dy_feat', c_int * 2), ('dy_true', c_int * 2),
                ('x', c_void_p * 2), ('x_feat', c_int * 2), ('x_true', c_int * 2),
                ('G', (c_void_p * 2) * 2), ('ld', (c_int * 2) * 2), ('col0', (c_int * 2) * 2), ('gb', c_void_p * 2),
                ('tfrag_fmt', c_int), ('dy_scale', c_void_p * 2), ('workspace', c_void_p), ('workspace_bytes', c_int64)]


_lib.register('nerf_mlp_wgrad_multi', c_int, [POINTER(WgradDesc), c_int64, c_int, _P])
_lib.register('nerf_mlp_wgrad_batch', c_int, [POINTER(WgradDesc), c_int, c_int64, _P])
_lib.register('nerf_mlp_wgrad_config', c_int, [c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)])
_lib.register('nerf_mlp_wgrad8_config', c_int, [c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)])
_lib.register('nerf_mlp_wgrad', c_int, [_P, c_int, _P, c_int, c_int64, _P, c_int, c_int, c_int, c_int, _P, c_int, _P])

STREAMED = (['density_net0.%d' % i for i in (0, 2, 4, 6)] + ['density_net1.%d' % i for i in (0, 2, 4, 6)]
            + ['rgb_net0.0', 'rgb_net1.0'])
RAY_SHAPES = {'cone': 0, 'cylinder': 1}
# Storage format of the weight-gradient operands the chain kernels save (NERF_TFRAG_* of include/nerf_amd.h).
# `config.wgrad_operands` selects it per model: 'bf16' (16-bit copies) or 'mx' (activations e4m3, gradients block-scaled
# e2m3; 'e4m3' is accepted as the older name of the same setting); this is the default.
TFRAG_FORMATS = {'bf16': 0, 'mx': 1, 'e4m3': 1}
DEFAULT_WGRAD_OPERANDS = 'mx'




def _raw(t):
    """Device pointer of any contiguous tensor (byte buffers, bf16 fragments)."""
    assert t.is_cuda and t.is_contiguous()
    return c_void_p(t.data_ptr())


class Geom:
    """Layer sizes of one network and the format of its saved weight-gradient operands: everything the buffer sizes
    depend on.  `ints()` / `from_ints` carry it through a custom op's integer-list argument."""

    def __init__(self, D, V, H, W, sar, fmt):
        self.D, self.V, self.H, self.W, self.sar, self.fmt = int(D), int(V), int(H), int(W), bool(sar), int(fmt)
        self.view = self.V > 0                      # without view directions (model.py:78-92, 283-302) no colour layer
        self.HT, self.WT = (self.H + 31) // 32, (self.W + 31) // 32
        self.DKS, self.VKS = (self.D + 15) // 16, (self.V + 15) // 16
        self.DT, self.VT = (self.DKS + 1) // 2, (self.VKS + 1) // 2

    @property
    def dims(self):
        """nerf_mlp_dims of the C ABI (built on use: a compiler tracing through this class never meets ctypes)."""
        return MlpDims(self.D, self.V, self.H, self.W, int(self.sar))

    @staticmethod
    def of_model(model):
        cfg = model.config
        view = hasattr(model, 'rgb_net0')
        return Geom(6 * (cfg.max_deg - cfg.min_deg), 3 + 6 * (cfg.viewdirs_max_deg - cfg.viewdirs_min_deg) if view else 0,
                    cfg.hidden, cfg.num_samples if view else 0, hasattr(model, 'final_angle'),
                    TFRAG_FORMATS[getattr(cfg, 'wgrad_operands', None) or DEFAULT_WGRAD_OPERANDS])

    def ints(self):
        return [self.D, self.V, self.H, self.W, int(self.sar), self.fmt]

    @staticmethod
    def from_ints(v):
        return Geom(*v)

    def param_names(self):
        """The network's parameters in the order the custom ops take and return them."""
        names = SAR_PARAM_NAMES if self.sar else PARAM_NAMES
        return [k for k in names if self.view or not k.startswith(('rgb_net0', 'rgb_net1'))]


class Packed(Geom):
    """bf16 weight streams + fp32 constants of one network, valid until the next optimizer step."""

    def __init__(self, model, need_bwd=True, params=None, geom=None):
        """From a model, or from `params` (name -> tensor) and `geom`."""
        if geom is None:
            geom = Geom.of_model(model)
        Geom.__init__(self, *geom.ints())
        p = params if params is not None else {k: v.detach() for k, v in model.named_parameters()}
        dev = p['density_net0.0.weight'].device
        assert self.sar == ('final_angle.0.weight' in p) and self.view == ('rgb_net0.0.weight' in p)
        H, W, D, V = self.H, self.W, self.D, self.V
        HP, WP = 32 * self.HT, 32 * self.WT
        nf, nb, nc = c_int64(), c_int64(), c_int64()
        check(lib().nerf_mlp_sizes(ctypes.byref(self.dims), ctypes.byref(nf), ctypes.byref(nb), ctypes.byref(nc)))
        self.wfwd = torch.empty(nf.value, dtype=torch.uint8, device=dev)
        self.wbwd = torch.empty(nb.value, dtype=torch.uint8, device=dev) if need_bwd else None

        # density head row and stacked colour heads (fp32 [out,in] like every streamed matrix)
        if self.sar:
            gamma, beta = p['final_density.0.weight'], p['final_density.0.bias']
            wd, bd = p['final_density.1.weight'], p['final_density.1.bias']
            self.head_density = (wd * gamma[None, :]).contiguous()
            self.head_colour = torch.cat([p['final_rgb.0.weight'], p['final_angle.0.weight']], 0).contiguous()
            head_bias = torch.cat([p['final_rgb.0.bias'], p['final_angle.0.bias']])
        else:
            wd, bd = p['final_density.0.weight'], p['final_density.0.bias']
            self.head_density = wd.contiguous()
            self.head_colour = p['final_rgb.0.weight'].contiguous()
            head_bias = p['final_rgb.0.bias']
        if self.view:
            # rgb_net0 (no activation, model.py:82-84) folded into rgb_net1: one matrix on [trunk output | view features]
            self.W8, self.b8 = f32c(p['rgb_net0.0.weight']), f32c(p['rgb_net0.0.bias'])
            self.W9 = f32c(p['rgb_net1.0.weight'])
            fold = self.W9.clone()
            # (a fixed summation order: atomic split-K sums would make the bf16 rounding of the folded matrix vary from
            # one packing of the same weights to the next)
            matmul_small(self.W9, (H + V, 1), self.W8, (H, 1), W, H, H, out=fold)                    # W9[:, :H] @ W8
            bias9 = matmul_small(self.b8, (H, 1), self.W9, (1, H + V), 1, W, H, bias=f32c(p['rgb_net1.0.bias']))[0]  # b8 W9a^T + b9
            tail = [fold, f32c(self.head_density), f32c(self.head_colour)]
        else:
            # one head tile on the trunk output: density row, then the colour (and angle) heads
            tail = [None, torch.cat([self.head_density, self.head_colour], 0).contiguous(), None]
        mats = [f32c(p[n + '.weight']) for n in STREAMED[:8]] + tail
        self._keep = mats
        arr = (c_void_p * 11)(*[m.data_ptr() if m is not None else None for m in mats])
        check(lib().nerf_mlp_pack(ctypes.byref(self.dims), arr, _raw(self.wfwd),
                                  _raw(self.wbwd) if need_bwd else None, stream()))

        # fp32 constants block (layout: Net::C_* in mlp_bf16.hip): the pieces are concatenated once and gathered into
        # place through an index map that is fixed per geometry (two launches where one slice assignment per piece
        # was a dozen to three dozen small copies per packing)
        pieces = [p[n + '.bias'] for n in STREAMED[:8]]                 # -> i * HP .. i * HP + H
        if self.view:
            pieces.append(bias9)                                        # -> 8 HP .. + W
        if not self.sar:
            pieces.append(bd.reshape(1))                                # -> row 0 of the density head tile
        pieces.append(head_bias)                                        # -> rows 1.. of the head tile / the colour head tile
        if self.sar:
            pieces += [self.head_density.sum().reshape(1), ((wd[0] * beta).sum() + bd[0]).reshape(1)]
        idx, fixed = _consts_map(self, nc.value, dev, head_bias.numel())
        c = torch.cat([f32c(x).reshape(-1) for x in pieces] + [fixed]).index_select(0, idx)
        self.consts = c
        # algorithmic work per sample (true layer sizes): forward MACs, input-gradient MACs (no gradient
        # w.r.t. the encodings), weight-gradient MACs; bytes = bf16 operands the weight-gradient pass must read
        heads = 1 + (5 if self.sar else 3)
        if self.view:
            macs_fwd = D * H + 3 * H * H + (H + D) * H + 3 * H * H + H + H * H + (H + V) * W + (heads - 1) * W
        else:
            macs_fwd = D * H + 3 * H * H + (H + D) * H + 3 * H * H + heads * H
        macs_dx = macs_fwd - D * H - D * H - V * W
        self.fwd_flops = 2.0 * macs_fwd
        self.bwd_dx_flops = 2.0 * macs_dx
        self.wgrad_flops = 2.0 * macs_fwd
        # what the chain kernels issue for the true layer sizes: rgb_net0 is folded into rgb_net1, one H x H product
        # less per pass (transposition products and padding are overhead, not counted)
        fold = H * H if self.view else 0
        self.fwd_flops_exec = 2.0 * (macs_fwd - fold)
        self.bwd_dx_flops_exec = 2.0 * (macs_dx - fold)
        # bf16 operand bytes per sample that the weight-gradient launches read (see _wgrad_plan)
        self.wgrad_bytes = sum(_wgrad_launch_bytes(self, dys, xs) for dys, xs, _ in _wgrad_plan(self))


_CONSTS_MAP = {}


def _consts_map(g, n_consts, dev, n_head_bias):
    """Where each float of the constants block comes from in the concatenation Packed builds: an int64 index per float
    (device tensor, cached per geometry) and the three fixed values [0, 1/H, LayerNorm eps] that end the concatenation."""
    key = (tuple(g.ints()), str(dev))
    if key not in _CONSTS_MAP:
        H, W, HP, WP = g.H, g.W, 32 * g.HT, 32 * g.WT
        src = 0
        pos = {}                                    # destination -> source
        for i in range(8):
            pos.update({i * HP + k: src + k for k in range(H)})
            src += H
        o = 8 * HP
        if g.view:
            pos.update({o + k: src + k for k in range(W)})
            src += W
        o += WP
        if not g.sar:
            pos[o] = src
            src += 1
        ob = o + 1 if not g.view else o + 32       # head biases: behind the density row, or the colour head tile
        pos.update({ob + k: src + k for k in range(n_head_bias)})
        src += n_head_bias
        o += 64
        if g.sar:
            pos[o + 6], pos[o + 7] = src, src + 1
            src += 2
            pos[o + 8], pos[o + 9] = src + 1, src + 2
        idx = torch.full((n_consts,), src, dtype=torch.int64)          # default: the zero
        for dst, s_ in pos.items():
            idx[dst] = s_
        _CONSTS_MAP[key] = (idx.to(dev), torch.tensor([0.0, 1.0 / H, 1e-5], dtype=torch.float32).to(dev))
    return _CONSTS_MAP[key]


def _wgrad_plan(pk):
    """The weight-gradient launches of one backward pass.  Each launch: (dY segments, X segments, blocks) with
    a dY segment = (dY index, padded features, true features, parameter owning the bias or None),
    an X segment = (source, padded features, true features), and blocks[i][j] = (parameter, column offset) or None.
    X sources: ('x', i) saved activation i, 'enc', 'view'.  dY 10 / 11 are the colour / density head tiles.
    Buffers shared by two gradients ride in one launch, so every operand is read once per pass."""
    H, W, D, V = pk.H, pk.W, pk.D, pk.V
    HP, WP, DP, VP = 32 * pk.HT, 32 * pk.WT, 32 * pk.DT, 32 * pk.VT
    one = lambda di, nf, nt, name, xs, kf, kt: ([(di, nf, nt, name)], [(xs, kf, kt)], [[(name, 0)]])
    plan = [one(0, HP, H, 'density_net0.0', 'enc', DP, D)]
    plan += [one(l, HP, H, 'density_net0.%d' % i, ('x', l - 1), HP, H) for l, i in ((1, 2), (2, 4), (3, 6))]
    # density_net1.0 takes [X3 | encoding]
    if pk.fmt and pk.HT + pk.DT > 10:
        # (max_deg 16: eleven X tiles are more than a 4-wave workgroup of the 8/6-bit kernel holds accumulators for
        # beside eight dY tiles; the two column blocks take a launch each, the second without the bias sums)
        plan += [one(4, HP, H, 'density_net1.0', ('x', 3), HP, H),
                 ([(4, HP, H, None)], [('enc', DP, D)], [[('density_net1.0', H)]])]
    else:
        plan += [([(4, HP, H, 'density_net1.0')], [(('x', 3), HP, H), ('enc', DP, D)],
                  [[('density_net1.0', 0), ('density_net1.0', H)]])]
    plan += [one(l, HP, H, 'density_net1.%d' % i, ('x', l - 1), HP, H) for l, i in ((5, 2), (6, 4), (7, 6))]
    # the folded colour layer and the density head both read the trunk output X7: [dY9 | density tile] x [X7 | view]
    # in one launch.  dY9^T X7 lands in the scratch pair '_g98' (fold_gradients turns it into the gradients of
    # rgb_net0.0 and of the trunk columns of rgb_net1.0); dY9^T view is the gradient of rgb_net1.0's view columns.
    # (SAR: the density tile carries rows g*rstd, g*rstd*mean, g; they land in the scratch pair '_dens')
    if not pk.view:
        # one head tile on the trunk output, rows: density, rgb 1..3 (SAR: angle 4..5, and the LayerNorm sums
        # g*rstd*mean, g in rows 6..7: only their column sums are used); lands in the scratch pair '_hd'
        return plan + [one(11, 32, 8 if pk.sar else 4, '_hd', ('x', 7), HP, H)]
    dens = ('_dens', 3) if pk.sar else ('final_density.0', 1)
    plan += [([(9, WP, W, '_g98'), (11, 32, dens[1], dens[0])], [(('x', 7), HP, H), ('view', VP, V)],
              [[('_g98', 0), ('rgb_net1.0', H)], [(dens[0], 0), None]])]
    # (SAR: rows 0..2 final_rgb, 3..4 final_angle, in the scratch pair '_heads')
    plan += [one(10, 32, 5, '_heads', ('x', 9), WP, W) if pk.sar else one(10, 32, 3, 'final_rgb.0', ('x', 9), WP, W)]
    return plan


def _wgrad_launch_bytes(pk, dsegs, xsegs):
    """Operand bytes per sample that one weight-gradient launch reads."""
    dfeats, xfeats = sum(s[1] for s in dsegs), sum(s[1] for s in xsegs)
    if pk.fmt == 0:
        return 2.0 * (dfeats + xfeats)
    # e4m3 activations, e2m3 gradients + 256 scale bytes per 32-sample block and dY segment
    return 1.0 * xfeats + 0.75 * dfeats + 8.0 * len(dsegs)


def _param_versions(model):
    # torch bumps a tensor's version on every in-place write (optimizer steps, load_state_dict, .copy_), and
    # data_ptr changes when a parameter is replaced: together they identify the weights the streams were packed from
    return tuple((p.data_ptr(), p._version) for p in model.parameters())


def packed_for(model, need_bwd=True):
    """The packed bf16 streams of `model`, re-packed whenever a parameter has been written since the last packing."""
    pk = getattr(model, '_packed', None)
    versions = _param_versions(model) + (getattr(model.config, 'wgrad_operands', None),)
    if pk is None or pk.versions != versions or (need_bwd and pk.wbwd is None):
        pk = Packed(model, need_bwd)
        pk.versions = versions
        model._packed = pk
    return pk


_PACKED_BY_PARAMS = {}


def packed_for_params(weights, geom, need_bwd=True):
    """The packed streams for a parameter list in `geom.param_names()` order (what the custom ops receive), cached on
    the tensors' storage and version like `packed_for`; a handful of entries at most (one per live network)."""
    key = tuple(geom.ints()) + tuple((w.data_ptr(), w._version) for w in weights)
    pk = _PACKED_BY_PARAMS.get(key)
    if pk is None or (need_bwd and pk.wbwd is None):
        if len(_PACKED_BY_PARAMS) >= 8:
            _PACKED_BY_PARAMS.clear()
        pk = Packed(None, need_bwd, params={k: w.detach() for k, w in zip(geom.param_names(), weights)}, geom=geom)
        _PACKED_BY_PARAMS[key] = pk
    return pk


def n_blocks(n_samples):
    return ((n_samples + 255) // 256) * 4


def encode_frag(t_vals, origins, directions, radii, ray_shape, min_deg, max_deg):
    """(a2)+(a3): fenceposts -> Gaussians -> IPE, as bf16 fragments (uint8 buffer)."""
    t_vals = f32c(t_vals)
    n, s1 = t_vals.shape
    S = s1 - 1
    dks = (6 * (max_deg - min_deg) + 15) // 16
    out = torch.empty(n_blocks(n * S) * dks * 2 * 1024, dtype=torch.uint8, device=t_vals.device)
    check(lib().nerf_encode_frag(ptr(t_vals), ptr(f32c(origins).reshape(n, 3)), ptr(f32c(directions).reshape(n, 3)),
                                 ptr(f32c(radii).reshape(n)), n, S, RAY_SHAPES[ray_shape], min_deg, max_deg,
                                 _raw(out), stream()))
    return out


def view_frag(dirs, min_deg, max_deg):
    dirs = f32c(dirs).reshape(-1, 3)
    vks = (6 * (max_deg - min_deg) + 3 + 15) // 16
    out = torch.empty(dirs.shape[0] * vks * 2 * 16, dtype=torch.uint8, device=dirs.device)
    check(lib().nerf_view_frag(ptr(dirs), dirs.shape[0], min_deg, max_deg, _raw(out), stream()))
    return out


def act_block_bytes(pk):
    """Bytes per block of 64 samples of each buffer the forward chain saves (index as in `acts`; None = unused)."""
    tiles = [pk.HT] * 9 + [pk.WT]
    piece = 4096 >> pk.fmt                          # transposed fragments: bf16 or e4m3
    used = lambda i, t: i != 8 and t > 0            # (no X8 since the fold; no X9 without view directions)
    per = [t * piece if used(i, t) else None for i, t in enumerate(tiles)]
    # ReLU sign words: per block of 32 samples, chunks of four words per lane (ceil(ceil(tiles / 2) / 4) KiB)
    per += [2 * (((t + 1) // 2 + 3) // 4) * 1024 if used(i, t) else None for i, t in enumerate(tiles)]
    per += [pk.DT * piece, pk.VT * piece if pk.view else None]
    if pk.sar:
        per.append(32 * pk.HT * 128)
    return per


def dy_block_bytes(pk):
    """Bytes per block of 64 samples of each dY buffer of the backward chain (None = unused); in the 8/6-bit format 12
    more entries: the scale bytes of each (8 bytes per feature row and 32-sample block)."""
    tiles = [pk.HT] * 9 + [pk.WT, 1 if pk.view else 0, 1]
    piece = 1536 if pk.fmt else 4096                # per tile and 64 samples: e2m3 (2 x 768 B) or bf16
    per = [t * piece if (i != 8 and t > 0) else None for i, t in enumerate(tiles)]
    if pk.fmt:
        per += [512 if b is not None else None for b in per[:12]]
    return per


def mlp_forward(pk, enc, view, n_samples, ray_div, n_rays, save=False, acts=None, raw=None):
    """-> raw_density [M], raw_rgb [M,3], raw_angle [M,2] or None, acts (list of byte buffers) or None.
    `acts` (buffers laid out as act_block_bytes says) and `raw = (raw_density, raw_rgb)` (SAR: `(raw_density,
    raw_rgb, raw_angle, ln_stats)`) may be supplied by a caller that owns persistent buffers; otherwise they are
    allocated here."""
    dev = enc.device
    nblk = n_blocks(n_samples)
    raw_angle = ln_stats = None
    if raw is not None and len(raw) == 4:           # SAR caller with persistent buffers
        raw_density, raw_rgb, raw_angle, ln_stats = raw
    else:
        raw_density, raw_rgb = raw if raw is not None else (torch.empty(n_samples, device=dev), torch.empty(n_samples, 3, device=dev))
        raw_angle = torch.empty(n_samples, 2, device=dev) if pk.sar else None
        ln_stats = torch.empty(n_samples, 2, device=dev) if (pk.sar and save) else None
    arr = None
    if acts is not None:
        arr = (c_void_p * len(acts))(*[a.data_ptr() if a is not None else None for a in acts])
    elif save:
        # 0..9 X0..X7, X9 as T-frags (8 unused: rgb_net0's output is never formed), 10..19 their ReLU sign words,
        # 20 enc, 21 view T-frags
        acts = [torch.empty(nblk * b, dtype=torch.uint8, device=dev) if b is not None else None
                for b in act_block_bytes(pk)]        # (22, SAR only: the trunk output once more in frag order)
        arr = (c_void_p * len(acts))(*[a.data_ptr() if a is not None else None for a in acts])
    saved = sum(a.numel() for a in acts if a is not None) if acts is not None else 0   # bytes the launch writes for the backward pass
    with profiling.timed('mlp_fwd_kernel', flops=pk.fwd_flops * n_samples, bytes_=float(saved),
                         flops_exec=pk.fwd_flops_exec * n_samples):
        check(lib().nerf_mlp_fwd(ctypes.byref(pk.dims), _raw(pk.wfwd), ptr(pk.consts), _raw(enc), _raw(view) if pk.view else None,
                                 n_samples, ray_div, n_rays, arr, pk.fmt, ptr(raw_density), ptr(raw_rgb), ptr(raw_angle),
                                 ptr(ln_stats), stream()))
    return raw_density, raw_rgb, raw_angle, acts, ln_stats


def mlp_backward(pk, acts, n_samples, g_density, g_rgb, raw_rgb, g_angle=None, raw_density=None, raw_angle=None,
                 ln_stats=None, dys=None, n_real=None):
    """Backward chain -> list of 12 dY T-frag buffers (dY0..dY9, colour head tile, density head tile).
    The SAR heads also need the angle gradient and the forward's density / angle outputs and LayerNorm statistics.
    `dys` may be supplied (dy_block_bytes layout); `n_real` is the sample count the work is accounted for."""
    dev = g_density.device
    nblk = n_blocks(n_samples)
    if dys is None:
        dys = [torch.empty(nblk * b, dtype=torch.uint8, device=dev) if b is not None else None for b in dy_block_bytes(pk)]
    a_arr = (c_void_p * len(acts))(*[a.data_ptr() if a is not None else None for a in acts])
    d_arr = (c_void_p * len(dys))(*[d.data_ptr() if d is not None else None for d in dys])
    g_density, g_rgb = f32c(g_density).reshape(-1), f32c(g_rgb).reshape(-1, 3)
    if pk.sar:
        g_angle = f32c(g_angle).reshape(-1, 2)
    with profiling.timed('mlp_bwd_kernel', flops=pk.bwd_dx_flops * (n_real or n_samples),
                         bytes_=float(sum(d.numel() for d in dys if d is not None)),
                         flops_exec=pk.bwd_dx_flops_exec * (n_real or n_samples)):
        check(lib().nerf_mlp_bwd(ctypes.byref(pk.dims), _raw(pk.wbwd), ptr(pk.consts), n_samples, ptr(g_density),
                                 ptr(g_rgb), ptr(g_angle), ptr(raw_density), ptr(raw_rgb), ptr(raw_angle),
                                 ptr(ln_stats), a_arr, d_arr, pk.fmt, stream()))
    return dys


_WGRAD_KERNEL = {}


def wgrad_kernel_name(n_feat, k_feat, fmt=0):
    """Symbol of the kernel instantiation that handles these feature counts (ints, or for the 8/6-bit format pairs:
    the padded features of the launch's two dY segments and of its two X segments)."""
    key = (n_feat, k_feat, fmt)
    if key not in _WGRAD_KERNEL and fmt:
        (n0, n1), (k0, k1) = [(v, 0) if isinstance(v, int) else (tuple(v) + (0,))[:2] for v in (n_feat, k_feat)]
        tn, tk, gs = c_int(), c_int(), c_int()
        check(lib().nerf_mlp_wgrad8_config(n0, n1, k0, k1, ctypes.byref(tn), ctypes.byref(tk), ctypes.byref(gs)))
        _WGRAD_KERNEL[key] = 'mlp_wgrad8_kernel<%d, %d, %d>' % (tn.value, tk.value, gs.value)
    if key not in _WGRAD_KERNEL:
        nw, tn, tk, gt = c_int(), c_int(), c_int(), c_int()
        check(lib().nerf_mlp_wgrad_config(n_feat, k_feat, ctypes.byref(nw), ctypes.byref(tn), ctypes.byref(tk),
                                          ctypes.byref(gt)))
        _WGRAD_KERNEL[key] = 'mlp_wgrad_kernel<%d, %d, %d, %d>' % (nw.value, tn.value, tk.value, gt.value)
    return _WGRAD_KERNEL[key]


def _wgrad(dy, n_feat, x, k_feat, n_samples, G, col0, n_true, k_true, gb):
    """Single-block launch (dW[:, col0:col0+k_true] += dY^T X)."""
    with profiling.timed(wgrad_kernel_name(n_feat, k_feat), flops=2.0 * n_true * k_true * n_samples,
                         bytes_=2.0 * (n_feat + k_feat) * n_samples):
        check(lib().nerf_mlp_wgrad(_raw(dy), n_feat, _raw(x), k_feat, n_samples, ptr(G), G.stride(0), col0, n_true,
                                   k_true, ptr(gb), 0, stream()))


def fold_gradients(pk, grads):
    """Gradients of rgb_net0.0 and of rgb_net1.0's trunk columns from what the folded layer's launch produced:
    G = sum_m dY9 x7^T (grads['_g98.weight'], [W,H]) and s = sum_m dY9 (grads['_g98.bias']).  With
    z9 = W9a (W8 x7 + b8) + W9v view + b9:   dW9a = G W8^T + s b8^T,  db9 = s,  dW8 = W9a^T G,  db8 = W9a^T s.
    Small fp32 products in a fixed summation order; they add into the running gradients."""
    H, W, V = pk.H, pk.W, pk.V
    G, s = grads['_g98.weight'], grads['_g98.bias']
    g9, g8 = grads['rgb_net1.0.weight'], grads['rgb_net0.0.weight']
    matmul_small(G, (H, 1), pk.W8, (1, H), W, H, H, out=g9, accumulate=True)          # += G W8^T   (columns 0..H-1)
    matmul_small(s, (1, 1), pk.b8, (H, 1), W, H, 1, out=g9, accumulate=True)          # += s b8^T
    matmul_small(pk.W9, (1, H + V), G, (H, 1), H, H, W, out=g8, accumulate=True)      # += W9a^T G
    matmul_small(pk.W9, (1, H + V), s, (1, 1), H, 1, W, out=grads['rgb_net0.0.bias'].view(H, 1), accumulate=True)   # += W9a^T s
    grads['rgb_net1.0.bias'].add_(s)


def mlp_weight_grads(pk, acts, dys, n_samples, grads, after_trunk=None, grid_hint=0, workspace=None):
    """Accumulate all weight / bias gradients into `grads` (name -> zero-initialised or running fp32 tensor).
    `after_trunk()` (optional) is called once the launches that write the trunk layers' gradients (density_net0 /
    density_net1: nine tenths of the bytes) are queued and before the remaining ones: a multi-rank caller starts
    their all-reduce there, under the rest of the pass."""
    dev = pk.consts.device
    grads = dict(grads)
    # (the scratch pairs '_hd' / '_g98' may come zeroed from the caller, who then clears all its scratch with one fill)
    if not pk.view:
        hd_w = grads['_hd.weight'] if '_hd.weight' in grads else torch.zeros(8, pk.H, device=dev)
        hd_b = grads['_hd.bias'] if '_hd.bias' in grads else torch.zeros(8, device=dev)
        grads['_hd.weight'], grads['_hd.bias'] = hd_w, hd_b
        _wgrad_launches(pk, acts, dys, n_samples, grads, after_trunk, grid_hint, workspace)
        if pk.sar:      # (the caller assembles the LayerNorm / density gradients from these, see _MlpSar)
            grads['_dens.weight'][0].add_(hd_w[0])
            grads['_dens.bias'][1:3].add_(hd_b[6:8])
            grads['_heads.weight'].add_(hd_w[1:6])
            grads['_heads.bias'].add_(hd_b[1:6])
        else:
            grads['final_density.0.weight'].add_(hd_w[0:1])
            grads['final_density.0.bias'].add_(hd_b[0:1])
            grads['final_rgb.0.weight'].add_(hd_w[1:4])
            grads['final_rgb.0.bias'].add_(hd_b[1:4])
        return
    if '_g98.weight' not in grads:
        grads['_g98.weight'], grads['_g98.bias'] = torch.zeros(pk.W, pk.H, device=dev), torch.zeros(pk.W, device=dev)
    _wgrad_launches(pk, acts, dys, n_samples, grads, after_trunk, grid_hint, workspace)
    fold_gradients(pk, grads)


_WGRAD_WORKSPACE = {}
WGRAD_WORKSPACE_BYTES = 128 << 20


def _wgrad_workspace(dev):
    """Scratch for the two-stage flush of the weight-gradient launches (one buffer per device, reused by every
    launch: launches on a stream run one after another)."""
    ws = _WGRAD_WORKSPACE.get(dev)
    if ws is None:
        ws = _WGRAD_WORKSPACE[dev] = torch.empty(WGRAD_WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
    return ws


WGRAD_BATCHED = True        # launches of one kernel instantiation go out as one launch (nerf_mlp_wgrad_batch)


def _wgrad_launches(pk, acts, dys, n_samples, grads, after_trunk=None, grid_hint=0, workspace=None):
    """The weight-gradient launches of one backward pass (see _wgrad_plan), grouped by kernel instantiation: the
    descriptors of a group go to the library in one call, which runs them as one launch.  The groups that write
    only trunk gradients go first; `after_trunk()` is called between them and the rest.  `grid_hint` > 0: that many
    workgroups per launch (a pass that shares the chip with another stream), each descriptor a launch of its own;
    `workspace`: the scratch of the two-stage flush (passes that run side by side need one each)."""
    ws = workspace if workspace is not None else _wgrad_workspace(pk.consts.device)
    groups = {}
    for dsegs, xsegs, blocks in _wgrad_plan(pk):
        trunk = all(blk is None or blk[0].startswith('density_net') for row in blocks for blk in row)
        d = WgradDesc()
        d.tfrag_fmt = pk.fmt
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        flops = flops_exec = 0.0
        for i, (di, nf, nt, name) in enumerate(dsegs):
            d.dy[i], d.dy_feat[i], d.dy_true[i] = dys[di].data_ptr(), nf, nt
            if pk.fmt:
                d.dy_scale[i] = dys[12 + di].data_ptr()
            if name is not None:
                d.gb[i] = grads[name + '.bias'].data_ptr()
        for j, (xs, kf, kt) in enumerate(xsegs):
            x = acts[20] if xs == 'enc' else acts[21] if xs == 'view' else acts[xs[1]]
            d.x[j], d.x_feat[j], d.x_true[j] = x.data_ptr(), kf, kt
        for i, row in enumerate(blocks):
            for j, blk in enumerate(row):
                if blk is not None:
                    w = grads[blk[0] + '.weight']
                    d.G[i][j], d.ld[i][j], d.col0[i][j] = w.data_ptr(), w.stride(0), blk[1]
                    flops += 2.0 * dsegs[i][2] * xsegs[j][2] * n_samples
                    flops_exec += 2.0 * dsegs[i][2] * xsegs[j][2] * n_samples
                    if blk[0] == '_g98':        # this block stands for rgb_net0's weight gradient as well
                        flops += 2.0 * pk.H * pk.H * n_samples
        nf_all, kf_all = sum(s[1] for s in dsegs), sum(s[1] for s in xsegs)
        if pk.fmt:      # (the instantiation depends on the split into segments)
            nf_all, kf_all = tuple(s[1] for s in dsegs), tuple(s[1] for s in xsegs)
        # algorithmic work of the launch: every operand byte once, 2 FLOP per true multiply-add
        key = (trunk, wgrad_kernel_name(nf_all, kf_all, pk.fmt), nf_all, kf_all) if WGRAD_BATCHED else len(groups)
        g = groups.setdefault(key, dict(name=wgrad_kernel_name(nf_all, kf_all, pk.fmt), trunk=trunk,
                                        descs=[], flops=0.0, flops_exec=0.0, bytes=0.0))
        g['descs'].append(d)
        g['flops'] += flops
        g['flops_exec'] += flops_exec
        g['bytes'] += _wgrad_launch_bytes(pk, dsegs, xsegs) * n_samples
    signalled = after_trunk is None
    for g in sorted(groups.values(), key=lambda g: not g['trunk']):
        if not g['trunk'] and not signalled:
            after_trunk()
            signalled = True
        for lo in range(0, len(g['descs']), 8):             # (a launch carries at most eight descriptors)
            part = g['descs'][lo:lo + 8]
            arr = (WgradDesc * len(part))(*part)
            share = len(part) / len(g['descs'])
            with profiling.timed(g['name'], flops=g['flops'] * share, flops_exec=g['flops_exec'] * share, bytes_=g['bytes'] * share):
                if grid_hint > 0:
                    for d in part:
                        check(lib().nerf_mlp_wgrad_multi(ctypes.byref(d), n_samples, int(grid_hint), stream()))
                else:
                    check(lib().nerf_mlp_wgrad_batch(arr, len(part), n_samples, stream()))
    if not signalled:
        after_trunk()


PARAM_NAMES = [n + s for n in STREAMED + ['final_density.0', 'final_rgb.0'] for s in ('.weight', '.bias')]


def _present(names, params):
    """The names the model has (without view directions there is no rgb_net0 / rgb_net1, model.py:79-89)."""
    return [k for k in names if k in params]

_ACCUMULATE_INTO_GRAD = False


class accumulate_into_grad:
    """Within this context the NeRF backward adds the weight gradients straight into each parameter's existing
    `.grad` (the weight-gradient kernels accumulate with float atomics anyway) and hands autograd nothing for
    them: no zero-filled temporaries, no `grad += temporary` per parameter and level.  Only for loops that own
    their `.grad` buffers and read gradients from there (`train.nerf_train_step`); `torch.autograd.grad` and
    hooks on the parameters would see no gradient, so it is opt-in."""

    def __enter__(self):
        global _ACCUMULATE_INTO_GRAD
        self.prev, _ACCUMULATE_INTO_GRAD = _ACCUMULATE_INTO_GRAD, True

    def __exit__(self, *exc):
        global _ACCUMULATE_INTO_GRAD
        _ACCUMULATE_INTO_GRAD = self.prev


class _MlpNerf(torch.autograd.Function):
    """raw_density [M,1], raw_rgb [M,3] = MLP(enc frags, view frags); gradients flow to the 24 parameters."""

    @staticmethod
    def forward(ctx, model, encf, viewf, n_samples, ray_div, n_rays, *params):
        need_grad = any(ctx.needs_input_grad[6:])
        pk = packed_for(model, need_bwd=need_grad)
        d, rgb, _, acts, _ = mlp_forward(pk, encf, viewf, n_samples, ray_div, n_rays, save=need_grad)
        ctx.pk, ctx.acts, ctx.n = pk, acts, n_samples
        ctx.shapes = [(p.shape, p.device) for p in params]
        ctx.params = params
        ctx.names = _present(PARAM_NAMES, dict(model.named_parameters()))
        ctx.save_for_backward(rgb)
        return d.reshape(-1, 1), rgb

    @staticmethod
    def backward(ctx, g_density, g_rgb):
        (rgb,) = ctx.saved_tensors
        pk = ctx.pk
        dys = mlp_backward(pk, ctx.acts, ctx.n, g_density, g_rgb, rgb)
        in_place = _ACCUMULATE_INTO_GRAD and all(
            q.grad is not None and q.grad.dtype == torch.float32 and q.grad.is_contiguous() for q in ctx.params)
        names = ctx.names
        if in_place:
            grads = {n: q.grad for n, q in zip(names, ctx.params)}
        else:
            grads = {n: torch.zeros(shape, device=dev, dtype=torch.float32) for n, (shape, dev) in zip(names, ctx.shapes)}
        mlp_weight_grads(pk, ctx.acts, dys, ctx.n, grads)
        ctx.acts = ctx.params = None
        if in_place:
            return (None,) * (6 + len(names))
        return (None, None, None, None, None, None) + tuple(grads[n] for n in names)


def query_nerf(model, t_vals, rays, viewf, S):
    """One level of the NeRF MLP on the bf16 path: fenceposts in, raw density / colour out."""
    cfg = model.config
    n = t_vals.shape[0]
    encf = encode_frag(t_vals, rays.origins, rays.directions, rays.radii, model.ray_shape, cfg.min_deg, cfg.max_deg)
    p = dict(model.named_parameters())
    params = [p[k] for k in _present(PARAM_NAMES, p)]
    if not (torch.is_grad_enabled() and any(q.requires_grad for q in params)):      # inference: nothing saved
        d, rgb, _, _, _ = mlp_forward(packed_for(model, need_bwd=False), encf, viewf, n * S, S, n, save=False)
        return d.reshape(-1, 1), rgb
    return _MlpNerf.apply(model, encf, viewf, n * S, S, n, *params)


def can_render_level(model, S):
    """The fused inference launch (forward chain + compositing) exists for the NeRF heads when the samples of whole
    rays fill a 256-sample workgroup tile and nothing needs the raw outputs."""
    return (256 % S == 0 and not torch.is_grad_enabled() and not getattr(model, 'return_raw', False)
            and not (model.randomized and model.density_noise))


def render_level(model, t_vals, rays, viewf, S, need_weights=True):
    """One level of model.py:113-141 in one launch: fenceposts in, composited colour / distance / acc (and the
    sample weights the next level resamples from) out.  Inference only (see can_render_level)."""
    cfg = model.config
    n = t_vals.shape[0]
    dev = t_vals.device
    encf = encode_frag(t_vals, rays.origins, rays.directions, rays.radii, model.ray_shape, cfg.min_deg, cfg.max_deg)
    pk = packed_for(model, need_bwd=False)
    comp, dist, acc = torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
    weights = torch.empty(n, S, device=dev) if need_weights else None
    act = _lib.NerfActivation(1, float(model.rgb_padding), float(model.density_bias))
    with profiling.timed('mlp_fwd_kernel', flops=pk.fwd_flops * n * S, bytes_=0.0, flops_exec=pk.fwd_flops_exec * n * S):
        check(lib().nerf_mlp_fwd_composite(ctypes.byref(pk.dims), _raw(pk.wfwd), ptr(pk.consts), _raw(encf),
                                           _raw(viewf) if pk.view else None, n, S, ptr(f32c(t_vals)),
                                           ptr(f32c(rays.directions).reshape(n, 3)), int(bool(model.white_bkgd)),
                                           ctypes.byref(act), ptr(comp), ptr(dist), ptr(acc), ptr(weights), stream()))
    return comp, dist, acc, weights


SAR_PARAM_NAMES = [n + s for n in STREAMED + ['final_density.0', 'final_density.1', 'final_rgb.0', 'final_angle.0']
                   for s in ('.weight', '.bias')]


class _MlpSar(torch.autograd.Function):
    """raw_density [M,1], raw_rgb [M,3], angle [M,2] = SAR MLP(enc frags, view frags) (model.py:343-359);
    gradients flow to the 28 parameters, the LayerNorm pair and the density weight through the rows the
    backward chain leaves in the density head tile."""

    @staticmethod
    def forward(ctx, model, encf, viewf, n_samples, ray_div, n_rays, *params):
        need_grad = any(ctx.needs_input_grad[6:])
        pk = packed_for(model, need_bwd=need_grad)
        d, rgb, ang, acts, ln = mlp_forward(pk, encf, viewf, n_samples, ray_div, n_rays, save=need_grad)
        ctx.pk, ctx.acts, ctx.n = pk, acts, n_samples
        ctx.names = _present(SAR_PARAM_NAMES, dict(model.named_parameters()))
        ctx.params = {n: p.detach() for n, p in zip(ctx.names, params)}
        ctx.save_for_backward(d, rgb, ang, ln)
        return d.reshape(-1, 1), rgb, ang

    @staticmethod
    def backward(ctx, g_density, g_rgb, g_angle):
        d, rgb, ang, ln = ctx.saved_tensors
        pk, P = ctx.pk, ctx.params
        dys = mlp_backward(pk, ctx.acts, ctx.n, g_density, g_rgb, rgb, g_angle, d, ang, ln)
        dev = d.device
        grads = {n: torch.zeros_like(P[n]) for n in ctx.names}
        grads['_dens.weight'], grads['_dens.bias'] = torch.zeros(3, pk.H, device=dev), torch.zeros(3, device=dev)
        grads['_heads.weight'] = torch.zeros(5, pk.W if pk.view else pk.H, device=dev)       # the heads' input width
        grads['_heads.bias'] = torch.zeros(5, device=dev)
        mlp_weight_grads(pk, ctx.acts, dys, ctx.n, grads)
        ctx.acts = None
        grads.update(sar_head_gradients(P, grads))
        return (None, None, None, None, None, None) + tuple(grads[n] for n in ctx.names)


def sar_head_gradients(P, grads):
    """Gradients of the SAR heads from the sums the weight-gradient pass left in the scratch pairs `_dens`, `_heads`.
    d = sum_f w_f (gamma_f xhat_f + beta_f) + b, xhat = (x - mean) rstd:
    A_f = sum_m g rstd x_f, B = sum_m g rstd mean, Sg = sum_m g  ->  dL/d(w gamma)_f = A_f - B."""
    gamma, beta = P['final_density.0.weight'], P['final_density.0.bias']
    w = P['final_density.1.weight'][0]
    du = grads['_dens.weight'][0] - grads['_dens.bias'][1]
    sg = grads['_dens.bias'][2]
    return {'final_density.1.weight': (gamma * du + beta * sg)[None], 'final_density.1.bias': sg[None],
            'final_density.0.weight': w * du, 'final_density.0.bias': w * sg,
            'final_rgb.0.weight': grads['_heads.weight'][:3], 'final_rgb.0.bias': grads['_heads.bias'][:3],
            'final_angle.0.weight': grads['_heads.weight'][3:], 'final_angle.0.bias': grads['_heads.bias'][3:]}


def add_sar_head_gradients(P, grads):
    """`grads[k] += sar_head_gradients(P, grads)[k]` in ten launches instead of sixteen (fused multiply-adds; the
    products round once instead of twice, inside the tolerance of the fp32 atomic sums they follow)."""
    gamma, beta = P['final_density.0.weight'], P['final_density.0.bias']
    w = P['final_density.1.weight'][0]
    du = grads['_dens.weight'][0] - grads['_dens.bias'][1]
    sg = grads['_dens.bias'][2]
    grads['final_density.1.weight'][0].addcmul_(gamma, du).addcmul_(beta, sg)
    grads['final_density.1.bias'].add_(sg)
    grads['final_density.0.weight'].addcmul_(w, du)
    grads['final_density.0.bias'].addcmul_(w, sg)
    grads['final_rgb.0.weight'].add_(grads['_heads.weight'][:3])
    grads['final_rgb.0.bias'].add_(grads['_heads.bias'][:3])
    grads['final_angle.0.weight'].add_(grads['_heads.weight'][3:])
    grads['final_angle.0.bias'].add_(grads['_heads.bias'][3:])


def query_sar(model, t_vals, origins, ray_d, radii, viewf, R, S):
    """One level of the SAR MLP on the bf16 path.  Row i takes the view features of ray i mod R, as upstream's
    tiling does (model.py:351-354)."""
    cfg = model.config
    encf = encode_frag(t_vals, origins, ray_d, radii, model.ray_shape, cfg.min_deg, cfg.max_deg)
    p = dict(model.named_parameters())
    params = [p[k] for k in _present(SAR_PARAM_NAMES, p)]
    if not (torch.is_grad_enabled() and any(q.requires_grad for q in params)):      # inference: nothing saved
        d, rgb, ang, _, _ = mlp_forward(packed_for(model, need_bwd=False), encf, viewf, R * S, R, R, save=False)
        return d.reshape(-1, 1), rgb, ang
    return _MlpSar.apply(model, encf, viewf, R * S, R, R, *params)
