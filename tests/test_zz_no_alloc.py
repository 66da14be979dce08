"""The allocation-free SAR scatter entry point against the ordinary one, through the C ABI."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(R, S, nsam, seed):
    g = torch.Generator().manual_seed(seed)
    near, mpp = 900.0, 0.5
    far = near + mpp * (nsam - 2)
    t = torch.sort(torch.rand(R, S + 1, generator=g) * (far - near) + near, dim=-1)[0]
    t[R // 2] = t[R // 2 - 1]                               # two rays in one range bin
    rgb, ang = torch.rand(R, S, 3, generator=g), torch.rand(R, S, 2, generator=g)
    den = torch.rand(R, S, generator=g) * 0.2
    den[R // 2] = den[R // 2 - 1]
    az, el = torch.randn(R, generator=g) * 0.04, torch.randn(R, generator=g) * 0.1
    dirs = torch.stack([torch.sin(az) * torch.cos(el), torch.cos(az) * torch.cos(el), -torch.sin(el)], -1)
    power = torch.rand(R, generator=g) * 1e9
    bins = torch.arange(nsam) * mpp + near
    return [x.cuda().contiguous() for x in (rgb, ang, den, power, t, dirs, bins)]


def _call(static, ins, R, S, nsam):
    from nerf_amd import _lib
    L = _lib.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    dev = 'cuda'
    out = dict(pulse=torch.full((nsam, 2), 7.0, device=dev), dist=torch.empty(R, device=dev), acc=torch.empty(R, device=dev),
               w=torch.empty(R, S, device=dev), alpha=torch.empty(R, S, device=dev),
               ray_bin=torch.empty(R, device=dev, dtype=torch.int32), comp=torch.empty(R, 5, device=dev),
               status=torch.full((1,), 5, device=dev, dtype=torch.int32))
    act = _lib.NerfActivation(1, 1e-4, -1.0)
    head = [P(x) for x in ins] + [R, S, nsam, 0.03, ctypes.byref(act)]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if static:
        rc = L.nerf_scatter_fwd_static(*head, P(out['pulse']), P(out['dist']), P(out['acc']), P(out['w']), P(out['alpha']),
                                       P(out['ray_bin']), P(out['comp']), P(out['status']), st)
    else:
        ray_ret = torch.empty(R, 2, device=dev)
        winner = torch.empty(nsam, device=dev, dtype=torch.int32)
        rc = L.nerf_scatter_fwd(*head, P(out['pulse']), P(out['dist']), P(out['acc']), P(out['w']), P(out['alpha']),
                                P(ray_ret), P(out['ray_bin']), P(out['comp']), P(winner), P(out['status']), st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out


def test_static_scatter_matches_the_allocating_entry_point():
    """Two calls in a row, the second on a smaller pulse (fewer rays, fewer bins: stale winners and returns of the
    first call stay in the library's array beyond the part the second call clears): every output bit for bit
    equal to nerf_scatter_fwd's.  Sizes beyond the array are refused."""
    from nerf_amd import _lib
    for R, S, nsam, seed in ((700, 48, 512, 1), (130, 20, 96, 2)):
        ins = _inputs(R, S, nsam, seed)
        want = _call(False, ins, R, S, nsam)
        got = _call(True, ins, R, S, nsam)
        assert int(want['status']) == 0
        assert float(want['pulse'].abs().sum()) > 0
        for k in want:
            assert torch.equal(got[k], want[k]), (R, k)
    L = _lib.lib()
    null = ctypes.c_void_p(None)
    args = [null] * 7 + [1 << 20, 4, 64, 0.03, None] + [null] * 9
    assert L.nerf_scatter_fwd_static(*args) == -3
    args = [null] * 7 + [16, 4, 1 << 20, 0.03, None] + [null] * 9
    assert L.nerf_scatter_fwd_static(*args) == -3


def test_static_scatter_replays_from_a_captured_graph():
    """The use the entry point exists for: after one ordinary call on the device (which looks the library's array
    up; that lookup is no stream operation and must not happen under capture), the call is captured into a HIP graph
    with torch.cuda.graph -- nothing in it allocates -- and every replay, on fresh inputs copied into the captured
    tensors, gives nerf_scatter_fwd's outputs bit for bit."""
    from nerf_amd import _lib
    L = _lib.lib()
    R, S, nsam = 384, 24, 256
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    _call(True, _inputs(R, S, nsam, 3), R, S, nsam)                     # the priming call
    ins = _inputs(R, S, nsam, 4)
    dev = 'cuda'
    out = dict(pulse=torch.zeros(nsam, 2, device=dev), dist=torch.zeros(R, device=dev), acc=torch.zeros(R, device=dev),
               w=torch.zeros(R, S, device=dev), alpha=torch.zeros(R, S, device=dev),
               ray_bin=torch.zeros(R, device=dev, dtype=torch.int32), comp=torch.zeros(R, 5, device=dev),
               status=torch.full((1,), 5, device=dev, dtype=torch.int32))
    act = _lib.NerfActivation(1, 1e-4, -1.0)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = L.nerf_scatter_fwd_static(*[P(x) for x in ins], R, S, nsam, 0.03, ctypes.byref(act), P(out['pulse']),
                                       P(out['dist']), P(out['acc']), P(out['w']), P(out['alpha']), P(out['ray_bin']),
                                       P(out['comp']), P(out['status']), st)
    assert rc == 0, rc
    for seed in (4, 5):
        for dst, src in zip(ins, _inputs(R, S, nsam, seed)):
            dst.copy_(src)
        graph.replay()
        torch.cuda.synchronize()
        want = _call(False, ins, R, S, nsam)
        assert float(want['pulse'].abs().sum()) > 0
        for k in want:
            assert torch.equal(out[k], want[k]), (seed, k)
