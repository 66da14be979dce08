"""Per-kernel-family timing with HIP events on the launching stream (used by bench.py).

`with timed('mlp_fwd', flops=..., bytes=...)` brackets a group of launches with two events recorded on
torch's current stream -- the stream every kernel of this package is launched on -- and accumulates the
algorithmic work handed in (`flops`: the reference's FLOPs for this work; `flops_exec`: the FLOPs the kernel
issues for the true layer sizes, where the two differ).  Disabled (zero overhead beyond one `if`) unless `enable()` was called."""
import contextlib

import torch

_enabled = False
_records = {}


def enable(flag=True):
    global _enabled
    _enabled = flag
    if flag:
        _records.clear()


@contextlib.contextmanager
def timed(name, flops=0.0, bytes_=0.0, flops_exec=None):
    if not _enabled:
        yield
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    yield
    e1.record()
    r = _records.setdefault(name, dict(events=[], flops=0.0, flops_exec=0.0, bytes=0.0, calls=0))
    r['events'].append((e0, e1))
    r['flops'] += flops
    r['flops_exec'] += flops if flops_exec is None else flops_exec
    r['bytes'] += bytes_
    r['calls'] += 1


def summary():
    """name -> dict(ms total, calls, flops, bytes); synchronises."""
    torch.cuda.synchronize()
    out = {}
    for name, r in _records.items():
        ms = sum(a.elapsed_time(b) for a, b in r['events'])
        out[name] = dict(ms=ms, calls=r['calls'], flops=r['flops'], flops_exec=r['flops_exec'], bytes=r['bytes'])
    return out
