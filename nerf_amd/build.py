"""Build libnerf_amd.so (gfx950) in-tree with hipcc.

    python -m nerf_amd.build [--force]

Each .hip translation unit is compiled to an object with its own flags, then linked into
`nerf_amd/lib/libnerf_amd.so`.  hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(LIBDIR, 'obj')
LIB = os.path.join(LIBDIR, 'libnerf_amd.so')

COMMON = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function']
# translation unit -> extra flags.  The sampling / compositing kernels must round every fp32
# operation separately (parity with the reference's op-by-op torch arithmetic).
# mlp_bf16.hip is compiled in parts (-DNERF_NET_PART=p): the chain kernels of each large network geometry take
# minutes to compile, so each gets its own object and they build side by side.
UNITS = {
    'api.o': ('api.hip', []),
    'sampling.o': ('sampling.hip', ['-ffp-contract=off']),
    'composite.o': ('composite.hip', ['-ffp-contract=off']),
    'sar_pulse.o': ('sar_pulse.hip', []),
    'gemm_f32.o': ('gemm_f32.hip', []),
    # VGPR-form matrix-core instructions where an accumulator tile lives in the vector registers: the outputs of the
    # larger launches exceed the 256 accumulation registers, and the default form copies such tiles in and out
    # of them around every product (layer 4's launch: 1.10 -> 0.86 ms)
    'wgrad8.o': ('wgrad8.hip', ['-mllvm', '-amdgpu-mfma-vgpr-form']),
}
UNITS.update({'mlp_bf16_p%d.o' % p: ('mlp_bf16.hip', ['-DNERF_NET_PART=%d' % p]) for p in range(8)})   # default contraction
JOBS = int(os.environ.get('MAX_JOBS', '8'))


def _hipcc():
    for c in ('/opt/rocm/bin/hipcc', 'hipcc'):
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return 'hipcc'


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.h', '.inc'))]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'nerf_amd.h'))
    headers.append(os.path.abspath(__file__))
    jobs, objs = [], []
    for name, (unit, extra) in UNITS.items():
        src = os.path.join(CSRC, unit)
        obj = os.path.join(OBJDIR, name)
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([_hipcc()] + COMMON + extra + ['-c', src, '-o', obj])
    for f in os.listdir(OBJDIR):                      # objects of an older unit list must not be linked
        if f.endswith('.o') and f not in UNITS:
            os.remove(os.path.join(OBJDIR, f))

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(JOBS, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
