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
UNITS = {
    'api.hip': [],
    'sampling.hip': ['-ffp-contract=off'],
    'composite.hip': ['-ffp-contract=off'],
    'gemm_f32.hip': [],
    'mlp_bf16.hip': [],      # default contraction: MFMA chain, fp32 epilogues
    'wgrad8.hip': [],
}


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
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'nerf_amd.h'))
    headers.append(os.path.abspath(__file__))
    jobs, objs = [], []
    for unit, extra in UNITS.items():
        src = os.path.join(CSRC, unit)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJDIR, unit.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([_hipcc()] + COMMON + extra + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
