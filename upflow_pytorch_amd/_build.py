"""Build libupflow_hip.so (the C-ABI library of include/upflow_hip.h) in-tree with hipcc for gfx950.

No torch headers are involved: the library is plain HIP C++ behind an extern "C" surface, so a build
is a handful of `hipcc -c` calls (seconds) and cross-compiles without a GPU.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libupflow_hip.so')

# (source, extra flags).  The sampling kernels must not contract a*b+c into FMAs: the warp validity
# mask `grid_sample(ones) >= 1.0` of the reference is bit-sensitive (csrc/sampling.hpp).
SOURCES = [
    ('api.hip', []),
    ('corr81_fwd.hip', ['-ffp-contract=off']),      # (the fused normalisation must round like misc.hip's)
    ('corr81_bwd.hip', []),
    ('conv3x3.hip', []),
    ('conv_c8.hip', []),
    ('conv_x3.hip', []),
    ('conv_wgrad.hip', []),
    ('warp.hip', ['-ffp-contract=off']),
    ('sgu_blend.hip', ['-ffp-contract=off']),
    ('misc.hip', ['-ffp-contract=off']),
    ('loss.hip', ['-ffp-contract=off']),
]
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libupflow_hip.so next to this file."""
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith('.hpp')]
    headers.append(os.path.join(os.path.dirname(PKG), 'include', 'upflow_hip.h'))
    objdir = os.path.join(PKG, 'build')
    os.makedirs(objdir, exist_ok=True)
    objs = []
    cc = hipcc()
    jobs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([cc] + COMMON + extra + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    if jobs:                                           # independent translation units: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    if force or _stale(LIB, objs):
        cmd = [cc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
