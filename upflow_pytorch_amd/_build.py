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
    ('conv_pair.hip', []),
    ('conv_x3.hip', []),
    ('conv_wgrad.hip', []),
    ('warp.hip', ['-ffp-contract=off']),
    ('sgu_blend.hip', ['-ffp-contract=off']),
    ('misc.hip', ['-ffp-contract=off']),
    ('loss.hip', ['-ffp-contract=off']),
]
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']
# No packed-fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) anywhere in the library: on MI355X they were
# observed to return wrong results while a wave of a v_mfma_f32_16x16x32 kernel shares the SIMD — i.e. as soon as two of this
# library's kernels run concurrently on two HIP streams (runtime.PipelinedInference; csrc/corr81_allc_kernel.hpp has the record,
# tools/corr_race3.py the reproducer).  The feature is switched off for the device pass (the host pass prints "not a recognized
# feature for this target (ignoring feature)", which is filtered below); the arithmetic is the same, in scalar fp32 instructions.
NO_PACKED_FP32 = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']


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


def _flags_tag(extra):
    """Hash of everything on an object's command line besides its source: an object built with other flags (e.g. before
    NO_PACKED_FP32 existed — a CORRECTNESS flag) is stale whatever its time stamp says."""
    import hashlib
    return hashlib.sha256(' '.join(COMMON + NO_PACKED_FP32 + list(extra)).encode()).hexdigest()[:16]


def _flags_stale(obj, extra):
    try:
        return open(obj + '.flags').read().strip() != _flags_tag(extra)
    except OSError:
        return True


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
        if force or _stale(o, [s] + headers) or _flags_stale(o, extra):
            jobs.append(([cc] + COMMON + NO_PACKED_FP32 + extra + ['-c', s, '-o', o], o, extra))

    def run(job):
        cmd, obj, extra = job
        if os.path.exists(obj + '.flags'):
            os.remove(obj + '.flags')
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        err = '\n'.join(l for l in r.stderr.splitlines() if 'is not a recognized feature for this target' not in l)
        if err.strip():
            sys.stderr.write(err + '\n')
        if r.returncode:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        with open(obj + '.flags', 'w') as f:
            f.write(_flags_tag(extra) + '\n')
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
