"""Properties of the BUILT gfx950 code objects (no GPU needed: hipcc cross-compiles, llvm-objdump disassembles).

DESIGN §4c: packed-fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) were observed to return wrong results on
MI355X while a wave of a v_mfma_f32_16x16x32 kernel shares the SIMD, so the library is built with the `packed-fp32-ops` target
feature switched off (upflow_pytorch_amd/_build.py: NO_PACKED_FP32).  That is a compiler flag — one toolchain update, one
`#pragma`, one new translation unit built by hand away from silently returning — so the absence of those instructions is
asserted on the objects themselves."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'


def _disassemble(obj, tmp):
    fat = os.path.join(tmp, os.path.basename(obj) + '.fatbin')
    co = os.path.join(tmp, os.path.basename(obj) + '.co')
    subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, obj])
    subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + fat,
                           '--targets=' + TARGET, '--output=' + co])
    return subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--mcpu=gfx950', co], check=True, stdout=subprocess.PIPE,
                          text=True).stdout


@pytest.fixture(scope='module')
def disassembly(tmp_path_factory):
    if not all(os.path.exists(os.path.join(LLVM, t)) for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-objdump')):
        pytest.skip('ROCm LLVM tools not found')
    from upflow_pytorch_amd import _build
    _build.build()
    tmp = str(tmp_path_factory.mktemp('codegen'))
    objdir = os.path.join(ROOT, 'upflow_pytorch_amd', 'build')
    out = {}
    for src, _ in _build.SOURCES:
        out[src] = _disassemble(os.path.join(objdir, src.replace('.hip', '.o')), tmp)
    return out


def test_no_packed_fp32_instructions_in_any_code_object(disassembly):
    bad = {}
    for src, text in disassembly.items():
        hits = re.findall(r'\bv_pk_(?:add|mul|fma)_f32\b', text)
        if hits:
            bad[src] = len(hits)
    assert not bad, 'packed-fp32 VALU instructions in the built code objects (DESIGN §4c): %s' % bad


def test_the_matrix_core_kernels_are_matrix_core_code(disassembly):
    """The hot kernels are hand-written CDNA4 code: the convolutions issue 32x32x16 and 16x16x32 MFMAs, the cost volume the
    4x4x4 16-block form, and the convolutions stage octet tensors by LDS-DMA (buffer_load ... lds)."""
    conv = disassembly['conv3x3.hip'] + disassembly['conv_c8.hip']
    assert len(re.findall(r'v_mfma_f32_32x32x16_bf16', conv)) > 1000
    assert len(re.findall(r'v_mfma_f32_16x16x32_bf16', disassembly['conv_c8.hip'])) > 50
    assert len(re.findall(r'v_mfma_f32_4x4x4_16b_bf16', disassembly['corr81_fwd.hip'])) > 100
    assert len(re.findall(r'buffer_load_dwordx4 .* lds', disassembly['conv_c8.hip'])) > 50


def test_every_object_was_built_with_the_recorded_flags():
    """_build.py re-compiles an object whose recorded command-line hash differs from the current flags: the hash of
    every object on disk is the current one."""
    from upflow_pytorch_amd import _build
    _build.build()
    objdir = os.path.join(ROOT, 'upflow_pytorch_amd', 'build')
    for src, extra in _build.SOURCES:
        o = os.path.join(objdir, src.replace('.hip', '.o'))
        assert open(o + '.flags').read().strip() == _build._flags_tag(extra), src
        assert not _build._flags_stale(o, extra)
    assert _build._flags_stale(os.path.join(objdir, 'api.o'), ['-DSOMETHING_ELSE'])
