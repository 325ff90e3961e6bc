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


def _functions(text):
    """{symbol: body} of an llvm-objdump disassembly."""
    out, name, buf = {}, None, []
    for line in text.split('\n'):
        m = re.match(r'^[0-9a-f]+ <([^>]+)>:', line)
        if m:
            if name:
                out[name] = '\n'.join(buf)
            name, buf = m.group(1), []
        elif name:
            buf.append(line)
    if name:
        out[name] = '\n'.join(buf)
    return out


WATERFALL = re.compile(r's_and_saveexec_b64[^\n]*\n[^\n]*buffer_(?:load|store)_[^\n]*\n[^\n]*s_xor_b64 exec, exec[^\n]*\n[^\n]*s_cbranch_execnz')


def test_no_waterfall_loops_in_the_hot_kernels(disassembly):
    """hipcc wraps a buffer access whose DESCRIPTOR it considers divergent in a waterfall loop (v_readfirstlane x4,
    v_cmp x2, exec juggling, a branch) — the weight-gradient kernel's staging waves carried 13 of them per tile for three rounds
    because its tile decode divides on the vector unit.  The default kernels must have none (the descriptors are made uniform
    explicitly); the regression shows in the disassembly long before it shows in a benchmark."""
    bad = {}
    for src in ('conv_wgrad.hip', 'conv3x3.hip', 'conv_c8.hip', 'corr81_fwd.hip', 'corr81_bwd.hip', 'warp.hip', 'sgu_blend.hip', 'misc.hip', 'loss.hip'):
        for name, body in _functions(disassembly[src]).items():
            if src == 'conv_wgrad.hip' and 'wgrad_pc_kernel' not in name and 'reduce' not in name and 'act_grad' not in name:
                continue                              # (wgrad_kernel, the UPF_WGRAD_MODE=1 A/B kernel, is not on the default path)
            n = len(WATERFALL.findall(body))
            if n:
                bad[name[:80]] = n
    assert not bad, 'waterfall loops around buffer accesses: %s' % bad


def test_weight_gradient_producers_keep_the_other_set_of_loads_in_flight(disassembly):
    """the staging waves of wgrad_pc_kernel hold two tiles of loads in flight (13 + 13 16-byte loads) and land the older
    one: every s_waitcnt before an LDS write inside the tile loop must leave >= 13 loads outstanding (vmcnt(25) ... vmcnt(13)).
    Two code-generation accidents drained the queue instead (vmcnt(12) ... vmcnt(0)): a conditionally skipped second half of the
    loop, and loads duplicated in the two arms of a branch.  Checked on the 3x3 dilation-1 kernels (13 tasks per thread)."""
    fns = _functions(disassembly['conv_wgrad.hip'])
    checked = 0
    for name, body in fns.items():
        if not re.search(r'wgrad_pc_kernelINS_\w+ELi1ELb[01]E', name):
            continue
        lines = body.split('\n')
        # the waits that guard an LDS staging write (only the producers write 16-byte blocks): prologue + both halves of the loop
        waits = []
        for i, l in enumerate(lines):
            if 'ds_write_b128' in l:
                for back in lines[max(0, i - 3):i]:
                    m = re.search(r's_waitcnt vmcnt\((\d+)\)', back)
                    if m:
                        waits.append(int(m.group(1)))
        assert len(waits) >= 39, (name, len(waits))
        assert min(waits) >= 13, (name, waits)
        checked += 1
    assert checked >= 4                               # bf16 / fp16 x aligned / mixed


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
