"""The C-ABI library builds, loads, and exports every symbol include/upflow_hip.h declares
(no compute: runs without a GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'upflow_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(upf_[a-z0-9_]+)\s*\(', txt)))


def test_header_symbols_exported():
    import __graft_entry__ as g
    g.build()
    from upflow_pytorch_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), 'missing symbol %s' % n
    bound = set(_lib.SIGNATURES) | {'upf_version', 'upf_last_error', 'upf_normalize_workspace_bytes', 'upf_conv_packed_bytes', 'upf_conv_x3_packed_bytes',
             'upf_conv_set_option', 'upf_conv_c8_set_option', 'upf_conv_x3_set_option', 'upf_conv_packed_bytes_k', 'upf_conv_packed_bytes_k16', 'upf_conv_c8_k', 'upf_corr_set_option', 'upf_loss_partials', 'upf_msd_upup_partials', 'upf_conv_wgrad_supported', 'upf_conv_wgrad_workspace_bytes', 'upf_conv_wgrad_multi_workspace_bytes', 'upf_conv_bias_grad_workspace_bytes', 'upf_warp_backward_workspace_bytes', 'upf_sgu_blend_backward_workspace_bytes', 'upf_sgu_blend_forward_workspace_bytes', 'upf_corr81_norm_supported', 'upf_corr81_norm_workspace_bytes'}
    assert bound == set(names), (bound ^ set(names))


def test_version_and_error_strings():
    from upflow_pytorch_amd import _lib
    assert 'gfx950' in _lib.version()
    assert isinstance(_lib.lib().upf_last_error().decode(), str)


def test_no_cpu_fallback():
    """The product has no CPU path: CPU tensors are rejected loudly (never silently computed)."""
    import pytest
    import torch
    from upflow_pytorch_amd import ops
    a = torch.zeros(1, 4, 8, 8)
    with pytest.raises(RuntimeError):
        ops.corr81(a, a)
    with pytest.raises(RuntimeError):
        ops.warp(a, torch.zeros(1, 2, 8, 8))
    with pytest.raises(RuntimeError):
        ops.normalize(a)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'upflow_pytorch_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(d, f)


def test_correlation_cuda_importable_by_its_own_name():
    """`import correlation_cuda` — the literal import of the reference's model/correlation_package/correlation.py:4 — resolves to
    the top-level shim (repo root on sys.path), with the two pybind entry points (correlation_cuda.cc:169-172) and their 11 / 13
    positional arguments; install_correlation_cuda() registers the same functions under that name without touching sys.path."""
    import importlib
    import inspect
    import sys
    sys.modules.pop('correlation_cuda', None)
    mod = importlib.import_module('correlation_cuda')
    assert len(inspect.signature(mod.forward).parameters) == 11 and len(inspect.signature(mod.backward).parameters) == 13
    import upflow_pytorch_amd
    from upflow_pytorch_amd import correlation_cuda as impl
    assert mod.forward is impl.forward and mod.backward is impl.backward
    sys.modules.pop('correlation_cuda', None)
    assert upflow_pytorch_amd.install_correlation_cuda() is impl and sys.modules['correlation_cuda'] is impl
    sys.modules.pop('correlation_cuda', None)
