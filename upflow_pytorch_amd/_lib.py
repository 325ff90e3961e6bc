"""ctypes binding of libupflow_hip.so (include/upflow_hip.h) — the only way this package computes.

There is deliberately NO CPU fallback: if the library is missing or a call is made on a non-GPU
tensor the call raises.  (The CPU restatement of the reference lives in oracle/ and is test
infrastructure only.)
"""
import ctypes
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('UPF_HIP_LIB') or os.path.join(_PKG, 'libupflow_hip.so')   # (override: A/B runs of two builds)

UPF_F32, UPF_F16, UPF_BF16 = 0, 1, 2
MASK_NONE, MASK_LITERAL, MASK_ROBUST = 0, 1, 2
_DTYPES = {torch.float32: UPF_F32, torch.float16: UPF_F16, torch.bfloat16: UPF_BF16}

_c = ctypes
_vp, _i, _f, _ll = _c.c_void_p, _c.c_int, _c.c_float, _c.c_longlong

# symbol -> argtypes; every function declared in include/upflow_hip.h (tests check the two lists agree)
SIGNATURES = {
    'upf_corr81_forward': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _ll, _f, _vp],
    'upf_corr81_forward_timed': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _ll, _f, _vp, _i, _c.POINTER(_f), _c.POINTER(_f)],
    'upf_corr81_norm_forward': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _ll, _f, _vp, _vp],
    'upf_corr81_norm_forward_timed': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _ll, _f, _vp, _vp, _i, _c.POINTER(_f), _c.POINTER(_f)],
    'upf_corr81_norm_forward_c8_timed': [_vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _c.POINTER(_f), _c.POINTER(_f)],
    'upf_corr81_norm_forward_pitched': [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _ll, _f, _vp, _vp],
    'upf_corr81_norm_forward_c8_pitched': [_vp, _vp, _i, _vp, _ll, _i, _i, _i, _i, _i, _f, _vp, _vp],
    'upf_corr81_norm_forward_c8_timed_pitched': [_vp, _vp, _i, _vp, _ll, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _c.POINTER(_f), _c.POINTER(_f)],
    'upf_corr81_norm_forward_c8_timed_mixed': [_vp, _vp, _i, _vp, _ll, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _c.POINTER(_f), _c.POINTER(_f)],
    'upf_warp_forward_pitched': [_vp, _ll, _i, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'upf_conv_forward_gated': [_vp, _ll, _i, _vp, _vp, _vp, _ll, _i, _vp, _ll, _vp, _ll, _f, _i, _i, _i, _i, _i, _i, _vp],
    'upf_conv_forward_pitched': [_vp, _ll, _i, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    'upf_conv_forward_c8_pitched': [_vp, _ll, _i, _vp, _ll, _i, _i, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    'upf_corr81_norm_forward_mixed': [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _ll, _f, _vp, _vp],
    'upf_corr81_norm_forward_c8_mixed': [_vp, _vp, _i, _vp, _ll, _i, _i, _i, _i, _i, _i, _f, _vp, _vp],
    'upf_conv1x1_forward_mixed': [_vp, _ll, _i, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _vp],
    'upf_conv1x1_forward_c8_dual': [_vp, _ll, _i, _vp, _vp, _vp, _ll, _vp, _ll, _i, _i, _i, _i, _i, _f, _i, _i, _vp],
    'upf_corr81_backward': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'upf_correlation_forward': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'upf_correlation_backward': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'upf_correlation_out_shape': [_i, _i, _i, _i, _i, _i, _i, _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i)],
    'upf_warp_forward': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    'upf_warp_forward_strided': [_vp, _ll, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _vp],
    'upf_warp_forward_c8': [_vp, _ll, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _vp],
    'upf_flow_update': [_vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _vp],
    'upf_flow_update_c8': [_vp, _vp, _vp, _vp, _ll, _i, _i, _i, _vp],
    'upf_corr81_norm_forward_c8': [_vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _f, _vp, _vp],
    'upf_warp_backward': [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    'upf_flow_upsample_forward': [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    'upf_flow_upsample_backward': [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    'upf_sgu_blend_forward': [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'upf_sgu_blend_forward_flow16': [_vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _vp],
    'upf_sgu_blend_backward': [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'upf_normalize_forward': [_vp, _vp, _vp, _vp, _vp, _ll, _i, _i, _vp],
    'upf_normalize_backward': [_vp, _vp, _vp, _vp, _ll, _i, _i, _vp],
    'upf_conv_pack_weights': [_vp, _vp, _i, _i, _i, _i, _vp],
    'upf_conv_forward': [_vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    'upf_conv_x3_pack_weights': [_vp, _vp, _i, _i, _i, _vp],
    'upf_conv_x3_forward': [_vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    'upf_mfma_f16_denorm_probe': [_vp, _vp],
    'upf_conv_pack_weights_f32': [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    'upf_conv_pack_stacked_dgrad': [_c.POINTER(_vp), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _i, _c.POINTER(_vp), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _i, _i, _vp],
    'upf_conv_pack_weights_f32_multi': [_c.POINTER(_vp), _c.POINTER(_vp), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _c.POINTER(_i), _i, _i, _vp],
    'upf_conv_pack_weights_kmap': [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp],
    'upf_conv_forward_c8': [_vp, _ll, _i, _vp, _ll, _i, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    'upf_conv_pack_weights_kmap16': [_vp, _vp, _i, _i, _vp, _i, _i, _vp],
    'upf_conv_forward_c8_narrow': [_vp, _ll, _i, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _f, _i, _vp],
    'upf_conv_forward_c8_split': [_vp, _ll, _i, _vp, _vp, _vp, _ll, _i, _vp, _ll, _i, _i, _i, _i, _i, _f, _i, _vp],
    'upf_conv_forward_c8_narrow_init': [_vp, _ll, _i, _vp, _vp, _ll, _i, _i, _vp, _ll, _i, _i, _i, _i, _i, _f, _i, _vp],
    'upf_conv_pair_forward': [_vp, _ll, _i, _i, _vp, _vp, _f, _i, _i, _vp, _vp, _f, _i, _i, _vp, _ll, _i, _i, _i, _i, _i, _i, _vp],
    'upf_leaky_backward': [_vp, _vp, _vp, _ll, _f, _i, _vp],
    'upf_conv_wgrad': [_vp, _ll, _vp, _ll, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'upf_conv_bias_grad': [_vp, _ll, _vp, _vp, _i, _i, _i, _i, _vp],
    'upf_conv_wgrad_multi': [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'upf_conv_wgrad_multi_bias': [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp],
    'upf_space_to_depth2': [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'upf_conv_wgrad_s2d': [_vp, _i, _vp, _vp, _i, _i, _i, _vp],
    'upf_act_grad': [_vp, _ll, _vp, _ll, _vp, _ll, _vp, _ll, _vp, _i, _i, _i, _f, _i, _vp],
    'upf_conv_bias_grad_finish': [_vp, _i, _vp, _i, _vp],
    'upf_census_forward': [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    'upf_census_backward': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    'upf_boundary_warp_forward': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'upf_boundary_warp_backward': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'upf_robust_loss_forward': [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp],
    'upf_robust_loss_backward': [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp],
    'upf_grey_forward': [_vp, _vp, _i, _i, _vp],
    'upf_msd_upup_forward': [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _vp],
    'upf_msd_upup_backward': [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _vp],
    'upf_smooth_edge1_forward': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'upf_smooth_edge1_backward': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'upf_div_selftest': [_i, _vp, _vp],
    'upf_occ_check': [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp],
}



class WgradLevel(_c.Structure):
    """upf_wgrad_level (include/upflow_hip.h): one use of a shared convolution in a multi-level weight gradient."""
    _fields_ = [('x', _vp), ('x_batch_stride', _ll), ('grad_pre', _vp), ('g_batch_stride', _ll), ('B', _i), ('H', _i), ('W', _i)]


_lib = None


class UpflowHipError(RuntimeError):
    """Raised for a rejected argument or a failed launch (the reference raises RuntimeError through
    AT_ERROR("CUDA call failed"), correlation_cuda.cc:81-83)."""


def lib():
    """Load libupflow_hip.so once; fail loudly if it was not built (python __graft_entry__.py build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UpflowHipError(
                '%s is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = _i
        L.upf_normalize_workspace_bytes.argtypes = [_ll, _i]
        L.upf_normalize_workspace_bytes.restype = _ll
        L.upf_corr81_norm_supported.argtypes = [_i, _i]
        L.upf_corr81_norm_supported.restype = _i
        L.upf_corr81_norm_workspace_bytes.argtypes = [_i, _i, _i, _i]
        L.upf_corr81_norm_workspace_bytes.restype = _ll
        L.upf_corr_set_option.argtypes = [_c.c_char_p, _i]
        L.upf_corr_set_option.restype = _i
        L.upf_warp_backward_workspace_bytes.argtypes = [_i, _i, _i, _i]
        L.upf_warp_backward_workspace_bytes.restype = _ll
        L.upf_sgu_blend_backward_workspace_bytes.argtypes = [_i, _i, _i, _i, _i]
        L.upf_sgu_blend_backward_workspace_bytes.restype = _ll
        L.upf_sgu_blend_forward_workspace_bytes.argtypes = [_i, _i, _i, _i, _i]
        L.upf_sgu_blend_forward_workspace_bytes.restype = _ll
        L.upf_conv_wgrad_supported.argtypes = [_i] * 8
        L.upf_conv_wgrad_supported.restype = _i
        L.upf_conv_wgrad_workspace_bytes.argtypes = [_i] * 7
        L.upf_conv_wgrad_workspace_bytes.restype = _ll
        L.upf_conv_wgrad_multi_workspace_bytes.argtypes = [_vp, _i, _i, _i, _i, _i]
        L.upf_conv_wgrad_multi_workspace_bytes.restype = _ll
        L.upf_conv_bias_grad_workspace_bytes.argtypes = [_i]
        L.upf_conv_bias_grad_workspace_bytes.restype = _ll
        L.upf_loss_partials.argtypes = [_ll]
        L.upf_loss_partials.restype = _i
        L.upf_msd_upup_partials.argtypes = [_i, _i, _i]
        L.upf_msd_upup_partials.restype = _i
        L.upf_conv_packed_bytes.argtypes = [_i, _i, _i]
        L.upf_conv_packed_bytes.restype = _ll
        L.upf_conv_x3_packed_bytes.argtypes = [_i, _i, _i]
        L.upf_conv_x3_packed_bytes.restype = _ll
        L.upf_conv_set_option.argtypes = [_c.c_char_p, _i]
        L.upf_conv_set_option.restype = _i
        L.upf_conv_c8_set_option.argtypes = [_c.c_char_p, _i]
        L.upf_conv_c8_set_option.restype = _i
        L.upf_conv_x3_set_option.argtypes = [_c.c_char_p, _i]
        L.upf_conv_x3_set_option.restype = _i
        L.upf_conv_packed_bytes_k.argtypes = [_i, _i, _i]
        L.upf_conv_packed_bytes_k.restype = _ll
        L.upf_conv_c8_k.argtypes = [_i, _i]
        L.upf_conv_c8_k.restype = _i
        L.upf_conv_packed_bytes_k16.argtypes = [_i, _i]
        L.upf_conv_packed_bytes_k16.restype = _ll
        L.upf_version.restype = _c.c_char_p
        L.upf_last_error.restype = _c.c_char_p
        _lib = L
        # tuning hook: UPF_CONV_OPTS="sk_grid=48,rpw4_min=512" -> upf_conv_set_option (include/upflow_hip.h)
        for item in filter(None, os.environ.get('UPF_CONV_OPTS', '').split(',')):
            k, v = item.split('=')
            if L.upf_conv_set_option(k.strip().encode(), int(v)) == -2 ** 31:
                raise UpflowHipError('UPF_CONV_OPTS: unknown option %r' % k)
    return _lib


def version():
    return lib().upf_version().decode()


def dtype_code(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise UpflowHipError('unsupported dtype %s (float32 / float16 / bfloat16 only)' % t.dtype)


def check_gpu(*tensors, contiguous=True):
    """Every operand must be a contiguous tensor on the same GPU (the reference's kernels assume
    contiguous NCHW, correlation_cuda_kernel.cu:15-39, and never validate; we do).  contiguous=False: the caller has checked the
    layout itself (row-pitched tensors, ops.nchw_pitch)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise UpflowHipError('upflow_pytorch_amd operators run on the GPU only (got a %s tensor); '
                                 'there is no CPU fallback' % t.device)
        if contiguous and not t.is_contiguous():
            raise UpflowHipError('operand must be contiguous NCHW')
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise UpflowHipError('operands on different devices: %s vs %s' % (dev, t.device))
    return dev


def stream_ptr(device):
    return _vp(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return _vp(t.data_ptr()) if t is not None else _vp(0)


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise UpflowHipError('%s failed (%d): %s' % (name, rc, lib().upf_last_error().decode()))
