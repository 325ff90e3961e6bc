"""upflow_pytorch_amd — MI355X (gfx950) native hot path of UPFlow behind the reference's operator API.

    ops                    autograd operators, one hand-written HIP launch each (libupflow_hip.so)
    model.upflow           UPFlow_net drop-in shell (same config flags / state_dict keys / dict I/O)
    model.pwc_modules      conv factory, FeatureExtractor, WarpingLayer_no_div, estimators
    model.correlation_package.correlation   Correlation / CorrelationFunction
    utils.tools            tools.abstract_config / abstract_model / torch_warp / occ_check_model
    correlation_cuda       the reference's legacy pybind FFI signature over the C-ABI

The package computes on the GPU only; importing it needs neither a GPU nor the built library,
calling an operator needs both.
"""
__version__ = '0.1.0'


def install_correlation_cuda():
    """Make `import correlation_cuda` (model/correlation_package/correlation.py:4 of the reference) resolve to this package's
    implementation of the legacy FFI without touching sys.path; returns the module.  A module of that name that is already
    imported (the reference's own CUDA build) is left alone."""
    import sys
    from . import correlation_cuda as mod
    return sys.modules.setdefault('correlation_cuda', mod)
