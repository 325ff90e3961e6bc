"""`import correlation_cuda` — the reference's native module BY ITS OWN NAME.

`/root/reference/model/correlation_package/correlation.py:4` does `import correlation_cuda` (the pybind module built
from correlation_cuda.cc:169-172).  With this repository's root on `sys.path` (or this file copied next to the
reference's `model/`), that import resolves here and the reference's `CorrelationFunction` calls
`correlation_cuda.forward / backward` with its 11 / 13 positional arguments unchanged; the work is done by
libupflow_hip.so through `upflow_pytorch_amd.correlation_cuda` (ownership, return value and error behaviour documented
there).  `upflow_pytorch_amd.install_correlation_cuda()` registers the same module in `sys.modules` for programs that
cannot touch `sys.path`.
"""
from upflow_pytorch_amd.correlation_cuda import forward, backward  # noqa: F401

__all__ = ['forward', 'backward']
