"""Drop-in for `/root/reference/model/correlation_package/correlation.py`.

`Correlation(pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)(in1, in2)`
keeps the reference's constructor and call signature (correlation.py:47-61).  The legacy non-static
`CorrelationFunction` (correlation.py:6-44, rejected by modern torch: SURVEY.md §7-H6) becomes a
static autograd Function; the (4,1,4,1,1) configuration — the only one the model builds
(model/upflow.py:561-562) — runs the tuned 81-neighbour HIP kernel with its backward kernels, any
other parameter set the general kernels (forward wherever the reference's kernel stays inside its padded
buffers; gradients for kernel_size 1 / stride1 1, where the reference's backward kernels are the gradient of
its forward — include/upflow_hip.h: upf_correlation_backward).
"""
import torch
from torch.autograd import Function
from torch.nn.modules.module import Module

from ... import ops


def _is81(p, k, md, s1, s2):
    return (p, k, md, s1, s2) == (4, 1, 4, 1, 1)


class CorrelationFunction(Function):
    @staticmethod
    def forward(ctx, input1, input2, pad_size=3, kernel_size=3, max_displacement=20, stride1=1, stride2=2, corr_multiply=1):
        input1 = input1.contiguous()
        input2 = input2.contiguous()
        ctx.is81 = _is81(pad_size, kernel_size, max_displacement, stride1, stride2)
        ctx.params = (pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)
        ctx.save_for_backward(input1, input2)
        if ctx.is81:
            return ops.corr81_forward_raw(input1, input2)
        return ops.correlation_forward_general(input1, input2, pad_size, kernel_size, max_displacement,
                                               stride1, stride2, corr_multiply)

    @staticmethod
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        if ctx.is81:
            g1, g2 = ops.corr81_backward_raw(input1, input2, grad_output.to(input1.dtype))
        else:                                           # (raises UpflowHipError for kernel_size > 1 / stride1 > 1)
            g1, g2 = ops.correlation_backward_general(input1, input2, grad_output.to(input1.dtype), *ctx.params)
        return g1, g2, None, None, None, None, None, None


class Correlation(Module):
    def __init__(self, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2, corr_multiply=1):
        super(Correlation, self).__init__()
        self.pad_size = pad_size
        self.kernel_size = kernel_size
        self.max_displacement = max_displacement
        self.stride1 = stride1
        self.stride2 = stride2
        self.corr_multiply = corr_multiply

    def forward(self, input1, input2):
        with torch.cuda.device_of(input1):
            return CorrelationFunction.apply(input1, input2, self.pad_size, self.kernel_size, self.max_displacement,
                                             self.stride1, self.stride2, self.corr_multiply)
