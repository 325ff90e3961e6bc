"""The reference's native FFI, literally: `correlation_cuda.forward / backward` with the pybind
signatures of `/root/reference/model/correlation_package/correlation_cuda.cc:10-17, 89-97, 169-172`.

    forward (input1, input2, rInput1, rInput2, output, pad_size, kernel_size, max_displacement,
             stride1, stride2, corr_type_multiply) -> 1
    backward(input1, input2, rInput1, rInput2, gradOutput, gradInput1, gradInput2, <same 6 ints>) -> 1

Ownership as in the reference: the caller passes EMPTY tensors of the input's type/device
(`input1.new()`, correlation.py:22-24); the callee `resize_`s them and writes in place
(correlation_cuda.cc:36-42).  rInput1/rInput2 were the padded-NHWC scratch copies of the CUDA
implementation; this implementation reads NCHW directly and leaves them empty (size 0).
Failure raises RuntimeError (the reference: AT_ERROR("CUDA call failed"), .cc:81-83).
"""
from . import ops


def forward(input1, input2, rInput1, rInput2, output, pad_size, kernel_size, max_displacement, stride1, stride2,
            corr_type_multiply):
    a, b = input1.contiguous(), input2.contiguous()
    B, C, H, W = a.shape
    oc, oh, ow = ops.correlation_out_shape(H, W, pad_size, kernel_size, max_displacement, stride1, stride2)
    output.resize_(B, oc, oh, ow)                        # correlation_cuda.cc:36-42: resized, then written in place
    if output.dtype != a.dtype or output.device != a.device:
        raise ops.UpflowHipError('correlation_cuda.forward: output must have the inputs\' type and device (input1.new(), correlation.py:22-24)')
    ops.correlation_forward_general(a, b, pad_size, kernel_size, max_displacement, stride1, stride2, corr_type_multiply, out=output)
    return 1


def backward(input1, input2, rInput1, rInput2, gradOutput, gradInput1, gradInput2, pad_size, kernel_size,
             max_displacement, stride1, stride2, corr_type_multiply):
    a, b = input1.contiguous(), input2.contiguous()
    gradInput1.resize_(a.shape)                          # correlation_cuda.cc:113-117: resized, then written in place
    gradInput2.resize_(b.shape)
    if gradInput1.dtype != a.dtype or gradInput2.dtype != a.dtype or gradInput1.device != a.device or gradInput2.device != a.device:
        raise ops.UpflowHipError('correlation_cuda.backward: gradInput1 / gradInput2 must have the inputs\' type and device')
    ops.correlation_backward_general(a, b, gradOutput.to(a.dtype), pad_size, kernel_size, max_displacement, stride1, stride2,
                                     corr_type_multiply, g1=gradInput1, g2=gradInput2)
    return 1
