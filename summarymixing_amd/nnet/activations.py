"""Activation markers.  The kernels fuse the activation into GEMM / LayerNorm epilogues, so these classes
only name the function (constructor-compatible with the classes the reference recipes pass)."""
import torch

from .. import _lib as L


class Swish(torch.nn.Module):
    """x * sigmoid(x) (speechbrain.nnet.activations.Swish, beta = 1)."""

    def __init__(self, beta=1.0):
        super().__init__()
        if beta != 1.0:
            raise NotImplementedError("only beta=1 Swish is fused in the kernels")

    def forward(self, x):  # pragma: no cover - never on the hot path
        raise RuntimeError("activations are fused into the HIP kernels; this module is a marker")


class _LogSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from .. import ops
        x2 = ops.rows2d(x if x.is_contiguous() else x.contiguous())
        y = ops.log_softmax_fwd(x2)
        ctx.save_for_backward(y)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        from .. import ops
        (y,) = ctx.saved_tensors
        dy2 = ops.rows2d(dy if dy.is_contiguous() else dy.contiguous())
        return ops.log_softmax_bwd(dy2, y).view(dy.shape)


class Softmax(torch.nn.Module):
    """speechbrain.nnet.activations.Softmax as the recipes instantiate it (``apply_log: True``, last dim): the
    log-softmax in front of the CTC loss (…transducer.yaml:331).  HIP kernels, GPU only."""

    def __init__(self, apply_log=False, dim=-1, reshape=True, dtype=torch.float32):
        super().__init__()
        if not apply_log or dim != -1:
            raise NotImplementedError("only the log-softmax over the last dim used by the CTC head is implemented")

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("summarymixing_amd kernels run on the GPU only (no CPU fallback)")
        return _LogSoftmax.apply(x)


_BY_NAME = {"gelu": L.ACT_GELU, "swish": L.ACT_SWISH, "silu": L.ACT_SWISH, "leakyrelu": L.ACT_LEAKY_RELU,
            "leaky_relu": L.ACT_LEAKY_RELU, "relu": L.ACT_RELU, "identity": L.ACT_NONE, "none": L.ACT_NONE}


def act_code(activation):
    """class / instance / name -> SMX_ACT_* code.  Raises for anything the kernels do not implement."""
    if isinstance(activation, int):
        return activation
    if isinstance(activation, str):
        name = activation
    else:
        inst = activation() if isinstance(activation, type) else activation
        name = type(inst).__name__
        if isinstance(inst, torch.nn.GELU) and getattr(inst, "approximate", "none") != "none":
            raise NotImplementedError("only exact (erf) GELU is implemented")
        if isinstance(inst, torch.nn.LeakyReLU) and inst.negative_slope != 0.01:
            raise NotImplementedError("only LeakyReLU(0.01) is implemented")
    code = _BY_NAME.get(name.lower())
    if code is None:
        raise NotImplementedError(f"activation {name} is not implemented in the SummaryMixing kernels")
    return code
