"""speechbrain.nnet.linear.Linear as the recipes instantiate it for the heads around the encoder (``proj_enc``,
``proj_ctc``, …transducer.yaml:280-288): ``y = x W^T + b`` over the last dimension, parameters under ``.w`` (an
``nn.Linear`` holder, so reference checkpoints load unchanged).  Forward, dgrad, wgrad and the bias gradient run on the
MFMA GEMM of libsmx.so; GPU only."""
from typing import Optional

from torch import nn

from .. import functional as F
from .. import ops


class Linear(nn.Module):
    def __init__(self, n_neurons, input_shape: Optional[list] = None, input_size: Optional[int] = None, bias=True,
                 max_norm=None, combine_dims=False):
        super().__init__()
        if input_shape is None and input_size is None:
            raise ValueError("Expected one of input_shape or input_size")
        if max_norm is not None:
            raise NotImplementedError("max_norm weight renormalisation is not used by the SummaryMixing recipes")
        self.combine_dims = combine_dims
        if input_size is None:
            input_size = input_shape[-1]
            if len(input_shape) == 4 and combine_dims:
                input_size = input_shape[2] * input_shape[3]
        self.w = nn.Linear(input_size, n_neurons, bias=bias)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("summarymixing_amd kernels run on the GPU only (no CPU fallback)")
        if x.dim() == 4 and self.combine_dims:
            x = x.reshape(x.shape[0], x.shape[1], x.shape[2] * x.shape[3])
        lead = x.shape[:-1]
        W, b = self.w.weight, self.w.bias

        def run(xin, need_bwd):
            x2 = xin.reshape(-1, xin.shape[-1])
            if not x2.is_contiguous():
                x2 = x2.contiguous()
            Wc = F.wcast(W, xin.dtype)
            y, _ = F.linear_fwd(x2, Wc, b)
            if not need_bwd:
                return y.view(*lead, -1), None

            def bwd(dy):
                dy2 = dy.reshape(-1, dy.shape[-1])
                if not dy2.is_contiguous():
                    dy2 = dy2.contiguous()
                dx, _ = F.linear_bwd(dy2, x2, Wc, None, ops.L.ACT_NONE, None, 1.0, F.gacc(W), F.gacc(b),
                                     need_dx=xin.requires_grad)
                return dx.view(xin.shape) if dx is not None else None
            return y.view(*lead, -1), bwd
        return F.block(x, run, [W] + ([b] if b is not None else []))
