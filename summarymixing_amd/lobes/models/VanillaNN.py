"""ParallelLinear / VanillaNN with the reference's constructor signatures and state-dict keys
(reference: speechbrain/lobes/models/VanillaNN.py:26-196), running on the libsmx.so GEMM.

State-dict keys: ``linear.w.weight|bias`` (n_split == 1) or ``linear.weights|biases`` (n_split > 1), then
``linear_0``, ``linear_1`` ... for further blocks, exactly as the reference's Sequential names them.
"""
import math
from typing import Optional

import torch
from torch import nn

from ... import _lib as L
from ... import functional as F
from ... import ops
from ...nnet.activations import act_code


class Linear(nn.Module):
    """Parameter holder with the key layout of speechbrain.nnet.linear.Linear (``.w`` is an nn.Linear)."""

    def __init__(self, n_neurons, input_size, bias=True):
        super().__init__()
        self.w = nn.Linear(input_size, n_neurons, bias=bias)


class ParallelLinear(nn.Module):
    """y[..., m, :] = x[..., m, :] @ weights[m] + biases[m] for n_split heads (VanillaNN.py:58-117)."""

    def __init__(self, n_neurons, input_shape: Optional[list] = None, input_size: Optional[int] = None,
                 n_split: Optional[int] = 1, bias: Optional[bool] = True, combine_out_dims: Optional[bool] = True):
        super().__init__()
        if input_shape is None and input_size is None:
            raise ValueError("Expected one of input_shape or input_size")
        if input_size is None:
            input_size = input_shape[-1]
            if len(input_shape) == 4:
                input_size = input_shape[-1] * input_shape[-2]
        if input_size % n_split != 0 or n_neurons % n_split != 0:
            raise ValueError("input_size and n_neurons must be dividible by n_split!")
        self.n_split, self.combine_out_dims = n_split, combine_out_dims
        self.split_inp_dim, self.split_out_dim = input_size // n_split, n_neurons // n_split
        self.weights = nn.Parameter(torch.empty(n_split, self.split_inp_dim, self.split_out_dim))
        self.biases = nn.Parameter(torch.zeros(n_split, self.split_out_dim))
        # same init as the reference (:92-97): kaiming-uniform on weights AND biases
        nn.init.kaiming_uniform_(self.weights, a=math.sqrt(5))
        nn.init.kaiming_uniform_(self.biases, a=math.sqrt(5))

    def spec(self):
        return {"kind": "parallel", "W": self.weights, "b": self.biases}

    def forward(self, x):
        lead = x.shape[:2]
        x3 = x.reshape(lead[0], lead[1], -1)
        layers = [self.spec()]

        def run(xin, need_bwd):
            x2 = ops.rows2d(xin)
            y, saved = F.mlp_fwd(x2, layers, L.ACT_NONE, None, need_bwd, xin.dtype)

            def bwd(dy):
                dy2 = ops.rows2d(dy.contiguous().view(lead[0], lead[1], -1))
                return F.mlp_bwd(dy2, layers, L.ACT_NONE, saved, xin.dtype).view(xin.shape)
            y3 = y.view(lead[0], lead[1], -1)
            if not self.combine_out_dims:
                y3 = y3.view(lead[0], lead[1], self.n_split, self.split_out_dim)
            return y3, (bwd if need_bwd else None)
        return F.block(x3, run, [self.weights, self.biases])


class VanillaNN(nn.Module):
    """[Linear | ParallelLinear -> activation] x dnn_blocks (VanillaNN.py:153-196)."""

    def __init__(self, input_shape, activation: Optional[nn.Module] = torch.nn.LeakyReLU,
                 dnn_blocks: Optional[int] = 2, dnn_neurons: Optional[int] = 512, n_split: Optional[int] = 1):
        super().__init__()
        if isinstance(dnn_neurons, list) and len(dnn_neurons) != dnn_blocks:
            raise ValueError("The length of the dnn_neurons list must match dnn_blocks...")
        self.act = act_code(activation)
        self.n_split = n_split
        in_size = input_shape[-1]
        self._names = []
        for i in range(dnn_blocks):
            n = dnn_neurons[i] if isinstance(dnn_neurons, list) else dnn_neurons
            name = "linear" if i == 0 else f"linear_{i - 1}"
            if n_split > 1:
                mod = ParallelLinear(n, input_size=in_size, n_split=n_split, combine_out_dims=(i == dnn_blocks - 1))
            else:
                mod = Linear(n, in_size)
            self.add_module(name, mod)
            self._names.append(name)
            in_size = n

    def specs(self):
        out = []
        for name in self._names:
            m = getattr(self, name)
            out.append(m.spec() if isinstance(m, ParallelLinear) else {"kind": "linear", "W": m.w.weight, "b": m.w.bias})
        return out

    def forward(self, x):
        B, T = x.shape[0], x.shape[1]
        layers, act = self.specs(), self.act

        def run(xin, need_bwd):
            y, saved = F.mlp_fwd(ops.rows2d(xin), layers, act, None, need_bwd, xin.dtype)

            def bwd(dy):
                return F.mlp_bwd(ops.rows2d(dy.contiguous()), layers, act, saved, xin.dtype).view(xin.shape)
            return y.view(B, T, -1), (bwd if need_bwd else None)
        return F.block(x, run, list(self.parameters()))
