"""Stand-in for speechbrain.nnet.attention: only PositionalwiseFeedForward is functional."""
import torch
from torch import nn


class PositionalwiseFeedForward(nn.Module):
    def __init__(self, d_ffn, input_shape=None, input_size=None, dropout=0.0, activation=nn.ReLU):
        super().__init__()
        if input_size is None:
            input_size = input_shape[-1]
        self.ffn = nn.Sequential(
            nn.Linear(input_size, d_ffn), activation(), nn.Dropout(dropout), nn.Linear(d_ffn, input_size)
        )

    def forward(self, x):
        x = x.permute(1, 0, 2)
        x = self.ffn(x)
        return x.permute(1, 0, 2)


class _Unsupported(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("attention classes are import-time only on the SummaryMixing path")


class MultiheadAttention(_Unsupported):
    pass


class RelPosMHAXL(_Unsupported):
    pass


class RelPosEncXL(_Unsupported):
    pass
