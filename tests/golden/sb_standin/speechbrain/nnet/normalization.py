"""Stand-in for speechbrain.nnet.normalization.LayerNorm (torch LayerNorm held as .norm)."""
import torch


class LayerNorm(torch.nn.Module):
    def __init__(self, input_size=None, input_shape=None, eps=1e-05, elementwise_affine=True):
        super().__init__()
        if input_shape is not None:
            input_size = input_shape[2:]
        self.norm = torch.nn.LayerNorm(input_size, eps=eps, elementwise_affine=elementwise_affine)

    def forward(self, x):
        return self.norm(x)
