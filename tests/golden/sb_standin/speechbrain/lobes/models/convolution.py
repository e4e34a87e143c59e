"""Stand-in for speechbrain.lobes.models.convolution.ConvolutionalSpatialGatingUnit."""
import torch
from speechbrain.nnet.CNN import Conv1d
from speechbrain.nnet.normalization import LayerNorm


class ConvolutionalSpatialGatingUnit(torch.nn.Module):
    def __init__(self, input_size, kernel_size=31, dropout=0.0, use_linear_after_conv=False,
                 activation=torch.nn.Identity):
        super().__init__()
        if input_size % 2 != 0:
            raise ValueError("Input size must be divisible by 2!")
        n = input_size // 2
        self.norm = LayerNorm(n)
        self.conv = Conv1d(input_shape=(None, None, n), out_channels=n, kernel_size=kernel_size, stride=1,
                           padding="same", groups=n, conv_init="normal", skip_transpose=False)
        self.linear = None
        if use_linear_after_conv:
            self.linear = torch.nn.Linear(n, n)
            torch.nn.init.normal_(self.linear.weight, std=1e-6)
            torch.nn.init.ones_(self.linear.bias)
        torch.nn.init.ones_(self.conv.conv.bias)
        self.activation = activation()
        self.dropout = torch.nn.Dropout(dropout)

    def forward(self, x):
        x1, x2 = x.chunk(2, dim=-1)
        x2 = self.conv(self.norm(x2))
        if self.linear is not None:
            x2 = self.linear(x2)
        x2 = self.activation(x2)
        return self.dropout(x2 * x1)
