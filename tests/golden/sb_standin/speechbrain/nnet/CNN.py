"""Stand-in for speechbrain.nnet.CNN.Conv1d (channels-last, 'same' padding, reflect mode, groups)."""
import torch
import torch.nn.functional as F


class Conv1d(torch.nn.Module):
    def __init__(self, out_channels, kernel_size, input_shape=None, in_channels=None, stride=1, dilation=1,
                 padding="same", groups=1, bias=True, padding_mode="reflect", skip_transpose=False,
                 conv_init=None):
        super().__init__()
        if in_channels is None:
            in_channels = input_shape[-1]
        self.kernel_size, self.padding, self.padding_mode = kernel_size, padding, padding_mode
        self.skip_transpose = skip_transpose
        self.conv = torch.nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, dilation=dilation,
                                    padding=0, groups=groups, bias=bias)
        if conv_init == "normal":
            torch.nn.init.normal_(self.conv.weight, std=1e-6)

    def forward(self, x):
        if not self.skip_transpose:
            x = x.transpose(1, -1)
        if self.padding == "same":
            p = (self.kernel_size - 1) // 2
            x = F.pad(x, (p, p), mode=self.padding_mode)
        x = self.conv(x)
        if not self.skip_transpose:
            x = x.transpose(1, -1)
        return x
