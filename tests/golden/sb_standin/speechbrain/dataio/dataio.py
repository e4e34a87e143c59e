"""Stand-in for speechbrain.dataio.dataio.length_to_mask."""
import torch


def length_to_mask(length, max_len=None, dtype=None, device=None):
    if max_len is None:
        max_len = length.max().long().item()
    mask = torch.arange(max_len, device=length.device, dtype=length.dtype).expand(len(length), max_len) < length.unsqueeze(1)
    if dtype is None:
        dtype = length.dtype
    if device is None:
        device = length.device
    return torch.as_tensor(mask, dtype=dtype, device=device)
