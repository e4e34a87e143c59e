"""Log-mel filterbank front-end on MI355X (surface of speechbrain.lobes.features.Fbank as the recipes use it:
``Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=32)``, ...transducer.yaml:171-175).

frame+window kernel -> DFT as an exact-fp32 MFMA GEMM against a [cos | -sin] basis -> fused power/mel/dB kernel ->
per-utterance top_db clamp.  No parameters, no backward.  Arithmetic spec: oracle/smx_oracle.py::fbank (the reference
pins nothing here: upstream SpeechBrain code)."""
import math

import torch
from torch import nn

from .. import _lib as L
from .. import ops


class Fbank(nn.Module):
    def __init__(self, deltas=False, context=False, requires_grad=False, sample_rate=16000, f_min=0, f_max=None, n_fft=400,
                 n_mels=40, filter_shape="triangular", param_change_factor=1.0, param_rand_factor=0.0, left_frames=5,
                 right_frames=5, win_length=25, hop_length=10, amin=1e-10, top_db=80.0):
        super().__init__()
        if deltas or context or requires_grad or filter_shape != "triangular":
            raise NotImplementedError("only the plain triangular log-mel filterbank of the SummaryMixing recipes is built")
        self.sample_rate, self.n_fft, self.n_mels = sample_rate, n_fft, n_mels
        self.win = int(round(sample_rate / 1000.0 * win_length))
        self.hop = int(round(sample_rate / 1000.0 * hop_length))
        if self.win != n_fft:
            raise NotImplementedError("win_length must equal n_fft samples (recipe: 32 ms @ 16 kHz = 512)")
        self.amin, self.top_db = amin, top_db
        n_bins = n_fft // 2 + 1
        self.im_off = (n_bins + 3) // 4 * 4                                # 260 for 257 bins: 16-byte aligned halves
        k = torch.arange(n_bins, dtype=torch.float64)[:, None] * torch.arange(n_fft, dtype=torch.float64)[None, :]
        ang = 2.0 * math.pi * k / n_fft
        basis = torch.zeros(2 * self.im_off, n_fft, dtype=torch.float64)
        basis[:n_bins] = torch.cos(ang)
        basis[self.im_off:self.im_off + n_bins] = -torch.sin(ang)
        self.register_buffer("basis", basis.float(), persistent=False)
        self.register_buffer("window", torch.hamming_window(self.win, dtype=torch.float32), persistent=False)
        # HTK-mel triangular filters (n_mels, n_bins)
        f_max = sample_rate / 2 if f_max is None else f_max
        to_mel = lambda hz: 2595.0 * math.log10(1.0 + hz / 700.0)
        mel = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2, dtype=torch.float64)
        hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
        band = (hz[1:] - hz[:-1])[:-1]
        freqs = torch.linspace(0, sample_rate // 2, n_bins, dtype=torch.float64)
        slope = (freqs[None, :] - hz[1:-1][:, None]) / band[:, None]
        self.register_buffer("fb", torch.clamp(torch.minimum(slope + 1.0, -slope + 1.0), min=0.0).float().contiguous(),
                             persistent=False)

    def forward(self, wav, out_dtype=torch.float32):
        """wav (B, L) float32 on the GPU -> (B, 1 + L // hop, n_mels)."""
        assert wav.dim() == 2 and wav.dtype == torch.float32 and wav.is_cuda
        B, Lw = wav.shape
        T = 1 + Lw // self.hop
        frames = ops.frame_window(wav.contiguous(), self.window, T, self.n_fft, self.hop)
        spec = torch.empty((B * T, self.basis.shape[0]), dtype=torch.float32, device=wav.device)
        ops.gemm(L.GEMM_NT, frames, self.basis, spec, B * T, self.basis.shape[0], self.n_fft)
        return ops.mel_db(spec, self.im_off, self.fb, B, T, self.amin, self.top_db, out_dtype)
