"""CPU sanity of the front-end spec in oracle/smx_oracle.py (parity unpinned: upstream SpeechBrain arithmetic)."""
import math

import torch

from oracle import smx_oracle as O


def test_mel_filterbank_is_triangular_and_covers_the_band():
    fb = O.mel_filterbank(80, 512, 16000)
    assert fb.shape == (80, 257) and (fb >= 0).all() and fb.max() <= 1.0 + 1e-6
    peaks = fb.argmax(dim=1)
    assert (peaks[1:] >= peaks[:-1]).all()                       # centre frequencies increase
    assert (fb.sum(0)[2:250] > 0).all()                           # no uncovered bin inside the band


def test_fbank_sine_peaks_in_the_right_mel_bin_and_respects_top_db():
    sr, f0 = 16000, 1000.0
    t = torch.arange(sr) / sr
    wav = torch.sin(2 * math.pi * f0 * t)[None].double()
    db = O.fbank(wav, n_fft=512, win_length_ms=32, n_mels=80)
    assert db.shape == (1, 101, 80)
    fbm = O.mel_filterbank(80, 512, sr)
    expect = int(fbm[:, round(f0 / (sr / 2) * 256)].argmax())
    assert int(db[0, 50].argmax()) == expect
    assert float(db.max() - db.min()) <= 80.0 + 1e-6


def test_conv_frontend_shapes_and_grad():
    torch.manual_seed(0)
    sd = {"convblock_0.conv.weight": torch.randn(64, 1, 3, 3) * 0.2, "convblock_0.conv.bias": torch.zeros(64),
          "convblock_0.norm.weight": torch.ones(40, 64), "convblock_0.norm.bias": torch.zeros(40, 64),
          "convblock_1.conv.weight": torch.randn(32, 64, 3, 3) * 0.05, "convblock_1.conv.bias": torch.zeros(32),
          "convblock_1.norm.weight": torch.ones(20, 32), "convblock_1.norm.bias": torch.zeros(20, 32)}
    x = torch.randn(2, 41, 80, requires_grad=True)
    y = O.conv_frontend(x, sd)
    assert y.shape == (2, 11, 640)
    y.sum().backward()
    assert torch.isfinite(x.grad).all()
