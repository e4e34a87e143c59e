#!/usr/bin/env python3
"""Time the front-end (fbank, conv subsampling fwd / fwd+bwd) at the bench batch: 64 utterances x 20 s."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel
from summarymixing_amd.lobes.features import Fbank
from summarymixing_amd.lobes.models.convolution import ConvolutionFrontEnd
B, secs = 64, 20
wav = torch.randn(B, 16000 * secs, device="cuda") * 0.1
fb = Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=32).cuda()
feats = fb(wav)
t = time_kernel(lambda: fb(wav), 10, 2)
print(f"fbank   ({B} x {secs} s -> {tuple(feats.shape)}): {t*1e3:7.2f} ms   {B*secs/t:10.0f} x real time")
cnn = ConvolutionFrontEnd((None, None, 80), dropout=0.0).cuda()
x = feats.bfloat16()
with torch.no_grad():
    t = time_kernel(lambda: cnn(x), 10, 2)
print(f"conv fwd (bf16) -> {tuple(cnn(x).shape)}: {t*1e3:7.2f} ms")
def fb_step():
    for p in cnn.parameters(): p.grad = None
    y = cnn(x); y.backward(torch.ones_like(y))
t = time_kernel(fb_step, 5, 2)
print(f"conv fwd+bwd (bf16): {t*1e3:7.2f} ms")
