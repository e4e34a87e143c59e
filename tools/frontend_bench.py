#!/usr/bin/env python3
"""Front-end numbers at the bench batch (…transducer.yaml:171-175,247-254): waveform -> fbank -> InputNormalization -> CNN
(2 x Conv2d s2 + LayerNorm + LeakyReLU) -> (B, T/4, 640), B = 128 utterances x 20 s = the encoder's 128 x 500 frames.
Prints ms, frames/s and the share of the C2b encoder training step (pass its ms as argv[1], default 19.5)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_kernel                                                    # noqa: E402
from summarymixing_amd.lobes.features import Fbank, InputNormalization           # noqa: E402
from summarymixing_amd.lobes.models.convolution import ConvolutionFrontEnd      # noqa: E402

step_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 19.5
B, secs = (int(v) for v in os.environ.get("BS", "128,20").split(","))   # BS=10,15: the recipe's 150 s batch
wav = torch.randn(B, 16000 * secs, device="cuda") * 0.1
lens = torch.ones(B, device="cuda")
fb = Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=32).cuda()
norm = InputNormalization(norm_type="global", update_until_epoch=4).cuda()
cnn = ConvolutionFrontEnd((None, None, 80), dropout=0.0).cuda()


def features():
    return norm(fb(wav), lens, epoch=1)


feats = features()
enc_frames = B * (feats.shape[1] // 4)
t_fb = time_kernel(lambda: fb(wav), 10, 2)
t_ft = time_kernel(features, 10, 2)
x = feats.bfloat16()
with torch.no_grad():
    out = cnn(x)
    t_cf = time_kernel(lambda: cnn(x), 10, 2)


def fb_step():
    for p in cnn.parameters():
        p.grad = None
    y = cnn(x)
    y.backward(torch.ones_like(y))


t_cb = time_kernel(fb_step, 5, 2)


def whole_train():
    f = features().bfloat16()
    for p in cnn.parameters():
        p.grad = None
    y = cnn(f)
    y.backward(torch.ones_like(y))


t_all = time_kernel(whole_train, 5, 2)
print(f"# front-end at B = {B} x {secs} s of 16 kHz audio -> features {tuple(feats.shape)} -> CNN output {tuple(out.shape)} "
      f"({enc_frames} encoder frames); C2b encoder training step = {step_ms:.2f} ms")
for name, t in (("fbank (STFT -> mel -> dB)", t_fb), ("fbank + InputNormalization (global, training)", t_ft),
                ("CNN forward (bf16)", t_cf), ("CNN forward + backward (bf16)", t_cb),
                ("waveform -> (B,T/4,640), forward + CNN backward (a training step's front-end)", t_all)):
    print(f"{name:82s} {t * 1e3:7.2f} ms  {enc_frames / t / 1e6:7.2f} M encoder frames/s  {100 * t * 1e3 / step_ms:5.1f} % of the encoder step")
