#!/usr/bin/env python3
"""One multitask-CTC training step of the encoder side of the LibriSpeech Conformer-SummaryMixing recipe on the HIP path:
waveform -> Fbank -> InputNormalization -> ConvolutionFrontEnd -> TransformerASR.encode -> proj_enc -> proj_ctc ->
log_softmax -> ctc_cost, optimised by the flat-buffer AdamW (needs an MI355X; synthetic data).

    python examples/encoder_ctc_step.py [--steps 20] [--batch 16] [--seconds 8]

The module names and constructor arguments are the ones the reference YAML passes (…/LibriSpeech/ASR/transducer/
hparams/conformer_summarymixing_transducer.yaml); only the import root changes: speechbrain.* -> summarymixing_amd.*."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd.lobes.features import Fbank, InputNormalization                     # noqa: E402
from summarymixing_amd.lobes.models.convolution import ConvolutionFrontEnd                   # noqa: E402
from summarymixing_amd.lobes.models.transformer.TransformerASR import EncoderWrapper, TransformerASR  # noqa: E402
from summarymixing_amd.nnet.activations import Softmax                                       # noqa: E402
from summarymixing_amd.nnet.linear import Linear                                             # noqa: E402
from summarymixing_amd.nnet.losses import ctc_loss                                           # noqa: E402
from summarymixing_amd.trainer import FlatAdamW                                              # noqa: E402


class EncoderSide(torch.nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, layers=12, joint_dim=640, vocab=1000):
        super().__init__()
        self.compute_features = Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=32)
        self.normalize = InputNormalization(norm_type="global", update_until_epoch=4)
        self.CNN = ConvolutionFrontEnd((None, None, 80), num_blocks=2, num_layers_per_block=1, out_channels=(64, 32),
                                       kernel_sizes=(3, 3), strides=(2, 2), residuals=(False, False))
        self.enc = EncoderWrapper(TransformerASR(
            tgt_vocab=vocab, input_size=640, d_model=d_model, nhead=4, num_encoder_layers=layers, num_decoder_layers=0,
            d_ffn=d_ffn, dropout=0.1, encoder_module="conformer", attention_type="SummaryMixing", mode="SummaryMixing-fast",
            local_proj_hid_dim=[d_model], local_proj_out_dim=d_model, summary_hid_dim=[d_model], summary_out_dim=d_model,
            causal=False))
        self.proj_enc = Linear(joint_dim, input_size=d_model)
        self.proj_ctc = Linear(vocab, input_size=joint_dim)
        self.log_softmax = Softmax(apply_log=True)

    def forward(self, wav, wav_lens, epoch=0):
        feats = self.normalize(self.compute_features(wav, out_dtype=torch.bfloat16), wav_lens, epoch=epoch)
        x = self.enc(self.CNN(feats), wav_lens)
        return self.log_softmax(self.proj_ctc(self.proj_enc(x)).float())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=8.0)
    args = ap.parse_args()
    torch.manual_seed(0)
    model = EncoderSide().cuda().train()
    opt = FlatAdamW(model, lr=8e-4, betas=(0.9, 0.98), weight_decay=0.01, max_grad_norm=5.0, compute_dtype=torch.bfloat16)
    B, L = args.batch, int(16000 * args.seconds)
    wav = torch.randn(B, L, device="cuda") * 0.1
    wav_lens = torch.linspace(1.0, 0.6, B, device="cuda")
    for b in range(B):
        wav[b, int(L * float(wav_lens[b])):] = 0.0
    tokens = torch.randint(1, 1000, (B, 40), device="cuda")
    tok_lens = torch.ones(B, device="cuda")
    t0 = None
    for step in range(args.steps):
        if step == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        opt.zero_grad()
        loss = ctc_loss(model(wav, wav_lens), tokens, wav_lens, tok_lens, blank_index=0)
        loss.backward()
        opt.step()
        if step % 5 == 0 or step == args.steps - 1:
            print(f"step {step:3d}  ctc loss {loss.item():9.4f}")
    torch.cuda.synchronize()
    if t0 is not None and args.steps > 3:
        dt = (time.perf_counter() - t0) / (args.steps - 3)
        frames = B * (1 + L // 160 + 3) // 4
        print(f"{dt*1e3:.1f} ms/step, {frames/dt:,.0f} encoder frames/s including front-end and CTC head")


if __name__ == "__main__":
    main()
