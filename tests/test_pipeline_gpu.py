"""The whole encoder side of a recipe step on the HIP path - waveform -> Fbank -> InputNormalization ->
ConvolutionFrontEnd -> TransformerASR.encode (Conformer-SummaryMixing) -> proj_enc -> proj_ctc -> log_softmax -> ctc_cost -
against the CPU oracle chain on the same weights (fp32): loss and gradients of parameters at both ends of the chain."""
import pytest
import torch

from oracle import smx_oracle as O
from tests._util import rel_err

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("ln_fuse_mode")]   # (both LayerNorm dispatches: tests/conftest.py)


def test_waveform_to_ctc_loss_matches_oracle_chain():
    from summarymixing_amd.lobes.features import Fbank, InputNormalization
    from summarymixing_amd.lobes.models.convolution import ConvolutionFrontEnd
    from summarymixing_amd.lobes.models.transformer.TransformerASR import EncoderWrapper, TransformerASR
    from summarymixing_amd.nnet.activations import Softmax
    from summarymixing_amd.nnet.linear import Linear
    from summarymixing_amd.nnet.losses import ctc_loss
    torch.manual_seed(31)
    B, Lw, d, J, V, S = 2, 16000, 64, 96, 30, 5
    wav = torch.randn(B, Lw) * 0.1
    wav[1, 11200:] = 0.0                                           # second utterance is 70 % long (zero padded)
    wav_len = torch.tensor([1.0, 0.7])
    targets = torch.randint(1, V, (B, S))
    tg_rel = torch.tensor([1.0, 0.6])
    fbank = Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=32).cuda()
    norm = InputNormalization(norm_type="global").cuda().train()
    cnn = ConvolutionFrontEnd((None, None, 80), out_channels=(64, 32), dropout=0.0).cuda()
    net = TransformerASR(tgt_vocab=V, input_size=640, d_model=d, nhead=4, num_encoder_layers=2, num_decoder_layers=0,
                         d_ffn=128, dropout=0.0, encoder_module="conformer", conformer_activation="swish",
                         attention_type="SummaryMixing", mode="SummaryMixing-fast", local_proj_out_dim=d,
                         local_proj_hid_dim=[d], summary_hid_dim=[d], summary_out_dim=d, causal=False, kernel_size=15)
    enc = EncoderWrapper(net).cuda()
    proj_enc, proj_ctc = Linear(J, input_size=d).cuda(), Linear(V, input_size=J).cuda()
    # ---- HIP path ----
    feats = norm(fbank(wav.cuda()), wav_len.cuda())                # (B, 101, 80)
    x = enc(cnn(feats), wav_len.cuda())                            # (B, 26, 20, 32) -> (B, 26, d)
    lp = Softmax(apply_log=True)(proj_ctc(proj_enc(x)))
    loss = ctc_loss(lp, targets.cuda(), wav_len.cuda(), tg_rel.cuda(), 0)
    loss.backward()
    # ---- oracle chain ----
    sd_cnn = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in cnn.state_dict_for_oracle().items()}
    sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point())
          for k, v in enc.transformer.state_dict().items() if k != "positional_encoding.pe"}
    We, be = (p.detach().cpu().clone().requires_grad_(True) for p in (proj_enc.w.weight, proj_enc.w.bias))
    Wc, bc = (p.detach().cpu().clone().requires_grad_(True) for p in (proj_ctc.w.weight, proj_ctc.w.bias))
    fo = O.input_normalization(O.fbank(wav), wav_len, O.InputNormalizationState(), norm_type="global")
    assert rel_err(feats, fo) <= 2e-3
    xo = O.asr_encode(O.conv_frontend(fo, sd_cnn), wav_len, sd, "conformer", "swish", "SummaryMixing-fast", d)
    assert x.shape == xo.shape == (B, 26, d)
    lo = O.ctc_loss(O.log_softmax((xo @ We.t() + be) @ Wc.t() + bc), targets, wav_len, tg_rel, 0)
    lo.backward()
    assert rel_err(x, xo) <= 5e-3
    assert abs(loss.item() - lo.item()) <= 5e-3 * abs(lo.item()), (loss.item(), lo.item())
    fro = lambda a, b: float((a.detach().double().cpu() - b.detach().double()).norm() / b.detach().double().norm())
    assert fro(proj_ctc.w.weight.grad, Wc.grad) <= 1e-2 and fro(proj_enc.w.bias.grad, be.grad) <= 1e-2
    params = dict(enc.transformer.named_parameters())
    for k in ("custom_src_module.layers.0.w.weight", "encoder.layers.1.ffn_module2.1.ffn.0.weight", "encoder.norm.norm.weight"):
        assert fro(params[k].grad, sd[k].grad) <= 2e-2, k
    assert fro(cnn.blocks[1].conv.weight.grad, sd_cnn["convblock_1.conv.weight"].grad) <= 2e-2
    assert fro(cnn.blocks[0].norm.weight.grad, sd_cnn["convblock_0.norm.weight"].grad) <= 2e-2


def test_training_steps_reduce_the_ctc_loss():
    """40 optimizer steps (FlatAdamW: HIP clip + AdamW) on one fixed batch, bf16 compute, dropout on: the multitask CTC
    loss of the encoder + heads must go down substantially and every parameter must stay finite."""
    from summarymixing_amd.lobes.models.transformer.TransformerASR import EncoderWrapper, TransformerASR
    from summarymixing_amd.nnet.activations import Softmax
    from summarymixing_amd.nnet.linear import Linear
    from summarymixing_amd.nnet.losses import ctc_loss
    from summarymixing_amd.trainer import FlatAdamW
    torch.manual_seed(5)
    B, T, Fin, d, V, S = 8, 64, 80, 64, 20, 6

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            net = TransformerASR(tgt_vocab=V, input_size=Fin, d_model=d, nhead=4, num_encoder_layers=2, num_decoder_layers=0,
                                 d_ffn=128, dropout=0.1, encoder_module="conformer", conformer_activation="swish",
                                 attention_type="SummaryMixing", mode="SummaryMixing-fast", local_proj_out_dim=d,
                                 local_proj_hid_dim=[d], summary_hid_dim=[d], summary_out_dim=d, causal=False, kernel_size=15)
            self.enc = EncoderWrapper(net)
            self.proj_ctc = Linear(V, input_size=d)
    m = Model().cuda().train()
    opt = FlatAdamW(m, lr=2e-3, compute_dtype=torch.bfloat16)
    src = torch.randn(B, T, Fin).cuda().bfloat16()
    wav_len = torch.linspace(1.0, 0.6, B).cuda()
    targets = torch.randint(1, V, (B, S)).cuda()
    tg_rel = torch.ones(B).cuda()
    log_softmax = Softmax(apply_log=True)
    losses = []
    for _ in range(40):
        opt.zero_grad()
        loss = ctc_loss(log_softmax(m.proj_ctc(m.enc(src, wav_len)).float()), targets, wav_len, tg_rel, 0)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(torch.isfinite(torch.tensor(losses)))
    assert torch.isfinite(opt.flat_p).all()
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])


def test_gradient_accumulation_over_two_backward_passes():
    """Parameter gradients accumulate across backward passes (kernels add into param.grad; the deferred reductions are
    flushed per block): two passes on the same batch give twice the gradient of one."""
    from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoder
    torch.manual_seed(2)
    d = 64
    enc = ConformerEncoder(2, d, 128, 4, kernel_size=15, activation="swish", dropout=0.0, attention_type="SummaryMixing",
                           local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d],
                           mode="SummaryMixing-fast").cuda()
    x = torch.randn(3, 50, d, device="cuda")
    pad = (torch.arange(50, device="cuda")[None] < torch.tensor([50, 31, 44], device="cuda")[:, None])
    r = torch.randn(3, 50, d, device="cuda")

    def run():
        y, _ = enc(x, src_key_padding_mask=pad)
        (y * r).sum().backward()
    run()
    g1 = {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
    run()
    assert len(g1) > 20
    for n, p in enc.named_parameters():
        if n in g1:
            assert rel_err(p.grad, 2.0 * g1[n]) <= 1e-5, n
