"""CTC head: log-softmax + CTC loss / gradient kernels against torch's CPU implementation (what
speechbrain.nnet.losses.ctc_loss executes) through the oracle restatement of the SpeechBrain wrapper."""
import pytest
import torch

from oracle import smx_oracle as O
from tests._util import rel_err

pytestmark = pytest.mark.gpu


def _case(B, T, V, S, seed, full_first=True):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, T, V, generator=g) * 2.0
    targets = torch.randint(1, V, (B, S), generator=g)
    targets[0, 1:3] = targets[0, 0]                       # repeated labels (need a blank in between)
    in_rel = 0.6 + 0.4 * torch.rand(B, generator=g)
    tg_rel = 0.3 + 0.7 * torch.rand(B, generator=g)
    if full_first:
        in_rel[0], tg_rel[0] = 1.0, 1.0
    return logits, targets, in_rel, tg_rel


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("B,T,V,S", [(3, 50, 32, 9), (6, 200, 1000, 40), (2, 31, 17, 1), (4, 120, 5000, 70)])
@pytest.mark.parametrize("reduction", ["mean", "batchmean"])
def test_ctc_loss_and_gradient_match_torch(B, T, V, S, dtype, tol, reduction):
    from summarymixing_amd.nnet.activations import Softmax
    from summarymixing_amd.nnet.losses import ctc_loss
    logits, targets, in_rel, tg_rel = _case(B, T, V, S, 100 + B + T)
    x_ref = logits.to(dtype).float().clone().requires_grad_(True)   # same rounded inputs on both sides
    ref = O.ctc_loss(O.log_softmax(x_ref), targets, in_rel, tg_rel, 0, reduction)
    ref.backward()
    x = logits.to(dtype).cuda().detach().clone().requires_grad_(True)
    lp = Softmax(apply_log=True)(x)
    loss = ctc_loss(lp, targets.cuda(), in_rel.cuda(), tg_rel.cuda(), 0, reduction)
    loss.backward()
    assert abs(loss.item() - ref.item()) <= tol * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    assert rel_err(x.grad.float().cpu(), x_ref.grad) <= (5e-4 if dtype == torch.float32 else 3e-2)
    # frames beyond the input length carry no gradient
    in_len = (in_rel * T).round().int()
    for b in range(B):
        assert float(x.grad[b, in_len[b]:].abs().max() if in_len[b] < T else 0.0) == 0.0


def test_ctc_zero_infinity_and_log_softmax():
    from summarymixing_amd.nnet.activations import Softmax
    from summarymixing_amd.nnet.losses import ctc_loss
    torch.manual_seed(3)
    # utterance 1 cannot be aligned: 4 frames for 5 labels -> loss 0, gradient 0 (zero_infinity=True)
    logits = torch.randn(2, 10, 12)
    targets = torch.tensor([[1, 2, 3, 4, 5], [1, 1, 2, 2, 3]])
    in_rel, tg_rel = torch.tensor([1.0, 0.4]), torch.tensor([0.6, 1.0])
    xr = logits.clone().requires_grad_(True)
    ref = O.ctc_loss(O.log_softmax(xr), targets, in_rel, tg_rel, 0, "none")
    ref.sum().backward()
    x = logits.cuda().requires_grad_(True)
    lp = Softmax(apply_log=True)(x)
    assert rel_err(lp.detach().cpu(), O.log_softmax(logits)) <= 1e-6
    out = ctc_loss(lp, targets.cuda(), in_rel.cuda(), tg_rel.cuda(), 0, "none")
    out.sum().backward()
    assert out[1].item() == 0.0 and ref[1].item() == 0.0
    assert abs(out[0].item() - ref[0].item()) <= 1e-4 * abs(ref[0].item())
    assert float(x.grad[1].abs().max()) == 0.0
    assert rel_err(x.grad.cpu(), xr.grad) <= 5e-4
    # generic log-softmax backward (a dense upstream gradient)
    w = torch.randn(2, 10, 12)
    x2 = logits.cuda().requires_grad_(True)
    (Softmax(apply_log=True)(x2) * w.cuda()).sum().backward()
    x3 = logits.clone().requires_grad_(True)
    (torch.log_softmax(x3, -1) * w).sum().backward()
    assert rel_err(x2.grad.cpu(), x3.grad) <= 1e-5


def test_encoder_plus_ctc_head_step_matches_oracle():
    """A 'real' multitask step of the recipes' encoder side: EncoderWrapper -> proj_enc -> proj_ctc -> log_softmax ->
    ctc_cost, fp32, loss and gradients against the CPU oracle on the same weights."""
    from summarymixing_amd.lobes.models.transformer.TransformerASR import EncoderWrapper, TransformerASR
    from summarymixing_amd.nnet.activations import Softmax
    from summarymixing_amd.nnet.linear import Linear
    from summarymixing_amd.nnet.losses import ctc_loss
    torch.manual_seed(21)
    B, T, Fin, d, J, V, S = 3, 60, 80, 64, 96, 40, 8
    net = TransformerASR(tgt_vocab=V, input_size=Fin, d_model=d, nhead=4, num_encoder_layers=2, num_decoder_layers=0,
                         d_ffn=128, dropout=0.0, encoder_module="conformer", conformer_activation="swish",
                         attention_type="SummaryMixing", mode="SummaryMixing-fast", local_proj_out_dim=d,
                         local_proj_hid_dim=[d], summary_hid_dim=[d], summary_out_dim=d, causal=False, kernel_size=15)
    enc = EncoderWrapper(net).cuda()
    proj_enc, proj_ctc = Linear(J, input_size=d).cuda(), Linear(V, input_size=J).cuda()
    src = torch.randn(B, T, Fin)
    wav_len = torch.tensor([1.0, 0.7, 0.85])
    targets = torch.randint(1, V, (B, S))
    tg_rel = torch.tensor([1.0, 0.5, 0.75])
    x = enc(src.cuda(), wav_len.cuda())
    lp = Softmax(apply_log=True)(proj_ctc(proj_enc(x)))
    loss = ctc_loss(lp, targets.cuda(), wav_len.cuda(), tg_rel.cuda(), 0)
    loss.backward()
    # oracle
    sd = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point())
          for k, v in enc.transformer.state_dict().items() if k != "positional_encoding.pe"}
    We, be = (p.detach().cpu().clone().requires_grad_(True) for p in (proj_enc.w.weight, proj_enc.w.bias))
    Wc, bc = (p.detach().cpu().clone().requires_grad_(True) for p in (proj_ctc.w.weight, proj_ctc.w.bias))
    xo = O.asr_encode(src, wav_len, sd, "conformer", "swish", "SummaryMixing-fast", d)
    lo = O.ctc_loss(O.log_softmax((xo @ We.t() + be) @ Wc.t() + bc), targets, wav_len, tg_rel, 0)
    lo.backward()
    assert abs(loss.item() - lo.item()) <= 1e-3 * abs(lo.item()), (loss.item(), lo.item())
    assert rel_err(proj_ctc.w.weight.grad, Wc.grad) <= 2e-3 and rel_err(proj_ctc.w.bias.grad, bc.grad) <= 2e-3
    assert rel_err(proj_enc.w.weight.grad, We.grad) <= 2e-3 and rel_err(proj_enc.w.bias.grad, be.grad) <= 2e-3
    params = dict(enc.transformer.named_parameters())
    for k in ("custom_src_module.layers.0.w.weight", "encoder.layers.1.ffn_module2.1.ffn.0.weight",
              "encoder.layers.0.mha_layer.summary_local_merging.linear.w.weight", "encoder.layers.0.mha_layer.global_proj.linear.w.bias", "encoder.norm.norm.weight"):
        assert rel_err(params[k].grad, sd[k].grad) <= 3e-3, k


def test_ctc_gradient_is_bit_reproducible_with_repeated_labels():
    """The per-label occupancy sums of the CTC gradient follow the label's occurrence chain in a fixed order (no LDS
    atomics since round 3): a small vocabulary with many repeats, two runs, identical bits - and the torch reference."""
    from summarymixing_amd.nnet.activations import Softmax
    from summarymixing_amd.nnet.losses import ctc_loss
    g = torch.Generator().manual_seed(11)
    B, T, V, S = 4, 160, 6, 60                              # 5 labels over 60 positions: every label repeats ~12 times
    logits = torch.randn(B, T, V, generator=g) * 1.5
    targets = torch.randint(1, V, (B, S), generator=g)
    in_rel, tg_rel = torch.tensor([1.0, 0.9, 1.0, 0.8]), torch.tensor([1.0, 0.5, 0.7, 1.0])
    grads = []
    for _ in range(2):
        x = logits.cuda().requires_grad_(True)
        loss = ctc_loss(Softmax(apply_log=True)(x), targets.cuda(), in_rel.cuda(), tg_rel.cuda(), 0, "mean")
        loss.backward()
        grads.append(x.grad.clone())
    assert torch.equal(grads[0], grads[1])
    xr = logits.clone().requires_grad_(True)
    O.ctc_loss(O.log_softmax(xr), targets, in_rel, tg_rel, 0, "mean").backward()
    assert rel_err(grads[0].cpu(), xr.grad) <= 5e-4
