"""Front-end kernels against this repo's CPU spec in oracle/smx_oracle.py (the reference pins
nothing here: the arithmetic is un-vendored SpeechBrain code -> 'parity unpinned', self-consistency tests)."""
import pytest
import torch

from tests._util import rel_err

pytestmark = pytest.mark.gpu


def test_fbank_matches_cpu_spec():
    from oracle import smx_oracle as O
    from summarymixing_amd.lobes.features import Fbank
    torch.manual_seed(0)
    B, Lw = 3, 16000
    t = torch.arange(Lw) / 16000.0
    wav = 0.3 * torch.sin(2 * torch.pi * 440 * t)[None] * torch.tensor([1.0, 0.5, 0.1])[:, None] + 0.01 * torch.randn(B, Lw)
    wav[2, 9000:] = 0.0                                         # trailing silence -> exercises amin / top_db clamp
    ref = O.fbank(wav.double(), n_fft=512, win_length_ms=32, n_mels=80).float()
    fb = Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=32).cuda()
    out = fb(wav.cuda())
    assert out.shape == ref.shape == (B, 101, 80)
    assert (out.cpu() - ref).abs().max() < 2e-2                 # dB; fp32 DFT vs float64 reference
    assert (out[2].amax() - out[2].amin()).item() <= 80.0 + 1e-3


@pytest.mark.parametrize("Fm", [80, 40])      # 80: the recipe (fused first block); 40: F % 16 != 0 -> im2col + Linear + LayerNorm fallback
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 4e-2)])
def test_conv_frontend_fwd_bwd(dtype, tol, Fm):
    from oracle import smx_oracle as O
    from summarymixing_amd.lobes.models.convolution import ConvolutionFrontEnd
    torch.manual_seed(1)
    B, T = 3, 37                                                 # odd T: ceil division + reflect at both edges
    fe = ConvolutionFrontEnd((None, None, Fm), out_channels=(64, 32), dropout=0.0).cuda()
    with torch.no_grad():
        for blk in fe.blocks:
            blk.norm.weight.normal_(1.0, 0.1)
            blk.norm.bias.normal_(0.0, 0.1)
    x = torch.randn(B, T, Fm)
    sd = {k: v.double().cpu().requires_grad_(True) for k, v in fe.state_dict_for_oracle().items()}
    xr = x.double().requires_grad_(True)
    ref = O.conv_frontend(xr, sd)
    xg = x.cuda().to(dtype).requires_grad_(True)
    y = fe(xg)
    assert y.shape == (B, 10, Fm // 4, 32)
    assert rel_err(y.reshape(B, 10, -1), ref) <= tol
    r = torch.randn(ref.shape)
    (ref * r.double()).sum().backward()
    (y.reshape(B, 10, -1).float() * r.cuda()).sum().backward()
    # gradients: relative Frobenius error (bf16 activations are stored rounded between the two conv blocks; single
    # elements of a 9-tap gradient that nearly cancel are not a meaningful max-norm target)
    def fro(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).norm() / b.norm())
    gtol = 3 * tol
    assert fro(fe.blocks[0].conv.weight.grad, sd["convblock_0.conv.weight"].grad) <= gtol
    assert fro(fe.blocks[1].conv.weight.grad, sd["convblock_1.conv.weight"].grad) <= gtol
    assert fro(fe.blocks[1].conv.bias.grad, sd["convblock_1.conv.bias"].grad) <= gtol
    assert fro(fe.blocks[0].norm.weight.grad, sd["convblock_0.norm.weight"].grad) <= gtol
    assert fro(fe.blocks[1].norm.bias.grad, sd["convblock_1.norm.bias"].grad) <= gtol


def test_im2col_col2im_are_adjoint():
    """<im2col(x), c> == <x, col2im(c)> (exact adjoint incl. the reflected edges)."""
    from summarymixing_amd import ops
    torch.manual_seed(2)
    B, T, F_, C, Kp = 2, 9, 7, 4, 40
    x = torch.randn(B, T, F_, C, device="cuda")
    c = torch.randn(B * 5 * 4, Kp, device="cuda")
    c[:, 9 * C:] = 0
    lhs = (ops.im2col_s2(x, Kp) * c).sum()
    rhs = (x * ops.col2im_s2(c, B, T, F_, C)).sum()
    assert abs(lhs.item() - rhs.item()) <= 1e-3 * abs(lhs.item()) + 1e-3


def test_fbank_conv_encoder_chain_runs():
    """wav -> fbank -> conv subsampling -> EncoderWrapper: shapes of the recipe (T_fb/4 frames x 640 features)."""
    from summarymixing_amd.lobes.features import Fbank
    from summarymixing_amd.lobes.models.convolution import ConvolutionFrontEnd
    from summarymixing_amd.lobes.models.transformer.TransformerASR import EncoderWrapper, TransformerASR
    torch.manual_seed(3)
    wav = torch.randn(2, 16000 * 2).cuda() * 0.1
    feats = Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=32).cuda()(wav)            # (2, 201, 80)
    cnn = ConvolutionFrontEnd((None, None, 80), dropout=0.0).cuda()
    h = cnn(feats.bfloat16())
    assert h.shape == (2, 51, 20, 32)
    net = TransformerASR(tgt_vocab=10, input_size=640, d_model=64, nhead=4, num_encoder_layers=1, num_decoder_layers=0,
                         d_ffn=128, dropout=0.0, encoder_module="conformer", attention_type="SummaryMixing",
                         mode="SummaryMixing-fast", local_proj_hid_dim=[64], local_proj_out_dim=64, summary_hid_dim=[64],
                         causal=False).cuda()
    y = EncoderWrapper(net)(h, torch.tensor([1.0, 0.7]).cuda())
    assert y.shape == (2, 51, 64) and torch.isfinite(y).all()
    y.float().sum().backward()
    assert cnn.blocks[0].conv.weight.grad is not None and torch.isfinite(cnn.blocks[0].conv.weight.grad).all()


@pytest.mark.parametrize("norm_type", ["global", "batch", "sentence"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
def test_input_normalization_matches_spec(norm_type, dtype, tol):
    """InputNormalization (recipe key `normalize`) against the oracle restatement of the SpeechBrain semantics: ragged
    lengths, running global statistics over three training batches, frozen statistics in eval / past update_until_epoch."""
    from oracle import smx_oracle as O
    from summarymixing_amd.lobes.features import InputNormalization
    torch.manual_seed(7)
    mod = InputNormalization(norm_type=norm_type, update_until_epoch=2).cuda().train()
    st = O.InputNormalizationState()
    B, T, F = 4, 57, 80
    for step, epoch in enumerate([0, 0, 1, 5]):
        x = (torch.randn(B, T, F) * 7.0 - 20.0).to(dtype)
        lens = torch.tensor([1.0, 0.53, 0.8, 0.31])
        ref = O.input_normalization(x.double(), lens, st, norm_type=norm_type, update_until_epoch=2, epoch=epoch)
        y = mod(x.cuda(), lens.cuda(), epoch=epoch)
        assert rel_err(y, ref) <= tol, (step, rel_err(y, ref))
    if norm_type == "global":
        assert mod.count == 4 and st.count == 4
        assert rel_err(mod.glob_mean, st.glob_mean) <= 1e-4 and rel_err(mod.glob_std, st.glob_std) <= 1e-3
        mod.eval()
        x = (torch.randn(B, T, F) * 3.0).to(dtype)
        ref = O.input_normalization(x.double(), lens, st, norm_type="global", training=False)
        assert rel_err(mod(x.cuda(), lens.cuda()), ref) <= tol and mod.count == 4


@pytest.mark.parametrize("B,T,Fq", [(2, 19, 40), (2, 20, 40), (1, 5, 4), (3, 4, 7), (1, 1001, 40)])
def test_direct_conv_dgrad_matches_float64_conv_transpose(B, T, Fq):
    """smx_conv2d_s2_dgrad (the recipe's second block: 64 -> 32 channels, 3x3, stride 2, reflect pad 1) against autograd
    of the float64 convolution, and against the dgrad GEMM + col2im pair it replaces."""
    import torch.nn.functional as tF
    from summarymixing_amd import ops
    torch.manual_seed(T + Fq)
    C, O = 64, 32
    T2, F2 = (T + 1) // 2, (Fq + 1) // 2
    W = torch.randn(O, C, 3, 3) * 0.1
    dy = torch.randn(B, T2, F2, O)
    x = torch.zeros(B, C, T, Fq, dtype=torch.float64, requires_grad=True)
    Wb = W.bfloat16().double()
    dyb = dy.bfloat16().double()
    y = tF.conv2d(tF.pad(x, (1, 1, 1, 1), mode="reflect"), Wb, None, stride=2)            # (B, O, T2, F2)
    (y * dyb.permute(0, 3, 1, 2)).sum().backward()
    ref = x.grad.permute(0, 2, 3, 1)                                                       # (B, T, F, C)
    wg = W.permute(0, 2, 3, 1).reshape(O, 9 * C).cuda().bfloat16().contiguous()            # GEMM layout: column (dt*3+df)*C + c
    dy2 = dy.cuda().bfloat16().reshape(B * T2 * F2, O).contiguous()
    assert ops.conv2d_s2_dgrad_ok(dy2, C, O, T, Fq)
    got = ops.conv2d_s2_dgrad(dy2, wg, B, T, Fq, C)
    torch.cuda.synchronize()
    assert rel_err(got, ref) <= 1e-2                             # fp32 accumulation of bf16 products, one bf16 rounding at the end
    dcol = torch.empty(B * T2 * F2, 9 * C, dtype=torch.bfloat16, device="cuda")
    ops.gemm(ops.L.GEMM_NN, dy2, wg, dcol, B * T2 * F2, 9 * C, O, ops.epilogue())
    old = ops.col2im_s2(dcol, B, T, Fq, C)
    assert rel_err(got, old.float()) <= 2e-2                     # (the old path rounds the 9 C gradient columns to bf16 first)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)])
@pytest.mark.parametrize("B,T,Fq", [(2, 37, 80), (3, 8, 16), (1, 2, 160), (2, 101, 48)])
def test_fused_first_conv_block_matches_float64(B, T, Fq, dtype, tol):
    """smx_conv1_ln_fwd / _bwd (conv 3x3 s2 reflect + LayerNorm over (F/2, 64) + LeakyReLU in one pass; the backward recomputes
    the convolution) against float64 torch autograd: output, statistics and all four parameter gradients."""
    import torch.nn.functional as tF
    from summarymixing_amd import _lib as L, ops
    torch.manual_seed(T * Fq)
    O, F2, T2 = 64, Fq // 2, (T + 1) // 2
    x = torch.randn(B, T, Fq)
    W = (torch.randn(O, 1, 3, 3) * 0.4).double().requires_grad_(True)
    b = (torch.randn(O) * 0.2).double().requires_grad_(True)
    g = (torch.randn(F2, O) * 0.3 + 1).double().requires_grad_(True)
    be = (torch.randn(F2, O) * 0.3).double().requires_grad_(True)
    xq = x.cuda().to(dtype)
    xr = xq.double().cpu()
    y = tF.conv2d(tF.pad(xr[:, None], (1, 1, 1, 1), mode="reflect"), W, b, stride=2).permute(0, 2, 3, 1)   # (B, T2, F2, O)
    ref = tF.leaky_relu(tF.layer_norm(y, (F2, O), g, be, 1e-5), 0.01)
    assert ops.conv1_ln_ok(xq, O)
    w9 = W.detach().reshape(O, 9).float().cuda().contiguous()
    a, st = ops.conv1_ln_fwd(xq, w9, b.detach().float().cuda(), g.detach().float().cuda().view(-1),
                             be.detach().float().cuda().view(-1), 1e-5, L.ACT_LEAKY_RELU)
    assert rel_err(a.view(B, T2, F2, O), ref) <= tol
    mean = y.detach().reshape(B * T2, -1).mean(1)
    assert rel_err(st[:, 0], mean) <= 1e-4 if mean.abs().max() > 1e-3 else True
    da = torch.randn(B * T2, F2 * O)
    daq = da.cuda().to(dtype)
    (ref * daq.double().cpu().view(B, T2, F2, O)).sum().backward()
    gr = ops.conv1_ln_bwd(daq, xq, w9, b.detach().float().cuda(), g.detach().float().cuda().view(-1),
                          be.detach().float().cuda().view(-1), st, L.ACT_LEAKY_RELU)
    D = F2 * O
    gt = 3 * tol
    assert rel_err(gr[:D], g.grad.reshape(-1)) <= gt and rel_err(gr[D:2 * D], be.grad.reshape(-1)) <= gt
    assert rel_err(gr[2 * D:2 * D + 9 * O], W.grad.reshape(-1)) <= gt and rel_err(gr[2 * D + 9 * O:], b.grad) <= gt
    gr2 = ops.conv1_ln_bwd(daq, xq, w9, b.detach().float().cuda(), g.detach().float().cuda().view(-1),
                           be.detach().float().cuda().view(-1), st, L.ACT_LEAKY_RELU)
    assert torch.equal(gr, gr2)                                  # fixed-order reductions: bit-reproducible


@pytest.mark.parametrize("B,T,Fq", [(2, 19, 40), (2, 20, 40), (1, 5, 4), (3, 4, 7), (1, 1001, 40), (4, 64, 38)])
def test_patch_free_conv_forward_and_wgrad_match_float64(B, T, Fq):
    """smx_conv2d_s2_fwd / _wgrad (the GEMM kernels gather the 3x3-stride-2 patches from the channels-last input: no im2col)
    against the float64 convolution and its autograd weight / bias gradient; two wgrad runs are bit-identical."""
    import torch.nn.functional as tF
    from summarymixing_amd import ops
    torch.manual_seed(3 * T + Fq)
    C, O = 64, 32
    T2, F2 = (T + 1) // 2, (Fq + 1) // 2
    x = torch.randn(B, T, Fq, C).cuda().bfloat16()
    W = (torch.randn(O, C, 3, 3) * 0.1).bfloat16()
    bias = torch.randn(O) * 0.3
    Wd = W.double().requires_grad_(True)
    bd = bias.double().requires_grad_(True)
    ref = tF.conv2d(tF.pad(x.double().cpu().permute(0, 3, 1, 2), (1, 1, 1, 1), mode="reflect"), Wd, bd, stride=2).permute(0, 2, 3, 1)
    wg = W.permute(0, 2, 3, 1).reshape(O, 9 * C).cuda().contiguous()
    assert ops.conv2d_s2_direct_ok(x, O)
    y = ops.conv2d_s2_fwd(x, wg, bias.cuda(), O)
    assert rel_err(y.view(B, T2, F2, O), ref) <= 1e-2
    dy = torch.randn(B * T2 * F2, O).cuda().bfloat16()
    (ref * dy.double().cpu().view(B, T2, F2, O)).sum().backward()
    outs = []
    for _ in range(2):
        gw, gb = torch.zeros(O, 9 * C, device="cuda"), torch.zeros(O, device="cuda")
        ops.conv2d_s2_wgrad(dy, x, gw, gb)
        torch.cuda.synchronize()
        outs.append((gw.clone(), gb.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    gw_ref = Wd.grad.permute(0, 2, 3, 1).reshape(O, 9 * C)
    assert rel_err(outs[0][0], gw_ref) <= 1e-3 and rel_err(outs[0][1], bd.grad) <= 1e-3
