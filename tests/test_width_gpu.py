"""GPU parity AT THE BENCHMARKED WIDTHS: forward, dL/dx and EVERY parameter gradient of the HIP path against
torch.autograd of the fp64 CPU oracle, at frame counts large enough that the GEMM dispatch picks the same kernels as
bench.py does at 64 000 frames (wide 128x256 tile, LDS-DMA wgrad slabs + deferred reduction + side stream, all at their
defaults):

  C2b  2 Conformer-SummaryMixing layers  d=256 f=1024 l=256 k=31   33 000 ragged frames  (Conformer.py:479-537)
  C2a  1 Conformer-SummaryMixing layer   d=512 f=2048 l=512 k=31   16 500 ragged frames
  C4   1 Branchformer-SummaryMixing layer d=512 csgu 3072 k=31 T=250   16 500 ragged frames  (Branchformer.py:243-334)
  C2b  12-layer bf16 forward error report against fp64 (max-rel and RMS-rel)

Tolerances: fp32 1e-3 (north_star) on outputs and every gradient, both as max|a-b|/max|b| and as RMS-relative error.
bf16: activations are STORED in bf16 between kernels, so the error grows with depth; asserted per test below and the
measured values are written to gpurun_out/parity_errors.jsonl (quoted in DESIGN.md §I.1)."""
import pytest
import torch

from tests._util import report, rel_err, rms_rel

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("ln_fuse_mode")]   # (both LayerNorm dispatches: tests/conftest.py)


def _init(mod, seed):
    torch.manual_seed(seed)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_normal_(p)
            elif "bias" in n:
                p.normal_(0, 0.05)
            if "csgu.conv.conv.weight" in n:            # upstream init is N(0, 1e-6): give the taps a real gradient path
                p.normal_(0, 0.1)


def _batch(B, T, d, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, d, generator=g)
    lens = torch.round((0.5 + 0.5 * torch.rand(B, generator=g)) * T).long()
    lens[0] = T
    pad = torch.arange(T)[None] < lens[:, None]
    r = torch.randn(B, T, d, generator=g) / (B * T) ** 0.5
    return x, pad, r


_cache = {}


def _oracle(key, fwd, sd32, x, r):
    """fp64 outputs + gradients of L = sum(y * r) w.r.t. x and every floating-point entry of the state dict."""
    if key not in _cache:
        sd = {k: v.detach().double().requires_grad_(v.is_floating_point()) for k, v in sd32.items()}
        xd = x.double().requires_grad_(True)
        y = fwd(xd, sd)
        leaves = [xd] + [v for v in sd.values() if v.requires_grad]
        grads = torch.autograd.grad((y * r.double()).sum(), leaves, allow_unused=True)
        names = ["x"] + [k for k, v in sd.items() if v.requires_grad]
        _cache[key] = (y.detach(), dict(zip(names, grads)))
    return _cache[key]


def _run_gpu(mod, call, x, r, dtype):
    mod = mod.cuda().eval()          # dropout off (the Branchformer cell keeps global_dropout = 0.1 in train() like the reference)
    for p in mod.parameters():
        p.grad = None
    xg = x.cuda().to(dtype).requires_grad_(True)
    y = call(mod, xg)
    y.backward(r.cuda().to(dtype))
    torch.cuda.synchronize()
    return y.detach(), xg.grad, {n: p.grad for n, p in mod.named_parameters()}


def _compare(name, dtype, y, gx, gp, yref, gref, ftol, gtol, rms_ftol, rms_gtol):
    errs = {"out": (rel_err(y, yref), rms_rel(y, yref)), "dx": (rel_err(gx, gref["x"]), rms_rel(gx, gref["x"]))}
    for k, g in gp.items():
        if gref.get(k) is None:
            continue
        assert g is not None, f"{k}: no gradient produced"
        errs[k] = (rel_err(g, gref[k]), rms_rel(g, gref[k]))
    worst = max((v[0], k) for k, v in errs.items() if k != "out")
    worst_rms = max((v[1], k) for k, v in errs.items() if k != "out")
    report(name, {"dtype": str(dtype), "out_maxrel": errs["out"][0], "out_rmsrel": errs["out"][1],
                  "dx_maxrel": errs["dx"][0], "dx_rmsrel": errs["dx"][1], "worst_grad_maxrel": worst[0], "worst_grad": worst[1],
                  "worst_grad_rmsrel": worst_rms[0], "worst_grad_rms": worst_rms[1], "n_param_grads": len(errs) - 2})
    assert errs["out"][0] <= ftol and errs["out"][1] <= rms_ftol, ("out", errs["out"])
    for k, (e, erms) in errs.items():
        if k == "out":
            continue
        assert e <= gtol, (k, e)
        assert erms <= rms_gtol, (k, "rms", erms)


# tolerances: (forward max-rel, gradient max-rel, forward RMS-rel, gradient RMS-rel)
# bf16, Conformer (float32 residual stream = the reference's autocast semantics): the north_star bars, 1e-2 forward / 3e-2 gradients
TOLS = {torch.float32: (1e-3, 1e-3, 1e-3, 1e-3), torch.bfloat16: (1e-2, 3e-2, 1e-2, 2e-2)}
TOLS_BF16_STREAM = TOLS


def _conformer(layers, d, f):
    from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoder
    enc = ConformerEncoder(layers, d, f, 4, kernel_size=31, activation="swish", dropout=0.0, attention_type="SummaryMixing",
                           local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], mode="SummaryMixing-fast")
    _init(enc, 11)
    return enc


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_c2b_two_layers_fwd_bwd_all_grads(dtype):
    """BASELINE config 2 widths: what BENCH times (wide-tile dgrads, LDS-DMA wgrad slabs on the side stream, one deferred
    reduction per layer, bias gradients out of the wgrad, fused act-grad dgrads) composed over two layers."""
    from oracle import smx_oracle as O
    d, f, B, T = 256, 1024, 66, 500
    enc = _conformer(2, d, f)
    x, pad, r = _batch(B, T, d, 21)
    sd32 = {k: v.clone() for k, v in enc.state_dict().items()}
    yref, gref = _oracle("c2b", lambda xd, sd: O.conformer_encoder(xd, sd, "", "swish", "SummaryMixing-fast", d, None, pad), sd32, x, r)
    y, gx, gp = _run_gpu(enc, lambda m, xg: m(xg, src_key_padding_mask=pad.cuda())[0], x, r, dtype)
    _compare("c2b_2layers_d256_f1024_33000frames", dtype, y, gx, gp, yref, gref, *TOLS[dtype])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_c2a_one_layer_fwd_bwd_all_grads(dtype):
    """The recipe-faithful LibriSpeech widths (d=512, d_ffn=2048)."""
    from oracle import smx_oracle as O
    d, f, B, T = 512, 2048, 33, 500
    enc = _conformer(1, d, f)
    x, pad, r = _batch(B, T, d, 22)
    sd32 = {k: v.clone() for k, v in enc.state_dict().items()}
    yref, gref = _oracle("c2a", lambda xd, sd: O.conformer_encoder(xd, sd, "", "swish", "SummaryMixing-fast", d, None, pad), sd32, x, r)
    y, gx, gp = _run_gpu(enc, lambda m, xg: m(xg, src_key_padding_mask=pad.cuda())[0], x, r, dtype)
    _compare("c2a_1layer_d512_f2048_16500frames", dtype, y, gx, gp, yref, gref, *TOLS[dtype])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_c4_branchformer_layer_cv_dims_fwd_bwd_all_grads(dtype):
    """BASELINE config 4 widths (CommonVoice yaml): d=512, nhead=1, full mode, csgu 3072, k=31 reflect-padded, T=250."""
    from oracle import smx_oracle as O
    from summarymixing_amd.lobes.models.transformer.Branchformer import BranchformerEncoder
    d, B, T = 512, 66, 250
    enc = BranchformerEncoder(1, d, 1, kernel_size=31, activation=torch.nn.GELU, dropout=0.0, attention_type="SummaryMixing",
                              csgu_linear_units=3072, local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d],
                              summary_out_dim=d, mode="SummaryMixing")
    _init(enc, 12)
    x, pad, r = _batch(B, T, d, 23)
    sd32 = {k: v.clone() for k, v in enc.state_dict().items()}
    yref, gref = _oracle("c4", lambda xd, sd: O.branchformer_encoder(xd, sd, "", "gelu", "SummaryMixing", d, None, pad), sd32, x, r)
    y, gx, gp = _run_gpu(enc, lambda m, xg: m(xg, src_key_padding_mask=pad.cuda())[0], x, r, dtype)
    _compare("c4_branchformer_1layer_d512_csgu3072_16500frames", dtype, y, gx, gp, yref, gref, *TOLS_BF16_STREAM[dtype])


def test_c2b_twelve_layers_bf16_forward_error_report():
    """The benchmarked model (12 layers, d=256) in bf16 against the fp64 oracle, forward: max-rel and RMS-rel reported."""
    from oracle import smx_oracle as O
    d, f, B, T = 256, 1024, 66, 500
    enc = _conformer(12, d, f)
    x, pad, _ = _batch(B, T, d, 24)
    sd = {k: v.double() for k, v in enc.state_dict().items()}
    with torch.no_grad():
        ref = O.conformer_encoder(x.double(), sd, "", "swish", "SummaryMixing-fast", d, None, pad)
        enc = enc.cuda().eval()
        y32, _ = enc(x.cuda(), src_key_padding_mask=pad.cuda())
        y16, _ = enc(x.cuda().bfloat16(), src_key_padding_mask=pad.cuda())
    e = {"fp32_maxrel": rel_err(y32, ref), "fp32_rmsrel": rms_rel(y32, ref), "bf16_maxrel": rel_err(y16, ref),
         "bf16_rmsrel": rms_rel(y16, ref)}
    report("c2b_12layers_forward_33000frames", e)
    assert e["fp32_maxrel"] <= 1e-3 and e["fp32_rmsrel"] <= 1e-3, e
    assert e["bf16_maxrel"] <= 1e-2 and e["bf16_rmsrel"] <= 1e-2, e          # north_star bar at the benchmarked depth (2.1e-2 with a bf16 stream)


def test_c4_eighteen_layers_bf16_forward_error_report():
    """BASELINE config 4 at its REAL depth and width (…CommonVoice…branchformer_summarymixing.yaml:79-94: 18 layers, d = 512, nhead 1, full
    mode, csgu 3072, GELU) on 8 250 ragged frames against the fp64 oracle, forward.  The error of the k-layer prefix stacks is recorded
    too, so that a failing bar says WHERE the precision goes."""
    from oracle import smx_oracle as O
    from summarymixing_amd.lobes.models.transformer.Branchformer import BranchformerEncoder
    d, B, T, L = 512, 33, 250, 18
    enc = BranchformerEncoder(L, d, 1, kernel_size=31, activation=torch.nn.GELU, dropout=0.0, attention_type="SummaryMixing",
                              csgu_linear_units=3072, local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d],
                              summary_out_dim=d, mode="SummaryMixing")
    _init(enc, 13)
    x, pad, _ = _batch(B, T, d, 25)
    sd = {k: v.double() for k, v in enc.state_dict().items()}
    with torch.no_grad():
        refs, h = [], x.double()
        for i in range(L):                                  # the oracle's layer loop (Branchformer.py:447-491), prefixes kept
            h = O.branchformer_layer(h, sd, f"layers.{i}.", "gelu", "SummaryMixing", d, None, pad)
            refs.append(h)
        ref = O.layer_norm(h, sd, "norm.norm.weight", "norm.norm.bias", eps=1e-6)
        assert float((ref - O.branchformer_encoder(x.double(), sd, "", "gelu", "SummaryMixing", d, None, pad)).abs().max()) == 0.0
        enc = enc.cuda().eval()
        y32, _ = enc(x.cuda(), src_key_padding_mask=pad.cuda())
        y16, _ = enc(x.cuda().bfloat16(), src_key_padding_mask=pad.cuda())
        # the error as it accumulates: the k-layer prefix encoders (same weights, the final LayerNorm behind layer k) against the
        # oracle's own prefix - the full stack keeps its float32 stream across layers, so prefixes are run as stacks, not layer by layer
        per_depth = {}
        full_sd = enc.state_dict()
        for k in (1, 3, 6, 9, 12, 15):
            sub = BranchformerEncoder(k, d, 1, kernel_size=31, activation=torch.nn.GELU, dropout=0.0, attention_type="SummaryMixing",
                                      csgu_linear_units=3072, local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d],
                                      summary_out_dim=d, mode="SummaryMixing").cuda().eval()
            sub.load_state_dict({n: v for n, v in full_sd.items() if not n.startswith("layers.") or int(n.split(".")[1]) < k}, strict=True)
            rk = O.layer_norm(refs[k - 1], sd, "norm.norm.weight", "norm.norm.bias", eps=1e-6)
            yk, _ = sub(x.cuda().bfloat16(), src_key_padding_mask=pad.cuda())
            per_depth[k] = (round(rel_err(yk, rk), 5), round(rms_rel(yk, rk), 5))
    e = {"fp32_maxrel": rel_err(y32, ref), "fp32_rmsrel": rms_rel(y32, ref), "bf16_maxrel": rel_err(y16, ref),
         "bf16_rmsrel": rms_rel(y16, ref), "bf16_maxrel_rmsrel_of_the_k_layer_prefix": per_depth}
    report("c4_18layers_d512_csgu3072_forward_8250frames", e)
    assert e["fp32_maxrel"] <= 1e-3 and e["fp32_rmsrel"] <= 1e-3, e
    assert e["bf16_maxrel"] <= 1e-2 and e["bf16_rmsrel"] <= 1e-2, e          # north_star bar at config 4's depth and width


@pytest.mark.parametrize("mode", ["SummaryMixing-fast", "SummaryMixing"])
def test_c5_long_utterance_cell_vs_oracle(mode):
    """BASELINE config 5's length through the cell: ONE utterance of T = 30 000 frames at d = 512 (plus a ragged second one, so that the
    padding mask is live) against the fp64 oracle - the O(T) pool at a length 12 x the reference's positional-encoding limit
    (Transformer.py:306,335).  north_star bars: fp32 1e-3, bf16 1e-2 (max-rel and RMS-rel)."""
    from oracle import smx_oracle as O
    from summarymixing_amd.nnet.summary_mixing import SummaryMixing
    d, B, T = 512, 2, 30000
    torch.manual_seed(31)
    cell = SummaryMixing(enc_dim=d, nhead=1, local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], summary_out_dim=d,
                         activation=torch.nn.GELU, global_dropout=0.0, mode=mode)
    _init(cell, 14)
    g = torch.Generator().manual_seed(32)
    x = torch.randn(B, T, d, generator=g)
    pad = torch.arange(T)[None] < torch.tensor([T, 17321])[:, None]
    sd = {k: v.double() for k, v in cell.state_dict().items()}
    with torch.no_grad():
        ref = O.summary_mixing(x.double(), sd, "", mode, "gelu", d, None, pad)
        cell = cell.cuda().eval()
        y32 = cell(x.cuda(), src_padding_mask=pad.cuda())
        y16 = cell(x.cuda().bfloat16(), src_padding_mask=pad.cuda())
    valid = pad[:, :, None].expand_as(ref)                   # (padded frames of the fast / full modes are defined too: compared as well)
    e = {"fp32_maxrel": rel_err(y32, ref), "fp32_rmsrel": rms_rel(y32, ref), "bf16_maxrel": rel_err(y16, ref),
         "bf16_rmsrel": rms_rel(y16, ref), "bf16_maxrel_valid": rel_err(y16.cpu().float()[valid], ref[valid])}
    report(f"c5_cell_{mode}_d512_T30000", e)
    assert e["fp32_maxrel"] <= 1e-3 and e["fp32_rmsrel"] <= 1e-3, e
    assert e["bf16_maxrel"] <= 1e-2 and e["bf16_rmsrel"] <= 1e-2, e


def test_c5_long_utterance_conformer_layer_vs_oracle():
    """One Conformer-SummaryMixing layer (LS-yaml widths d = 512, d_ffn = 2048, k = 31) on one utterance of T = 30 000 frames + a ragged
    second one against the fp64 oracle: depthwise conv, LayerNorms and the cell at config 5's length (Conformer.py:479-537)."""
    from oracle import smx_oracle as O
    d, f, B, T = 512, 2048, 2, 30000
    enc = _conformer(1, d, f)
    g = torch.Generator().manual_seed(33)
    x = torch.randn(B, T, d, generator=g)
    pad = torch.arange(T)[None] < torch.tensor([T, 21007])[:, None]
    sd = {k: v.double() for k, v in enc.state_dict().items()}
    with torch.no_grad():
        ref = O.conformer_encoder(x.double(), sd, "", "swish", "SummaryMixing-fast", d, None, pad)
        enc = enc.cuda().eval()
        y32, _ = enc(x.cuda(), src_key_padding_mask=pad.cuda())
        y16, _ = enc(x.cuda().bfloat16(), src_key_padding_mask=pad.cuda())
    e = {"fp32_maxrel": rel_err(y32, ref), "fp32_rmsrel": rms_rel(y32, ref), "bf16_maxrel": rel_err(y16, ref), "bf16_rmsrel": rms_rel(y16, ref)}
    report("c5_conformer_layer_d512_f2048_T30000", e)
    assert e["fp32_maxrel"] <= 1e-3 and e["fp32_rmsrel"] <= 1e-3, e
    assert e["bf16_maxrel"] <= 1e-2 and e["bf16_rmsrel"] <= 1e-2, e
