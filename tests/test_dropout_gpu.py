"""Training-mode dropout on the HIP path: keep rate / scaling, mask determinism per seed, backward uses the forward
mask (gradient consistency), eval() == dropout-free, and the reference's placement (per-frame dropout of the
concatenated [local, summary], summary_mixing.py:237-239,282-284)."""
import pytest
import torch

from tests._util import rel_err

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("ln_fuse_mode")]   # (both LayerNorm dispatches: tests/conftest.py)


def test_dropout_kernel_statistics_and_determinism():
    from summarymixing_amd import ops
    x = torch.ones(4096, 512, device="cuda", dtype=torch.bfloat16)
    for p in (0.1, 0.15, 0.5):
        y = ops.dropout(x, p, 1234)
        keep = (y != 0).float().mean().item()
        assert abs(keep - (1 - p)) < 3e-3
        assert abs(y.float().mean().item() - 1.0) < 1e-2                    # inverted dropout preserves the mean
        assert torch.equal(y, ops.dropout(x, p, 1234))                      # same seed -> same mask
        assert not torch.equal(y, ops.dropout(x, p, 1235))
    # strided views address the same mask by (row, col)
    big = torch.ones(100, 64, device="cuda")
    a = ops.dropout(big[:, :32], 0.3, 7)
    b = torch.empty(100, 64, device="cuda")
    ops.dropout(big[:, :32], 0.3, 7, out=b[:, 32:])
    assert torch.equal(a, b[:, 32:])
    # no obvious structure along rows / columns
    m = (ops.dropout(torch.ones(2048, 256, device="cuda"), 0.5, 99) != 0).float()
    assert (m.mean(0) - 0.5).abs().max() < 0.06 and (m.mean(1) - 0.5).abs().max() < 0.15


def test_dropout_mask_has_no_duplicate_at_pair_distance_0x02000200():
    """the pair hash read only 24 bits of idx ^ (idx >> 16), so elements 2 * 0x02000200 apart (inside one 64000 x 1024
    hidden tensor) always shared their keep / drop decision.  With the top index byte folded in they are independent."""
    from summarymixing_amd import ops
    N, D = 66000, 1024                                   # 67.6 M elements > 2 * 0x02000200 + margin
    x = torch.ones(N, D, device="cuda", dtype=torch.bfloat16)
    m = (ops.dropout(x, 0.5, 4242) != 0).view(-1)
    dist = 2 * 0x02000200
    a, b = m[: m.numel() - dist], m[dist:]
    agree = (a == b).float().mean().item()
    assert abs(agree - 0.5) < 5e-3, f"masks {dist} elements apart agree {agree:.4f} of the time (0.5 = independent, 1.0 = the old duplicate)"
    assert abs(m.float().mean().item() - 0.5) < 2e-3


@pytest.mark.parametrize("p", [0.15, 0.5])
@pytest.mark.parametrize("seed", [1234, 0xDEADBEEFCAFEF00D, 77])
def test_dropout_mask_is_uncorrelated(p, seed):
    """The counter-based mask must look like independent Bernoulli draws: no correlation between elements a few columns or
    rows apart (a one-multiply mixer - an arithmetic progression mod 2^32 - gave 0.4 at column lags 4 / 8 / 16), row means
    within the binomial spread.  Both mask generators (standalone kernel, GEMM epilogue) share the device function."""
    from summarymixing_amd import ops
    n, D = 4096, 256
    k = (ops.dropout(torch.ones(n, D, device="cuda", dtype=torch.bfloat16), p, seed) != 0).double()
    m = k.mean().item()
    assert abs(m - (1 - p)) < 3e-3
    c, v = k - m, m * (1 - m)
    for lag in (1, 2, 3, 4, 5, 6, 8, 16, 32, 64, 128):
        assert abs((c[:, :-lag] * c[:, lag:]).mean().item()) / v < 0.01, ("column lag", lag)
    for lag in (1, 2, 3, 4, 7):
        assert abs((c[:-lag] * c[lag:]).mean().item()) / v < 0.01, ("row lag", lag)
    assert abs((c[:-1, :-1] * c[1:, 1:]).mean().item()) / v < 0.01
    sd_row, sd_col = (v / D) ** 0.5, (v / n) ** 0.5
    assert (k.mean(1) - m).abs().max().item() < 5.5 * sd_row and (k.mean(0) - m).abs().max().item() < 5.5 * sd_col


@pytest.mark.parametrize("mode", ["SummaryMixing-fast", "SummaryMixing"])
def test_cell_dropout_backward_uses_forward_mask(mode):
    """y is piecewise linear in x for relu; with a fixed seed stream the directional derivative must equal <dx, v>."""
    from summarymixing_amd import ops
    from summarymixing_amd.nnet.summary_mixing import SummaryMixing
    torch.manual_seed(0)
    m = SummaryMixing(32, 2, [32], 32, [32], 32, activation="swish", global_dropout=0.3, mode=mode).cuda().train()
    x = torch.randn(3, 40, 32, device="cuda", requires_grad=True)
    pad = (torch.arange(40, device="cuda")[None] < torch.tensor([40, 23, 31], device="cuda")[:, None])
    r = torch.randn(3, 40, 32, device="cuda")

    def f(xx):
        ops._drop_state["counter"] = 1000          # replay the same seed stream -> same masks
        return (m(xx, src_padding_mask=pad) * r).sum()
    y = f(x)
    y.backward()
    v = torch.randn_like(x)
    eps = 1e-3
    with torch.no_grad():
        num = (f(x + eps * v) - f(x - eps * v)) / (2 * eps)
    ana = (x.grad * v).sum()
    assert abs(num.item() - ana.item()) <= 2e-2 * max(1.0, abs(ana.item())), (num.item(), ana.item())
    # eval() disables it and matches the dropout-free module
    m.eval()
    m2 = SummaryMixing(32, 2, [32], 32, [32], 32, activation="swish", global_dropout=0.0, mode=mode).cuda()
    m2.load_state_dict(m.state_dict())
    assert rel_err(m(x.detach(), src_padding_mask=pad), m2(x.detach(), src_padding_mask=pad)) < 1e-6


def test_conformer_layer_trains_with_dropout():
    from summarymixing_amd import ops
    from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoderLayer
    torch.manual_seed(1)
    d = 64
    layer = ConformerEncoderLayer(d_model=d, d_ffn=128, nhead=4, kernel_size=31, activation="swish", dropout=0.15,
                                  attention_type="SummaryMixing", local_proj_hid_dim=[d], local_proj_out_dim=d,
                                  summary_hid_dim=[d], mode="SummaryMixing-fast").cuda().train()
    x = torch.randn(4, 70, d, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    pad = (torch.arange(70, device="cuda")[None] < torch.tensor([70, 33, 51, 64], device="cuda")[:, None])
    ops._drop_state["counter"] = 5
    y1, _ = layer(x, src_key_padding_mask=pad)
    y1.float().sum().backward()
    assert torch.isfinite(y1).all() and torch.isfinite(x.grad).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters())
    ops._drop_state["counter"] = 5
    y2, _ = layer(x, src_key_padding_mask=pad)
    assert torch.equal(y1, y2)                                             # same seeds -> bit-identical
    y3, _ = layer(x, src_key_padding_mask=pad)
    assert not torch.equal(y1, y3)                                         # fresh seeds -> different masks
    layer.eval()
    ye1, _ = layer(x, src_key_padding_mask=pad)
    ye2, _ = layer(x, src_key_padding_mask=pad)
    assert torch.equal(ye1, ye2)


def test_fused_epilogue_dropout_equals_standalone_kernel():
    """The GEMM epilogue's fused dropout and smx_act_mask_bwd's regenerated mask index elements exactly like
    smx_dropout (n*M + m), so fused and unfused paths are interchangeable (checked in fp32: no rounding in between)."""
    from summarymixing_amd import _lib as L, ops
    torch.manual_seed(2)
    N, M, K = 300, 136, 72
    x = torch.randn(N, K, device="cuda")
    w = torch.randn(M, K, device="cuda") * 0.2
    b = torch.randn(M, device="cuda")
    res = torch.randn(N, M, device="cuda")
    mask = (torch.rand(N, device="cuda") > 0.3).view(torch.uint8)
    seed, p = 0xABCDEF0123, 0.25
    y = torch.empty(N, M, device="cuda")
    ops.gemm(L.GEMM_NT, x, w, y, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, row_mask=mask, res=res, alpha=0.5,
                                                        drop=(p, seed)))
    a = torch.empty(N, M, device="cuda")
    ops.gemm(L.GEMM_NT, x, w, a, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH))
    ops.dropout(a, p, seed, out=a)
    ref = res + 0.5 * a * mask.bool()[:, None]
    assert rel_err(y, ref) < 1e-6
    dy = torch.randn(N, M, device="cuda")
    z = torch.randn(N, M, device="cuda")
    dz = torch.empty_like(dy)
    ops.act_mask_bwd(dy, z, mask, L.ACT_GELU, 0.5, dz, drop=(p, seed))
    dz2 = torch.empty_like(dy)
    ops.act_mask_bwd(ops.dropout(dy, p, seed), z, mask, L.ACT_GELU, 0.5, dz2)
    assert rel_err(dz, dz2) < 1e-6


def test_csgu_fused_output_dropout_equals_standalone_kernel():
    """The rolling CSGU forward applies the CSGU's own dropout to its output: same mask as smx_dropout on the result
    (index = global row * D + channel), values equal up to the one bf16 rounding the separate pass adds."""
    from summarymixing_amd import _lib as L, ops
    torch.manual_seed(2)
    B, T, D, k = 4, 250, 128, 31
    x = torch.randn(B * T, D, device="cuda").bfloat16()
    gate = torch.randn(B * T, D, device="cuda").bfloat16()
    w = torch.randn(D, k, device="cuda") * 0.3
    bias = torch.randn(D, device="cuda")
    y1 = ops.dwconv_fwd(x, w, bias, B, T, D, k, False, L.PAD_REFLECT, 0, gate, drop=(0.15, 99))
    y0 = ops.dwconv_fwd(x, w, bias, B, T, D, k, False, L.PAD_REFLECT, 0, gate)
    ops.dropout(y0, 0.15, 99, out=y0)
    assert torch.equal(y1 == 0, y0 == 0) and 0.13 < (y1 == 0).float().mean().item() < 0.17
    assert rel_err(y1, y0) < 1e-2
    # a shape the rolling kernel does not take: the wrapper falls back to the separate pass with the same mask
    D2 = 48
    x2, g2 = x[:, :D2].contiguous(), gate[:, :D2].contiguous()
    a = ops.dwconv_fwd(x2, w[:D2].contiguous(), bias[:D2].contiguous(), B, T, D2, k, False, L.PAD_REFLECT, 0, g2, drop=(0.15, 5))
    b = ops.dwconv_fwd(x2, w[:D2].contiguous(), bias[:D2].contiguous(), B, T, D2, k, False, L.PAD_REFLECT, 0, g2)
    ops.dropout(b, 0.15, 5, out=b)
    assert torch.equal(a, b)


@pytest.mark.parametrize("kind", ["conformer", "branchformer"])
def test_training_mode_gradients_match_finite_differences(kind):
    """With the dropout seeds pinned (counter reset before every call) a training-mode layer is a deterministic, piecewise
    smooth function: the analytic input gradient (which regenerates every fused dropout mask in the backward kernels) must
    match central finite differences along random directions, and so must the gradient of a weight deep in the layer."""
    from summarymixing_amd import ops
    torch.manual_seed(3)
    d = 32
    if kind == "conformer":
        from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoderLayer
        layer = ConformerEncoderLayer(d_model=d, d_ffn=64, nhead=2, kernel_size=31, activation="swish", dropout=0.15,
                                      attention_type="SummaryMixing", local_proj_hid_dim=[d], local_proj_out_dim=d,
                                      summary_hid_dim=[d], mode="SummaryMixing-fast")
        wname = "ffn_module1.ffn.1.w_1.weight" if any("w_1" in n for n, _ in layer.named_parameters()) else None
    else:
        from summarymixing_amd.lobes.models.transformer.Branchformer import BranchformerEncoderLayer
        layer = BranchformerEncoderLayer(d_model=d, nhead=1, kernel_size=31, activation="gelu", dropout=0.15,
                                         csgu_linear_units=64, local_proj_hid_dim=[d], local_proj_out_dim=d,
                                         summary_hid_dim=[2 * d], summary_out_dim=d, mode="SummaryMixing")
        wname = "convolution_branch.pre_channel_proj.weight"
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_normal_(p)
        if kind == "branchformer":                        # upstream inits the CSGU conv to ~0 / 1: give it real taps
            layer.convolution_branch.csgu.conv.conv.weight.normal_(0, 0.2)
    layer = layer.cuda().train()
    names = dict(layer.named_parameters())
    wpar = names[wname] if wname in names else next(p for n, p in names.items() if p.dim() == 2)
    B, T = 3, 40
    x = torch.randn(B, T, d, device="cuda")
    pad = (torch.arange(T, device="cuda")[None] < torch.tensor([T, 25, 33], device="cuda")[:, None])
    r = torch.randn(B, T, d, device="cuda")

    def f(xx):
        ops._drop_state["counter"] = 11
        y, _ = layer(xx, src_key_padding_mask=pad)
        return (y * r).sum()

    xg = x.clone().requires_grad_(True)
    for p in layer.parameters():
        p.grad = None
    f(xg).backward()
    gx, gw = xg.grad.clone(), wpar.grad.clone()
    eps = 1e-2
    for trial in range(3):
        v = torch.randn_like(x)
        with torch.no_grad():
            fd = (f(x + eps * v) - f(x - eps * v)) / (2 * eps)
        an = (gx * v).sum()
        assert abs(float(fd - an)) <= 3e-2 * max(1.0, abs(float(an))), (kind, "dx", float(fd), float(an))
    u = torch.randn_like(wpar)
    eps = 3e-3                                            # (a unit-variance direction is a 5 % change of xavier weights at 1e-2:
    with torch.no_grad():                                 #  the curvature term alone reached 5 % of the derivative)
        w0 = wpar.detach().clone()
        wpar.copy_(w0 + eps * u); fp = f(x)
        wpar.copy_(w0 - eps * u); fm = f(x)
        wpar.copy_(w0)
    fd, an = (fp - fm) / (2 * eps), (gw * u).sum()
    assert abs(float(fd - an)) <= 3e-2 * max(1.0, abs(float(an))), (kind, "dW", float(fd), float(an))


@pytest.mark.parametrize("kind,seeds", [("conformer", 7), ("branchformer", 6)])
def test_dropout_site_count_matches_reference(kind, seeds):
    """Every dropout site of the reference layer draws exactly one seed per forward (the cell's dropout of
    cat[local, summary] draws two, one per half).  Reference sites - Conformer.py:458-472,507-536: FFN inner + FFN module
    (x2 modules), the cell's own (summary_mixing.py:237-239), the conv module's (Conformer.py:146-152) = 6 sites / 7 seeds;
    Branchformer.py:270-334 + upstream CSGU: the cell's own, dropout(x1), the CSGU's dropout(x1 * conv(x2)), dropout(x2),
    dropout(merge) = 5 sites / 6 seeds.  The cell's global_dropout is the cell default 0.1 in the Branchformer."""
    from summarymixing_amd import ops
    d = 32
    if kind == "conformer":
        from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoderLayer
        layer = ConformerEncoderLayer(d_model=d, d_ffn=64, nhead=2, kernel_size=7, activation="swish", dropout=0.15,
                                      attention_type="SummaryMixing", local_proj_hid_dim=[d], local_proj_out_dim=d,
                                      summary_hid_dim=[d], mode="SummaryMixing-fast")
        assert layer.mha_layer.global_dropout == 0.15           # Conformer.py:443 passes the layer dropout
    else:
        from summarymixing_amd.lobes.models.transformer.Branchformer import BranchformerEncoderLayer
        layer = BranchformerEncoderLayer(d_model=d, nhead=1, kernel_size=7, activation="gelu", dropout=0.15,
                                         csgu_linear_units=64, local_proj_hid_dim=[d], local_proj_out_dim=d,
                                         summary_hid_dim=[d], summary_out_dim=d, mode="SummaryMixing")
        assert layer.mha_layer.global_dropout == 0.1            # Branchformer.py:209-218: the cell default
    layer = layer.cuda().train()
    x = torch.randn(2, 20, d, device="cuda")
    c0 = ops._drop_state["counter"]
    layer(x)
    assert ops._drop_state["counter"] - c0 == seeds
    layer.eval()
    c0 = ops._drop_state["counter"]
    layer(x)
    assert ops._drop_state["counter"] == c0


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("p", [0.0, 0.15])
def test_branchformer_fused_backward_steps_match_the_separate_passes(dtype, tol, p):
    """The Branchformer layer's backward with its elementwise first steps inside the producing kernels (merge dgrad as two
    GEMMs with the dropout / activation backward of their consumers, SMX_SPLIT_MERGE_DGRAD; channel_proj1's activation backward
    inside the CSGU LayerNorm backward, SMX_PREACT_LN) against the same layer with those steps as separate passes: same
    seeds = same dropout masks, so output and every gradient agree up to the rounding of one intermediate tensor."""
    from summarymixing_amd import ops
    from summarymixing_amd.lobes.models.transformer import Branchformer as BM
    d, B, T = 64, 3, 70
    torch.manual_seed(11)
    layer = BM.BranchformerEncoderLayer(d_model=d, nhead=1, kernel_size=31, activation="gelu", dropout=p, csgu_linear_units=128,
                                        local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], summary_out_dim=d,
                                        mode="SummaryMixing").cuda().train()       # (fp32 master weights; the input's dtype is the compute dtype)
    x0 = torch.randn(B, T, d, device="cuda").to(dtype)
    r = torch.randn(B, T, d, device="cuda").to(dtype)
    pad = (torch.arange(T, device="cuda")[None] < torch.tensor([T, 50, 61], device="cuda")[:, None])   # True = valid frame
    results = []
    saved = (BM._SPLIT_MERGE_DGRAD, BM._PREACT_LN)
    try:
        for fused in (True, False):
            BM._SPLIT_MERGE_DGRAD = BM._PREACT_LN = fused
            ops._drop_state["counter"] = 1000                 # the same seeds for both runs
            layer.zero_grad()
            x = x0.clone().requires_grad_(True)
            y, _ = layer(x, src_key_padding_mask=pad)
            (y.float() * r.float()).sum().backward()
            results.append((y.detach().float(), x.grad.float(), {n: q.grad.float().clone() for n, q in layer.named_parameters() if q.grad is not None}))
    finally:
        BM._SPLIT_MERGE_DGRAD, BM._PREACT_LN = saved
    (ya, gxa, ga), (yb, gxb, gb) = results
    assert torch.equal(ya, yb)                                # the forward is untouched
    assert rel_err(gxa, gxb) <= tol
    assert ga.keys() == gb.keys() and len(ga) > 10
    for n in ga:
        assert rel_err(ga[n], gb[n]) <= tol, n
