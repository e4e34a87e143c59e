"""Size-independent properties at BASELINE.json's FULL sizes (where the CPU oracle would take too long):
config 5 pool (8 x 30000 x 512), the C2b encoder at its bench width, the long-utterance cell."""
import pytest
import torch

from tests._util import rel_err

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("ln_fuse_mode")]   # (both LayerNorm dispatches: tests/conftest.py)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_pool_full_size_split_consistency_and_linearity(dtype):
    """mean over T == length-weighted combination of the means of two halves (checksum of checksums); linear in S;
    invariant to a permutation of the valid frames; bit-reproducible."""
    from summarymixing_amd import ops
    torch.manual_seed(0)
    B, T, D = 8, 30000, 512
    s = torch.randn(B * T, D, device="cuda").to(dtype)
    lens = torch.randint(T // 2, T + 1, (B,), device="cuda")
    lens[0] = T
    mask = (torch.arange(T, device="cuda")[None] < lens[:, None])
    m8 = mask.reshape(-1).view(torch.uint8)
    full, inv = ops.masked_mean(s, m8, B, T, True, True)
    assert torch.equal(full, ops.masked_mean(s, m8, B, T, True)[0])
    h = T // 2
    s3 = s.view(B, T, D)
    a, _ = ops.masked_mean(s3[:, :h].reshape(B * h, D), mask[:, :h].reshape(-1).contiguous().view(torch.uint8), B, h, False)
    b, _ = ops.masked_mean(s3[:, h:].reshape(B * (T - h), D), mask[:, h:].reshape(-1).contiguous().view(torch.uint8), B, T - h, False)
    tol = 1e-5 if dtype == torch.float32 else 1e-5        # inputs are identical bf16 values: only fp32 summation order differs
    assert rel_err((a + b) * inv[:, None], full) <= 5e-5
    two, _ = ops.masked_mean((s.float() * 2).to(dtype), m8, B, T, True)
    assert rel_err(two, 2 * full) <= (1e-6 if dtype == torch.float32 else 1e-6)
    perm = torch.randperm(int(lens.min().item()), device="cuda")
    sp = s3.clone()
    sp[:, : perm.numel()] = s3[:, perm]
    permd, _ = ops.masked_mean(sp.view(B * T, D), m8, B, T, True)
    assert rel_err(permd, full) <= 5e-5


def test_long_utterance_cell_linear_time_and_padding_blind():
    """Config 5 shape through the cell: (4, 30000, 512) fast mode.  Valid-frame outputs do not depend on the content of
    padded frames (the cell masks before pooling, summary_mixing.py:257) and utterances do not interact."""
    from summarymixing_amd.nnet.summary_mixing import SummaryMixing
    torch.manual_seed(1)
    B, T, d = 4, 30000, 512
    m = SummaryMixing(d, 4, [d], d, [d], d, activation="swish", global_dropout=0.0, mode="SummaryMixing-fast").cuda().eval()
    x = torch.randn(B, T, d, device="cuda", dtype=torch.bfloat16)
    lens = torch.tensor([T, 17000, 25000, 9000], device="cuda")
    pad = torch.arange(T, device="cuda")[None] < lens[:, None]
    with torch.no_grad():
        y = m(x, src_padding_mask=pad)
        x2 = x.clone()
        x2[1, 17000:] = 7.0                                   # garbage in utterance 1's padding
        x2[3] = torch.randn(T, d, device="cuda", dtype=torch.bfloat16)   # a different utterance 3
        y2 = m(x2, src_padding_mask=pad)
    assert torch.equal(y[0], y2[0]) and torch.equal(y[2], y2[2])          # other utterances: bit-identical
    assert torch.equal(y[1, :17000], y2[1, :17000])                        # valid frames blind to padded content
    assert not torch.equal(y[3], y2[3])


def test_c2b_encoder_full_width_batch_independence_and_determinism():
    """12-layer d=256 Conformer-SM encoder at bench width (B=16 x T=500): per-utterance independence (LayerNorm only, no
    cross-utterance op) and run-to-run bit reproducibility of the forward."""
    import bench
    cfg = dict(bench.CONFIGS["c2b"], B=16)
    enc = bench.build_encoder(cfg, "cuda").eval()
    src, wav_len, _, _ = bench.synthetic_batch(cfg, 0, "cuda", torch.bfloat16)
    with torch.no_grad():
        y1 = enc(src, wav_len)
        y2 = enc(src, wav_len)
        src2 = src.clone()
        src2[5] = src[5].flip(0)
        y3 = enc(src2, wav_len)
    assert torch.isfinite(y1).all() and torch.equal(y1, y2)
    keep = [i for i in range(16) if i != 5]
    assert torch.equal(y1[keep], y3[keep]) and not torch.equal(y1[5], y3[5])


def test_gemm_full_size_linearity_and_fp32_agreement():
    """FFN up-projection shape (32000 x 256) x (256 x 1024): bf16 result vs the exact-fp32 MFMA path on the same
    (bf16-representable) operands, and additivity in the activations."""
    from summarymixing_amd import _lib as L, ops
    torch.manual_seed(2)
    N, K, M = 32000, 256, 1024
    x1 = torch.randn(N, K, device="cuda").bfloat16()
    x2 = torch.randn(N, K, device="cuda").bfloat16()
    w = (torch.randn(M, K, device="cuda") * 0.06).bfloat16()
    def mm(x, dt):
        y = torch.empty(N, M, device="cuda", dtype=torch.float32)
        ops.gemm(L.GEMM_NT, x.to(dt), w.to(dt), y, N, M, K, ops.epilogue(out_mode=L.OUT_F32))
        return y
    y16, y32 = mm(x1, torch.bfloat16), mm(x1, torch.float32)
    assert rel_err(y16, y32) <= 2e-6                      # same products, fp32 accumulation in both
    s = (x1.float() + x2.float())
    assert rel_err(mm(s, torch.float32), y32 + mm(x2, torch.float32)) <= 1e-5


def test_c5_twelve_layer_stack_at_full_size_properties():
    """BASELINE config 5 at the STACK level: the 12-layer d = 512 Conformer-SummaryMixing encoder on (8, 30000, 512), bf16 (what
    `bench.py --config c5` times; the oracle would need hours).  Properties: finite; per-utterance independence and
    padding-blindness of the SUMMARY (an utterance run ALONE, truncated to its own length, gives the same frames - except
    the last 12 x 15: the reference's conv module reads the content of padded frames through its k = 31 window, one halo
    per layer, SURVEY 7 "padding is not neutral"; the summary mean counts valid frames only); deterministic; linear in T."""
    import time
    from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoder
    torch.manual_seed(3407)
    B, T, d = 8, 30000, 512
    enc = ConformerEncoder(12, d, 2048, 4, kernel_size=31, activation="swish", dropout=0.0, attention_type="SummaryMixing",
                           local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], mode="SummaryMixing-fast")
    with torch.no_grad():
        for p in enc.parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_normal_(p)
    enc = enc.cuda().eval()
    g = torch.Generator().manual_seed(5)
    lens = torch.round((0.5 + 0.5 * torch.rand(B, generator=g)) * T).long()
    lens[0] = T
    pad = torch.arange(T)[None, :] < lens[:, None]
    x = (torch.randn(B, T, d, generator=g) * pad[..., None]).cuda().bfloat16()
    pad = pad.cuda()

    def run(xx, pp):
        with torch.no_grad():
            return enc(xx, src_key_padding_mask=pp)[0]
    y = run(x, pad)
    assert y.shape == (B, T, d) and torch.isfinite(y).all()
    assert torch.equal(y, run(x, pad))                                   # bit-reproducible
    for b in (1, 6):
        Lb = int(lens[b])
        alone = run(x[b:b + 1, :Lb].contiguous(), torch.ones(1, Lb, dtype=torch.bool, device="cuda"))
        keep = Lb - 12 * 15 - 8                                          # beyond the reach of the padded frames' content
        e = rel_err(y[b, :keep], alone[0, :keep])
        assert e <= 1e-2, (b, e)                                         # (different split-T pool partials / tile rows: bf16 noise only)
    # an utterance does not see its neighbours: replace every OTHER utterance by noise of another length pattern
    x2 = x.clone()
    x2[2:] = (torch.randn(B - 2, T, d, generator=g).cuda().bfloat16() * pad[2:, :, None])
    y2 = run(x2, pad)
    assert rel_err(y2[:2], y[:2]) <= 1e-5
    # linear in T
    def timed(xx, pp):
        run(xx, pp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(xx, pp)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    t_full = timed(x, pad)
    t_half = timed(x[:, :T // 2].contiguous(), pad[:, :T // 2].contiguous())
    assert 1.6 <= t_full / t_half <= 2.4, (t_full, t_half)


@pytest.mark.parametrize("B,T,d,f", [(1, 500, 256, 1024), (10, 375, 512, 2048), (6, 350, 256, 1024)])
def test_small_batch_training_step_is_bit_reproducible(B, T, d, f):
    """The round-6 small-batch kernels reduce in a FIXED order (split-K slabs summed by the row reducer in slab order, the slab-free
    grouped wgrad with one writer per gradient element, the pool's two-level fold, smx_reduce_jobs): the same training-mode step -
    dropout 0.15, the same seeds - twice from the same state gives bit-identical outputs, dL/dx and parameter gradients (one
    utterance, the recipe's 10 x 375 frames at d_model 512, 2100 frames).  No float atomics anywhere on these paths."""
    from summarymixing_amd import functional as F, ops
    from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoder
    torch.manual_seed(B * 100 + T)
    enc = ConformerEncoder(2, d, f, 4, kernel_size=31, activation="swish", dropout=0.15, attention_type="SummaryMixing",
                           local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], mode="SummaryMixing-fast").cuda().train()
    x0 = torch.randn(B, T, d, device="cuda").bfloat16()
    lens = torch.randint(T // 2, T + 1, (B,), device="cuda"); lens[0] = T
    pad = torch.arange(T, device="cuda")[None] < lens[:, None]
    r = torch.randn(B, T, d, device="cuda")
    runs = []
    for _ in range(2):
        ops._drop_state["counter"] = 1000                  # the same dropout seeds in both runs
        enc.zero_grad()
        x = x0.clone().requires_grad_(True)
        y, _ = enc(x, src_key_padding_mask=pad)
        (y.float() * r).sum().backward()
        F.flush_deferred()
        torch.cuda.synchronize()
        runs.append((y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in enc.named_parameters()}))
    (y1, g1, p1), (y2, g2, p2) = runs
    assert torch.isfinite(y1.float()).all() and (y1 != 0).any()
    assert torch.equal(y1, y2) and torch.equal(g1, g2)
    for n in p1:
        assert torch.equal(p1[n], p2[n]), n
