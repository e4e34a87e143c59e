"""Gradient accumulation (…transducer.yaml:65-66: grad_accumulation_factor 4): G forward + backward passes ADD into the flat
gradient buffer exactly like autograd's accumulation, and the fused form (trainer.fuse_microbatches: one batch of all the
micro-batches) gives the same gradients - rows are computed identically wherever they sit in the batch, only the order of
the fp32 sums over frames differs.  (Micro-batches of one padded length: with different lengths the reference's own
convolution module makes an utterance's edge frames depend on the padding behind it, trainer.fuse_microbatches.)"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _setup(dtype, kind):
    import bench
    from summarymixing_amd.trainer import FlatAdamW
    cfg = dict(bench.CONFIGS["c1"])
    if kind == "branchformer":
        cfg.update(kind="branchformer", csgu=288, nhead=1)
    enc = bench.build_encoder(cfg, torch.device("cuda"), 0.0)
    if kind == "branchformer":
        enc.eval()      # (its SummaryMixing cell keeps the reference's default dropout 0.1 in train(): masks would differ per pass)
    opt = FlatAdamW(enc, lr=1e-3, compute_dtype=dtype)
    return cfg, enc, opt


def _micro(cfg, B, T, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, cfg["input"], generator=g)
    wl = 0.5 + 0.5 * torch.rand(B, generator=g)
    wl[0] = 1.0
    valid = torch.arange(T)[None] < torch.round(wl * T)[:, None]
    r = torch.randn(B, T, cfg["d"], generator=g) * valid[..., None]
    return (x * valid[..., None]).cuda().to(dtype), wl.cuda(), r.cuda().to(dtype)


@pytest.mark.parametrize("kind", ["conformer", "branchformer"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-3)])
def test_accumulated_gradients_equal_the_fused_batch(dtype, tol, kind):
    from summarymixing_amd.trainer import fuse_microbatches
    cfg, enc, opt = _setup(dtype, kind)
    micro = [_micro(cfg, 3, 60, 1, dtype), _micro(cfg, 2, 60, 2, dtype), _micro(cfg, 4, 60, 3, dtype)]
    opt.zero_grad()
    for x, wl, r in micro:
        enc(x, wl).backward(r)
    from summarymixing_amd import functional as F
    F.flush_deferred()
    torch.cuda.synchronize()
    g_seq = opt.flat_g.clone()

    xs, wls = fuse_microbatches([(m[0], m[1]) for m in micro])
    rs = torch.cat([m[2] for m in micro])
    opt.zero_grad()
    enc(xs, wls).backward(rs)
    F.flush_deferred()
    torch.cuda.synchronize()
    g_fused = opt.flat_g.clone()
    assert float(g_seq.abs().max()) > 0
    err = float((g_seq - g_fused).abs().max() / g_fused.abs().max())
    assert err < tol, f"accumulated vs fused gradients: rel err {err:.3e}"
