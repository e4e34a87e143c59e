"""Uninitialised-read hunt: poison the caching allocator's free blocks with NaN, run training steps, look for NaN."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoder
from summarymixing_amd.trainer import FlatAdamW
from summarymixing_amd import functional as F

def poison():
    torch.cuda.synchronize()
    big = [torch.full((64 << 20,), float("nan"), device="cuda") for _ in range(8)]       # 8 x 256 MB
    mid = [torch.full((1 << 18,), float("nan"), device="cuda") for _ in range(256)]      # 256 x 1 MB
    small = [torch.full((n,), float("nan"), device="cuda") for n in (128, 512, 2048, 8192, 32768) for _ in range(64)]
    del big, mid, small
    torch.cuda.synchronize()

def run(d, f, B, T, dtype, dropout, k=31, layers=2):
    torch.manual_seed(7)
    enc = ConformerEncoder(layers, d, f, 4, kernel_size=k, activation="swish", dropout=dropout, attention_type="SummaryMixing",
                           local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], mode="SummaryMixing-fast").cuda()
    opt = FlatAdamW(enc, lr=1e-2, max_grad_norm=5.0, compute_dtype=dtype)
    g = torch.Generator().manual_seed(3)
    X = torch.randn(B, T, d, generator=g).cuda().to(dtype)
    R = torch.randn(B, T, d, generator=g).cuda().to(dtype)
    lens = torch.randint(T // 2, T + 1, (B,), generator=g); lens[0] = T
    PAD = (torch.arange(T)[None] < lens[:, None]).cuda()
    res = []
    for it in range(3):
        poison()
        opt.zero_grad()
        y, _ = enc(X, src_key_padding_mask=PAD)
        y.backward(R)
        opt.step()
        torch.cuda.synchronize()
        res.append((opt.flat_g.clone(), opt.flat_p.clone()))
        bad_g = [n for n, p in enc.named_parameters() if not torch.isfinite(p.grad).all()]
        print(f"d={d} {dtype} it={it}: y finite {torch.isfinite(y).all().item()}, non-finite grads: {bad_g[:6]} ({len(bad_g)}), skipped {opt.skipped_steps()}", flush=True)
    return res

for dtype in (torch.float32, torch.bfloat16):
    run(64, 128, 4, 120, dtype, 0.0)
run(64, 128, 4, 120, torch.bfloat16, 0.15)
run(256, 1024, 16, 500, torch.bfloat16, 0.15)
run(256, 1024, 66, 500, torch.bfloat16, 0.0, layers=1)
