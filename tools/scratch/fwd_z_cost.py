"""Forward of the C2b encoder with and without the tensors saved for the backward (pre-activations Z, LN stats)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
cfg = dict(bench.CONFIGS["c2b"])
enc = bench.build_encoder(cfg, "cuda", 0.15)
src, wav_len, r, _ = bench.synthetic_batch(cfg, 0, "cuda", torch.bfloat16)
from summarymixing_amd.trainer import FlatAdamW
opt = FlatAdamW(enc, compute_dtype=torch.bfloat16)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def fwd_train():
    y = enc(src, wav_len)          # parameters require grad -> need_bwd = True (Z and stats saved), dropout on
    del y
def fwd_nograd():
    with torch.no_grad():
        enc(src, wav_len)          # train mode (dropout on) but nothing saved
def full():
    opt.zero_grad(); y = enc(src, wav_len); y.backward(r); opt.step()
print(f"forward, saving for backward: {t(fwd_train):.2f} ms;  forward, nothing saved: {t(fwd_nograd):.2f} ms;  full step {t(full):.2f} ms")
