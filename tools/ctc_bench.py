#!/usr/bin/env python3
"""CTC head micro-benchmark at the C2b step's shape (B=128, T=500, vocab 1000, ~60 tokens per utterance)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import ops
from bench import time_kernel
B, T, V, S = 128, 500, 1000, 60
x = torch.randn(B * T, V, device="cuda").bfloat16()
tg = torch.randint(1, V, (B, S), device="cuda", dtype=torch.int32)
il = torch.randint(T // 2, T + 1, (B,), device="cuda", dtype=torch.int32)
tl = torch.randint(S // 2, S + 1, (B,), device="cuda", dtype=torch.int32)
gs = torch.full((B,), 1.0 / B, device="cuda")
lp = ops.log_softmax_fwd(x)
t1 = time_kernel(lambda: ops.log_softmax_fwd(x), iters=10, warm=2)
nll, ws = ops.ctc_fwd(lp, tg, il, tl, B, T, 0)
t2 = time_kernel(lambda: ops.ctc_fwd(lp, tg, il, tl, B, T, 0), iters=10, warm=2)
def bwd():
    ops.ctc_fwd(lp, tg, il, tl, B, T, 0)          # (the backward consumes the forward variables in place)
    return ops.ctc_bwd(lp, tg, il, tl, B, T, 0, nll, gs, ws)
t3 = time_kernel(bwd, iters=10, warm=2) - t2
g = bwd()
t4 = time_kernel(lambda: ops.log_softmax_bwd(g, lp), iters=10, warm=2)
nb = B * T * V * 2
print(f"log_softmax fwd {t1*1e6:7.1f} us ({2*nb/t1/1e9:5.0f} GB/s) | ctc alpha {t2*1e6:7.1f} us | ctc beta+grad {t3*1e6:7.1f} us "
      f"| log_softmax bwd {t4*1e6:7.1f} us ({3*nb/t4/1e9:5.0f} GB/s) | total {(t1+t2+t3+t4)*1e3:.2f} ms")
