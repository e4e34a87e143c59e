"""smx_pool_bcast alone: forward (mean + dropped repeat) and backward (sum of ds + act / mask backward), us per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from summarymixing_amd import _lib as L, ops

def timeit(fn, n=200):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for B, T, D in [(128, 500, 256), (64, 500, 256), (72, 500, 256), (10, 375, 512), (1, 500, 256), (128, 500, 512)]:
    s = torch.randn(B * T, D, device="cuda").bfloat16()
    z = torch.randn(B * T, D, device="cuda").bfloat16()
    mask = (torch.rand(B * T, device="cuda") > 0.1).to(torch.uint8)
    cat = torch.empty(B * T, 2 * D, device="cuda", dtype=torch.bfloat16)
    ds = torch.empty(B * T, D, device="cuda", dtype=torch.bfloat16)
    if not ops.pool_bcast_ok(B, T, D):
        print(B, T, D, "not taken"); continue
    _, inv = ops.pool_bcast(s, mask, B, T, ds=cat[:, D:], want_inv=True, drop=(0.15, 1234))
    tf = timeit(lambda: ops.pool_bcast(s, mask, B, T, ds=cat[:, D:], want_inv=True, drop=(0.15, 1234)))
    tb = timeit(lambda: ops.pool_bcast(cat[:, D:], None, B, T, ds=ds, scale=False, want_mean=False, inv_in=inv, z=z, mask_out=mask, act=L.ACT_SWISH))
    mb = B * T * D * 2 / 1e6
    print(f"B {B:4d} T {T:4d} D {D:4d}: forward {tf:6.1f} us ({2 * mb / tf:4.2f} TB/s)   backward {tb:6.1f} us ({3 * mb / tb:4.2f} TB/s)")
