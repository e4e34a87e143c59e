#!/usr/bin/env python3
"""Per-wave s_memtime stamps of one smx_ffn_fwd launch (chunk 1 = steady state)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, ops
from tools.ffn_bench import mk
N = int(os.environ.get("N", 64000)); D, F = 256, 1024
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
x, W1, b1, W2, b2, res, gam, bet = mk(N, D, F)
if mode == "train":
    fn = lambda: ops.ffn_fwd(x, W1, b1, W2, b2, L.ACT_SWISH, res, 0.5, (0.15, 1), (0.15, 2), save_z=True, ln_next=(gam, bet, 1e-5, True))
elif mode == "nodrop":
    fn = lambda: ops.ffn_fwd(x, W1, b1, W2, b2, L.ACT_SWISH, res, 0.5, None, None, save_z=True, ln_next=(gam, bet, 1e-5, True))
else:
    fn = lambda: ops.ffn_fwd(x, W1, b1, W2, b2, L.ACT_SWISH, res, 0.5, None, None, save_z=False, ln_next=None)
lib = L.lib(); lib.smx_debug_set_timing_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3): fn()
nw = ((N + 127) // 128) * 4
buf = torch.zeros(nw * 16, dtype=torch.int64, device="cuda")
lib.smx_debug_set_timing_buffer(ctypes.c_void_p(buf.data_ptr()))
fn(); torch.cuda.synchronize()
lib.smx_debug_set_timing_buffer(None)
s = buf.view(-1, 16).cpu().double()
s = s[(s[:, 0] > 0) & (s[:, 7] > 0)]
t0 = s[:, 0].min()
names = ["prologue (X, first sync)", "G1(0) + B(0)", "B(1) slots 0-15", "B(1) slots 16-31", "B(2..) + tail G2", "final epilogue"]
s2 = s[:, [0, 1, 2, 3, 5, 6, 7]]
d = s2[:, 1:] - s2[:, :-1]
print(f"{mode}: waves {len(s)}  kernel span {float(s[:, 7].max() - t0):.0f} ticks  mean wave lifetime {float((s[:, 7] - s[:, 0]).mean()):.0f}")
for i, n in enumerate(names): print(f"  {n:26s} mean {float(d[:, i].mean()):10.0f}  p10 {float(d[:, i].quantile(0.1)):10.0f}  p90 {float(d[:, i].quantile(0.9)):10.0f}")
print(f"  in SYNC: vmcnt wait mean {float(s[:, 8].mean()):.0f}  barrier mean {float(s[:, 9].mean()):.0f} (per wave, whole kernel; 64 syncs)")
first = s[:, 0] - t0
print(f"  wave start offset: median {float(first.median()):.0f} max {float(first.max()):.0f}; second-round waves {int((first > first.max() / 2).sum())}")
