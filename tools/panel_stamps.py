#!/usr/bin/env python3
"""Per-wave clock stamps of one smx_gemm_panel launch (libsmx_diag.so): where does a wave's lifetime go?
usage: SMX_LIB=summarymixing_amd/libsmx_diag.so D=512 F=2048 python tools/panel_stamps.py [fwd|fwdd|bias|ag]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, ops
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
N, d, f = int(os.environ.get("N", 64000)), int(os.environ.get("D", 512)), int(os.environ.get("F", 2048))
x = torch.randn(N, d, device="cuda").bfloat16()
W = (torch.randn(f, d, device="cuda") * 0.05).bfloat16(); wp = ops.weight_pack(W, bias=torch.randn(f, device="cuda"))
y = torch.empty(N, f, device="cuda", dtype=torch.bfloat16); z = torch.randn(N, f, device="cuda").bfloat16(); b = torch.randn(f, device="cuda")
e = {"fwd": lambda: ops.epilogue(act=L.ACT_SWISH, z=z), "fwdd": lambda: ops.epilogue(act=L.ACT_SWISH, z=z, drop=(0.15, 7)),
     "bias": lambda: ops.epilogue(), "ag": lambda: ops.epilogue(act=L.ACT_SWISH, act_grad_z=z, drop=(0.15, 7))}[mode]()
fn = lambda: ops.gemm_panel(x, wp, y, N, f, d, e)
assert "diag" in L.LIB_PATH, "run with SMX_LIB=summarymixing_amd/libsmx_diag.so"
lib = L.lib(); lib.smx_debug_set_timing_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3): fn()
nb = (N + 127) // 128
buf = torch.zeros(nb * 8 * 16, dtype=torch.int64, device="cuda")
lib.smx_debug_set_timing_buffer(ctypes.c_void_p(buf.data_ptr()))
fn(); torch.cuda.synchronize()
lib.smx_debug_set_timing_buffer(None)
s = buf.view(nb, 8, 16).cpu().double()
t0 = s[:, :, 0].min()
rounds = (f // 64 + 7) // 8
print(f"panel {mode} N={N} K={d} M={f}: kernel span {float(s.max() - t0) / 100:.1f} us (100 MHz ticks); workgroups {nb}")
first = s[:, 0, 0] - t0
print(f"  workgroup start: first wave of blocks: <1us: {int((first < 100).sum())}, later: {int((first >= 100).sum())} (median {float(first[first >= 100].median()) / 100 if (first >= 100).any() else 0:.1f} us)")
print(f"  panel load (start -> barrier): mean {float((s[:, :, 1] - s[:, :, 0]).mean()) / 100:.2f} us")
prev = s[:, :, 1]
for r in range(rounds):
    ml, ep = s[:, :, 2 + 2 * r], s[:, :, 3 + 2 * r]
    for grp, sl in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
        print(f"  round {r} {grp}: main loop {float((ml - prev)[:, sl].mean()) / 100:6.2f} us   epilogue {float((ep - ml)[:, sl].mean()) / 100:6.2f} us")
    prev = ep
life = s[:, :, 1 + 2 * rounds] - s[:, :, 0]
print(f"  wave lifetime mean {float(life.mean()) / 100:.1f} us, max {float(life.max()) / 100:.1f}")
b0 = s[0]
print("  block 0 timeline (us from kernel start), per wave: " )
for w in range(8):
    print("    w%d: " % w + " ".join(f"{float(v - t0) / 100:6.1f}" for v in b0[w, :2 + 2 * rounds]))
