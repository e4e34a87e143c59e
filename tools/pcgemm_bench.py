#!/usr/bin/env python3
"""Producer / consumer GEMM experiment (tools/experiments/pcgemm2.hip) against smx_gemm on the FFN up-projection shape.
    python tools/pcgemm_bench.py [N] [K] [M]"""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import time_kernel
from summarymixing_amd import _lib as L, ops
so = os.path.join(ROOT, "tools", "experiments", "libpcgemm.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffast-math",
                           "-fno-finite-math-only", "-I" + os.path.join(ROOT, "summarymixing_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "experiments", "pcgemm2.hip"), "-o", so])
lib = ctypes.CDLL(so)
vp, i64, ci = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
lib.pc_gemm.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, i64, ci, ci, ci, ci, ci, vp]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 256
M = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
torch.manual_seed(0)
x = torch.randn(N, K, device="cuda").bfloat16()
w = (torch.randn(M, K, device="cuda") * 0.05).bfloat16()
b = torch.randn(M, device="cuda")
z0 = torch.empty(N, M, device="cuda", dtype=torch.bfloat16); h0 = torch.empty_like(z0)
z1 = torch.zeros_like(z0); h1 = torch.zeros_like(z0)
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ref = lambda: ops.gemm(L.GEMM_NT, x, w, h0, N, M, K, ops.epilogue(bias=b, act=L.ACT_SWISH, z=z0))
def pc(blocks=256, ablate=0):
    rc = lib.pc_gemm(x.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), z1.data_ptr(), M, h1.data_ptr(), M, N, M, K, blocks, ablate, st())
    assert rc == 0, rc
ref(); pc(); torch.cuda.synchronize()
ez = (z1.float() - z0.float()).abs().max().item() / z0.float().abs().max().item()
eh = (h1.float() - h0.float()).abs().max().item() / h0.float().abs().max().item()
print(f"check: Z rel err {ez:.2e}  H rel err {eh:.2e}", flush=True)
nb = (N * K + M * K + 2 * N * M) * 2
t = time_kernel(ref, 20, 3); print(f"smx_gemm (tiled)      {t*1e6:7.1f} us  {nb/t/1e9:6.0f} GB/s", flush=True)
for blocks in (256, 248, 240):
    t = time_kernel(lambda: pc(blocks), 20, 3); print(f"pc_gemm blocks={blocks:3d}    {t*1e6:7.1f} us  {nb/t/1e9:6.0f} GB/s", flush=True)
for ab, name in ((1, "no epilogue work"), (2, "no MFMA"), (4, "no operand loads"), (3, "barriers + loads only"), (6, "epilogue + dump only"), (7, "barriers + dump only"), (5, "MFMA + dump only")):
    t = time_kernel(lambda: pc(256, ab), 20, 3); print(f"pc_gemm ablate={ab} ({name:22s}) {t*1e6:7.1f} us", flush=True)
