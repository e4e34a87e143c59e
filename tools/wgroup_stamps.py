"""Cycle stamps of the grouped wgrad kernel: per workgroup total cycles, cycles spent in (vmcnt wait + barrier), realtime."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd import _lib as L, functional as F
rows = 64000
shapes = [(1024, 256), (256, 1024), (1024, 256), (256, 1024), (512, 256), (256, 512), (512, 256), (256, 256)]
ops_ = [((torch.randn(rows, M, device="cuda") * 0.5).bfloat16(), torch.randn(rows, K, device="cuda").bfloat16(),
         torch.zeros(M, K, device="cuda"), torch.zeros(M, device="cuda")) for M, K in shapes]
def run():
    for dz, x, gW, gb in ops_:
        F._wgrad(dz, x, gW, rows, dz.shape[1], x.shape[1], gb)
    F.flush_deferred()
for _ in range(3):
    run()
buf = torch.zeros(4096 * 4, dtype=torch.int64, device="cuda")
lib = L.lib()
lib.smx_debug_set_wgroup_timing_buffer.argtypes = [ctypes.c_void_p]
lib.smx_debug_set_wgroup_timing_buffer(ctypes.c_void_p(buf.data_ptr()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
lib.smx_debug_set_wgroup_timing_buffer(None)
b = buf.view(-1, 4).cpu()
b = b[b[:, 3] > 0].double()
print(f"event time (kernel + reduce) {e0.elapsed_time(e1) * 1e3:.1f} us; workgroups {len(b)}")
tot, wait, real, nit = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
print(f"loop cycles per WG: mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}; per K step {(tot / nit).mean():.0f}")
print(f"wait+barrier cycles: mean {wait.mean():.0f} ({(wait / tot).mean() * 100:.1f} % of the loop); per K step {(wait / nit).mean():.0f}")
print(f"realtime ticks (100 MHz): mean {real.mean():.0f} -> loop {real.mean() / 100:.1f} us, shader clock ~ {(tot / real).mean() * 100:.0f} MHz")
