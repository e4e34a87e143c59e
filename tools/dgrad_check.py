import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as tF
from summarymixing_amd import ops
torch.manual_seed(0)
B, T, Fq, C, O = 1, 8, 8, 64, 32
T2, F2 = (T + 1) // 2, (Fq + 1) // 2
W = torch.randn(O, C, 3, 3) * 0.1
dy = torch.randn(B, T2, F2, O)
x = torch.zeros(B, C, T, Fq, dtype=torch.float64, requires_grad=True)
y = tF.conv2d(tF.pad(x, (1, 1, 1, 1), mode="reflect"), W.bfloat16().double(), None, stride=2)
(y * dy.bfloat16().double().permute(0, 3, 1, 2)).sum().backward()
ref = x.grad.permute(0, 2, 3, 1).float()
wg = W.permute(0, 2, 3, 1).reshape(O, 9 * C).cuda().bfloat16().contiguous()
dy2 = dy.cuda().bfloat16().reshape(B * T2 * F2, O).contiguous()
got = ops.conv2d_s2_dgrad(dy2, wg, B, T, Fq, C).float().cpu()
err = (got - ref).abs()
print("max err", err.max().item(), "ref max", ref.abs().max().item())
print("err by t:", err.amax(dim=(0, 2, 3)))
print("err by f:", err.amax(dim=(0, 1, 3)))
print("err by c:", err.amax(dim=(0, 1, 2)))
print("got[0,2,2,:8]", got[0, 2, 2, :8]); print("ref[0,2,2,:8]", ref[0, 2, 2, :8])
print("ratio", (got / ref)[0, 2, 2, :8])
# which taps does the kernel include?  least squares of got on the 9 single-tap references
refs = []
for dt in range(3):
    for df in range(3):
        Wm = torch.zeros_like(W); Wm[:, :, dt, df] = W[:, :, dt, df]
        x2 = torch.zeros(B, C, T, Fq, dtype=torch.float64, requires_grad=True)
        y2 = tF.conv2d(tF.pad(x2, (1, 1, 1, 1), mode="reflect"), Wm.bfloat16().double(), None, stride=2)
        (y2 * dy.bfloat16().double().permute(0, 3, 1, 2)).sum().backward()
        refs.append(x2.grad.permute(0, 2, 3, 1).float().reshape(-1))
A = torch.stack(refs, 1)
sol = torch.linalg.lstsq(A, got.reshape(-1, 1)).solution.view(3, 3)
print("tap coefficients (dt rows, df cols):\n", sol)
