"""Standalone ParallelLinear on the GPU against the reference-generated fixture g3_parallel_linear (3-D and 4-D inputs,
combine_out_dims both ways; speechbrain/lobes/models/VanillaNN.py:100-117) and, for the gradients, against autograd of
the oracle on the same weights."""
import pytest
import torch

from oracle import smx_oracle as O
from tests import _golden as G
from tests._util import TOL, rel_err

pytestmark = pytest.mark.gpu


def _module(sd, prefix, combine):
    from summarymixing_amd.lobes.models.VanillaNN import ParallelLinear
    w = sd[prefix + "weights"]
    H, f, h = w.shape
    m = ParallelLinear(H * h, input_size=H * f, n_split=H, combine_out_dims=combine)
    m.load_state_dict({"weights": w, "biases": sd[prefix + "biases"]}, strict=True)
    return m.cuda()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", ["x3_combine", "x4_nocombine", "x3_nocombine"])
def test_parallel_linear_golden(case, dtype):
    meta, a, sd, _ = G.load("g3_parallel_linear")
    prefix, xk, yk, combine = {"x3_combine": ("a.", "x3", "y3", True), "x4_nocombine": ("b.", "x4", "y4_nocombine", False),
                               "x3_nocombine": ("b.", "x3", "y3_nocombine", False)}[case]
    m = _module(sd, prefix, combine)
    x = a[xk].cuda().to(dtype).requires_grad_(True)
    y = m(x)
    assert tuple(y.shape) == tuple(a[yk].shape)          # 4-D (B,T,H,h) when the head dim is kept (VanillaNN.py:114-115)
    ftol, gtol = TOL[dtype]
    assert rel_err(y, a[yk]) <= ftol, rel_err(y, a[yk])
    # gradients: L = sum(y * r) against autograd of the oracle restatement
    torch.manual_seed(3)
    r = torch.randn(a[yk].shape)
    w = sd[prefix + "weights"].clone().requires_grad_(True)
    b = sd[prefix + "biases"].clone().requires_grad_(True)
    xo = a[xk].clone().requires_grad_(True)
    (O.parallel_linear(xo, w, b, combine) * r).sum().backward()
    (y.float() * r.cuda()).sum().backward()
    assert rel_err(x.grad, xo.grad) <= gtol, rel_err(x.grad, xo.grad)
    assert rel_err(m.weights.grad, w.grad) <= gtol, rel_err(m.weights.grad, w.grad)
    assert rel_err(m.biases.grad, b.grad) <= gtol, rel_err(m.biases.grad, b.grad)
