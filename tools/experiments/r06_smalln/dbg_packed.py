import sys, torch
sys.path.insert(0, "/root/repo")
from summarymixing_amd import functional as F, ops
from summarymixing_amd.trainer import FlatAdamW
from tests.test_trainer_gpu import _small_panel_encoder, _packed_images_current
F._PANEL_MIN_ROWS = 128
dtype, d = torch.bfloat16, 256
enc = _small_panel_encoder(d)
opt = FlatAdamW(enc, lr=3e-2, compute_dtype=dtype)
opt.use_device_step_counter(True)
x = torch.randn(4, 200, d, device="cuda").to(dtype); r = torch.randn(4, 200, d, device="cuda").to(dtype)
def step():
    opt.zero_grad(); (enc(x)[0].float() * r.float()).sum().backward(); opt.step()
step(); print("after step", len(F._packed), _packed_images_current(enc, dtype) if False else "")
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.no_grad(): enc(x)
    print("after warm fwd", len(F._packed), [(k[1], v[1]) for k, v in F._packed.items()][:3])
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=s):
        step()
    print("after capture", len(F._packed), [(k[1], v[1]) for k, v in F._packed.items()][:3])
    for _ in range(2): opt.replay(g1)
    with torch.no_grad(): enc(x)
    print("after eager", len(F._packed), [(k[1], v[1]) for k, v in F._packed.items()][:3])
    print(_packed_images_current(enc, dtype))
