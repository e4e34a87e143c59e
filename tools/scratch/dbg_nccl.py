
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["SMX_ROOT"])
from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoder
from summarymixing_amd.trainer import FlatAdamW

def model(dtype_seed=0):
    torch.manual_seed(7)
    d = 64
    enc = ConformerEncoder(2, d, 128, 4, kernel_size=31, activation="swish", dropout=0.0, attention_type="SummaryMixing",
                           local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], mode="SummaryMixing-fast")
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if p.dim() > 1:
                torch.nn.init.xavier_normal_(p)
            elif "bias" in n:
                p.normal_(0, 0.05)
    return enc.cuda()

def hooks(enc, opt):
    for layer in enc.layers:
        rng = opt.param_range(list(layer.parameters()))
        layer._on_bwd_done = (lambda r=rng: opt.reduce_bucket_async(*r))

def tail(enc, opt):
    first = opt.param_range(list(enc.layers[0].parameters()))[0]
    last = opt.param_range(list(enc.layers[-1].parameters()))[1]
    if first > 0:
        opt.reduce_bucket_async(0, first)
    if last < opt.total:
        opt.reduce_bucket_async(last, opt.total)

def one_step(enc, opt, x, pad, r, collective):
    opt.zero_grad()
    y, _ = enc(x, src_key_padding_mask=pad)
    y.backward(r)
    if collective:
        tail(enc, opt)
    opt.step()

g = torch.Generator().manual_seed(3)
B, T, d = 4, 120, 64
X = torch.randn(B, T, d, generator=g).cuda()
R = torch.randn(B, T, d, generator=g).cuda()
lens = torch.tensor([T, 77, 101, 64])
PAD = (torch.arange(T)[None] < lens[:, None]).cuda()

torch.cuda.set_device(0)
dtype = torch.bfloat16
ref = model()
ropt = FlatAdamW(ref, lr=1e-2, max_grad_norm=5.0, compute_dtype=dtype)
assert not ropt._collective
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
enc = model()
opt = FlatAdamW(enc, lr=1e-2, max_grad_norm=5.0, compute_dtype=dtype)
assert opt._collective and opt.world == 1, "SMX_FORCE_ALLREDUCE=1 must switch the RCCL bucket path on"
hooks(enc, opt)
for it in range(3):
    one_step(ref, ropt, X.to(dtype), PAD, R.to(dtype), False)
    one_step(enc, opt, X.to(dtype), PAD, R.to(dtype), True)
    torch.cuda.synchronize()
    print(it, "g", torch.equal(opt.flat_g, ropt.flat_g), "p", torch.equal(opt.flat_p, ropt.flat_p), "m", torch.equal(opt.exp_avg, ropt.exp_avg),
          "v", torch.equal(opt.exp_avg_sq, ropt.exp_avg_sq), "ss", opt._sumsq.item(), ropt._sumsq.item(), "clip", opt._clip.tolist(), ropt._clip.tolist(),
          "maxdiff p", (opt.flat_p - ropt.flat_p).abs().max().item(), "g", (opt.flat_g - ropt.flat_g).abs().max().item())
torch.cuda.synchronize()
assert torch.equal(opt.flat_g, ropt.flat_g), "gradients differ"
assert torch.equal(opt.flat_p, ropt.flat_p), "weights differ"
assert torch.equal(opt.shadow, ropt.shadow)
dist.barrier()
dist.destroy_process_group()
print("rank 0 OK")
