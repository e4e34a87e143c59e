"""FlatAdamW on the GPU: the HIP backward writing into the flat gradient buffer, per-layer buckets
all-reduced as each layer's backward ends, global-norm clip + smx_adamw_step - with world > 1 and over RCCL.

  * 2 ranks sharing cuda:0 over gloo (the box has ONE GPU; on a node the same code is one rank per GPU over RCCL):
    utterance-sharded step == single-process step on the concatenated batch (gradients and updated weights);
  * a single-rank `nccl` (= RCCL) group with SMX_FORCE_ALLREDUCE=1: the bucketed collective path gives bit-for-bit the
    weights of the collective-free step;
  * optimizer housekeeping: bf16 shadows follow load_state_dict, state_dict round trip, torch's zero_grad(set_to_none)
    does not lose gradients, a non-finite gradient norm skips the update."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("ln_fuse_mode")]   # (both LayerNorm dispatches: tests/conftest.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COMMON = r'''
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
'''

DP_WORKER = COMMON + r'''
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dtype = torch.bfloat16 if os.environ["SMX_DTYPE"] == "bf16" else torch.float32
torch.cuda.set_device(0)
# reference: ONE process, the whole batch, no collective (built before the process group exists); the DP job averages the
# shard gradients (1 / world), so the reference loss carries the same factor
ref = model()
ropt = FlatAdamW(ref, lr=1e-2, max_grad_norm=5.0, compute_dtype=dtype)
assert ropt.world == 1 and not ropt._collective
one_step(ref, ropt, X.to(dtype), PAD, (R / world).to(dtype), False)
ref_g = ropt.flat_g.clone()

dist.init_process_group("gloo", rank=rank, world_size=world)
enc = model()
reduce = os.environ.get("SMX_REDUCE", "allreduce")
gdt = torch.bfloat16 if os.environ.get("SMX_GRAD_DTYPE") == "bf16" else torch.float32
opt = FlatAdamW(enc, lr=1e-2, max_grad_norm=5.0, compute_dtype=dtype, reduce=reduce, grad_dtype=gdt)
assert opt.world == world and opt._collective
hooks(enc, opt)
sl = slice(rank * B // world, (rank + 1) * B // world)            # this rank's utterances
if os.environ.get("SMX_ACCUM") == "1":
    # gradient accumulation: the rank's shard as two micro-batches of one utterance, bucket hooks silent inside no_sync()
    xs, ps, rs = X[sl].to(dtype), PAD[sl], R[sl].to(dtype)
    opt.zero_grad()
    with opt.no_sync():
        y, _ = enc(xs[:1], src_key_padding_mask=ps[:1])
        y.backward(rs[:1])
        assert not opt._pending and not opt._reduced, "a collective was launched inside no_sync()"
    y, _ = enc(xs[1:], src_key_padding_mask=ps[1:])
    y.backward(rs[1:])
    tail(enc, opt)
    opt.step()
else:
    one_step(enc, opt, X[sl].to(dtype), PAD[sl], R[sl].to(dtype), True)

def close(a, b, what, tol):
    err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
    assert err < tol, f"rank {rank} {what}: rel err {err:.3e}"
tol = 2e-5 if dtype == torch.float32 else 2e-2
if gdt == torch.bfloat16:
    tol = max(tol, 1e-2)
if reduce == "rs_ag":                                             # this rank's shards of the reduce-scattered buckets
    W = world
    for a, b in [opt.param_range(list(l.parameters())) for l in enc.layers]:
        sa, sb = opt._shard(a, b)
        close(opt._shard_g[a // W:b // W] / world, ref_g[sa:sb], "reduce-scattered gradient shard", tol)
else:
    close(opt.flat_g / world, ref_g, "all-reduced gradient", tol)
# (Adam's first step moves every weight by ~lr * sign(g): with bf16 activations a near-zero gradient element may flip sign
#  between the sharded and the full-batch run, so the bf16 weights can only agree to ~2 lr = 2e-2 of max|w| ~ 1)
close(opt.flat_p, ropt.flat_p, "updated weights", tol if (dtype == torch.float32 and gdt == torch.float32) else 3e-2)
# every rank ends with the same weights, bit for bit
mine = opt.flat_p.cpu()
other = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(other, mine)
assert all(torch.equal(o, mine) for o in other), "ranks diverged"
dist.barrier()
print(f"rank {rank} OK")
'''

NCCL_WORKER = COMMON + r'''
torch.cuda.set_device(0)
dtype = torch.bfloat16
ref = model()
ropt = FlatAdamW(ref, lr=1e-2, max_grad_norm=5.0, compute_dtype=dtype)
assert not ropt._collective
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
enc = model()
reduce = os.environ.get("SMX_REDUCE", "allreduce")
gdt = torch.bfloat16 if os.environ.get("SMX_GRAD_DTYPE") == "bf16" else torch.float32
opt = FlatAdamW(enc, lr=1e-2, max_grad_norm=5.0, compute_dtype=dtype, reduce=reduce, grad_dtype=gdt)
assert opt._collective and opt.world == 1, "SMX_FORCE_ALLREDUCE=1 must switch the RCCL bucket path on"
hooks(enc, opt)
opt.measure_comm(True)
for it in range(3):
    one_step(ref, ropt, X.to(dtype), PAD, R.to(dtype), False)
    one_step(enc, opt, X.to(dtype), PAD, R.to(dtype), True)
    if it == 0 and gdt == torch.bfloat16:
        # gradients crossed the wire in bf16: one rounding (2^-9 relative per element) of what the update saw (checked on
        # the first step: afterwards the two models' weights differ by that rounding)
        got = opt._shard_g if reduce == "rs_ag" else opt.flat_g
        ok = (got - ropt.flat_g).abs() <= ropt.flat_g.abs() * 2.0 ** -8 + 1e-30
        assert bool(ok.all()), "bf16 gradient image off by more than one rounding"
torch.cuda.synchronize()
assert opt.comm_exposed_ms() >= 0.0 and len(opt._exposed) == 3
if gdt == torch.float32:
    # reduce-scatter / all-gather over one rank move every bucket through the shard buffers: still bit for bit
    assert torch.equal(opt.flat_g, ropt.flat_g), "gradients differ"
    assert torch.equal(opt.flat_p, ropt.flat_p), "weights differ"
    assert torch.equal(opt.shadow, ropt.shadow)
else:
    assert torch.equal(opt.shadow, opt.flat_p.bfloat16()), "bf16 shadows do not follow the gathered weights"
    upd = (opt.flat_p - ropt.flat_p).pow(2).mean().sqrt().item() / (3 * 1e-2)        # against three steps of size ~lr
    assert upd < 0.2, f"weights after three bf16-gradient steps: rms difference {upd:.3f} of the update size"
dist.barrier()
dist.destroy_process_group()
print("rank 0 OK")
'''


GRAPH_WORKER = COMMON + r'''
# data parallel + hipGraph: [forward + backward] and [update] captured separately, ONE eager all-reduce between them
torch.cuda.set_device(0)
dtype = torch.bfloat16
ref = model()
ropt = FlatAdamW(ref, lr=1e-2, max_grad_norm=5.0, compute_dtype=dtype)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
enc = model()
opt = FlatAdamW(enc, lr=1e-2, max_grad_norm=5.0, compute_dtype=dtype)
assert opt._collective
x, r = X.to(dtype), R.to(dtype)
def fwd_bwd():
    opt.zero_grad()
    y, _ = enc(x, src_key_padding_mask=PAD)
    y.backward(r)
opt.use_device_step_counter(True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fwd_bwd(); opt.all_reduce_all(); opt.update_only()              # warm-up step 1 (allocates this stream's workspaces)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(ga):
    fwd_bwd()
with torch.cuda.graph(gb):
    opt.update_only()
for _ in range(2):                                                   # steps 2 and 3 as graph replays
    ga.replay(); opt.all_reduce_all(); gb.replay()
for _ in range(3):
    one_step(ref, ropt, x, PAD, r, False)
torch.cuda.synchronize()
assert torch.equal(opt.flat_g, ropt.flat_g), "gradients differ"
# (the captured update takes AdamW's bias correction from the device step counter - powf on the device - the eager one
#  from the host: the weights agree to fp32 rounding, not bit for bit)
err = (opt.flat_p - ropt.flat_p).abs().max().item() / ropt.flat_p.abs().max().item()
assert err < 1e-6, f"weights differ: {err:.3e}"
dist.barrier()
dist.destroy_process_group()
print("rank 0 OK")
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(script, world, extra):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SMX_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        procs.append(subprocess.Popen([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    for rank, p in enumerate(procs):
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n[timeout]"
        assert p.returncode == 0 and f"rank {rank} OK" in out, f"rank {rank} failed:\n{out[-3000:]}"


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_two_ranks_hip_backward_through_buckets_equals_single_process(dtype):
    _launch(DP_WORKER, 2, {"SMX_DTYPE": dtype})


@pytest.mark.parametrize("reduce", ["allreduce", "rs_ag"])
def test_two_ranks_gradient_accumulation_under_no_sync(reduce):
    """Gradient accumulation under data parallelism (FlatAdamW.no_sync, the role of DDP's no_sync in the reference's fit_batch):
    every rank runs its shard as two micro-batches, the bucket hooks launch nothing during the first and reduce the accumulated
    sums behind the second; gradients and updated weights equal the single-process step on the whole batch."""
    _launch(DP_WORKER, 2, {"SMX_DTYPE": "fp32", "SMX_REDUCE": reduce, "SMX_ACCUM": "1"})


@pytest.mark.parametrize("reduce,grad_dtype", [("rs_ag", "fp32"), ("allreduce", "bf16"), ("rs_ag", "bf16")])
def test_two_ranks_reduce_modes(reduce, grad_dtype):
    """The same 2-rank job with reduce-scatter + sharded AdamW + all-gather and / or bf16 gradients on the wire."""
    _launch(DP_WORKER, 2, {"SMX_DTYPE": "fp32", "SMX_REDUCE": reduce, "SMX_GRAD_DTYPE": grad_dtype})


def test_single_rank_rccl_bucket_path_is_bit_identical():
    _launch(NCCL_WORKER, 1, {"SMX_FORCE_ALLREDUCE": "1"})


@pytest.mark.parametrize("reduce,grad_dtype", [("rs_ag", "fp32"), ("allreduce", "bf16"), ("rs_ag", "bf16")])
def test_single_rank_rccl_reduce_modes(reduce, grad_dtype):
    """FlatAdamW(reduce="rs_ag") / grad_dtype=bfloat16 over RCCL (one rank: the collectives are identities, the staging
    buffers, casts, shard arithmetic and the all-gather of the weights are the real ones; world 2 runs on gloo in
    tests/test_trainer_dist.py)."""
    _launch(NCCL_WORKER, 1, {"SMX_FORCE_ALLREDUCE": "1", "SMX_REDUCE": reduce, "SMX_GRAD_DTYPE": grad_dtype})


def test_dp_hipgraph_split_equals_eager_steps():
    """hipGraph under data parallelism (bench.py --graph with world > 1): two captured halves with one RCCL all-reduce of
    the flat gradient buffer between them reproduce three eager single-process steps (gradients bit for bit)."""
    _launch(GRAPH_WORKER, 1, {"SMX_FORCE_ALLREDUCE": "1"})


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` self-launches N ranks; with fewer GPUs visible it must fail loudly, never run 1 rank."""
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "GPU(s) visible" in (p.stderr + p.stdout), (p.returncode, p.stderr[-500:])
    assert '"n_gpus"' not in p.stdout


def _small(dtype=torch.bfloat16):
    from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoder
    from summarymixing_amd.trainer import FlatAdamW
    torch.manual_seed(5)
    d = 64
    enc = ConformerEncoder(1, d, 128, 4, kernel_size=31, activation="swish", dropout=0.0, attention_type="SummaryMixing",
                           local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], mode="SummaryMixing-fast").cuda()
    opt = FlatAdamW(enc, lr=1e-2, max_grad_norm=5.0, compute_dtype=dtype)
    x = torch.randn(3, 50, d, device="cuda").to(dtype)
    r = torch.randn(3, 50, d, device="cuda").to(dtype)
    return enc, opt, x, r


def test_shadows_follow_load_state_dict_and_optimizer_state_round_trips():
    enc, opt, x, r = _small()
    for _ in range(2):
        opt.zero_grad()
        enc(x)[0].backward(r)
        opt.step()
    sd_model = {k: v.clone() for k, v in enc.state_dict().items()}
    sd_opt = opt.state_dict()
    assert sd_opt["step"] == 2
    with torch.no_grad():
        y_ref = enc(x)[0].clone()
    # a third step from the checkpointed state, twice: directly, and after clobbering + restoring everything
    opt.zero_grad(); enc(x)[0].backward(r); opt.step()
    p_after = opt.flat_p.clone()
    with torch.no_grad():
        for p in enc.parameters():
            p.mul_(0.5)                                   # in-place change behind the optimizer's back
    opt.exp_avg.zero_(); opt.exp_avg_sq.fill_(7.0); opt.step_count = 99
    enc.load_state_dict(sd_model)                         # post-hook: the bf16 shadows must follow
    opt.load_state_dict(sd_opt)
    with torch.no_grad():
        assert torch.equal(enc(x)[0], y_ref), "bf16 shadows are stale after load_state_dict"
    opt.zero_grad(); enc(x)[0].backward(r); opt.step()
    assert torch.equal(opt.flat_p, p_after), "resume is not bit-identical"


def test_module_zero_grad_set_to_none_keeps_gradients():
    enc, opt, x, r = _small(torch.float32)
    opt.zero_grad(); enc(x)[0].backward(r)
    g_ref = opt.flat_g.clone()
    opt.step()
    p_ref = opt.flat_p.clone()
    enc2, opt2, _, _ = _small(torch.float32)
    enc2.zero_grad(set_to_none=True)                      # detaches every .grad from the flat buffer
    enc2(x)[0].backward(r)
    opt2.step()                                           # must fold the fresh gradient tensors back, not apply decay only
    assert torch.equal(opt2.flat_g, g_ref) and torch.equal(opt2.flat_p, p_ref)
    assert all(p.grad.data_ptr() == opt2.flat_g.data_ptr() + 4 * o for p, o in zip(opt2.params, opt2.offs))


def test_non_finite_gradient_norm_skips_the_update():
    enc, opt, x, r = _small()
    opt.zero_grad(); enc(x)[0].backward(r); opt.step()
    p0, m0, sh0 = opt.flat_p.clone(), opt.exp_avg.clone(), opt.shadow.clone()
    opt.zero_grad(); enc(x)[0].backward(r)
    opt.flat_g[123] = float("nan")                        # one poisoned gradient element (a bad batch)
    opt.step()
    assert opt.skipped_steps() == 1
    assert torch.equal(opt.flat_p, p0) and torch.equal(opt.exp_avg, m0) and torch.equal(opt.shadow, sh0)
    opt.zero_grad(); enc(x)[0].backward(r); opt.flat_g[5] = float("inf"); opt.step()
    assert opt.skipped_steps() == 2 and torch.equal(opt.flat_p, p0)
    opt.zero_grad(); enc(x)[0].backward(r); opt.step()    # a clean step still updates
    assert opt.skipped_steps() == 2 and not torch.equal(opt.flat_p, p0) and torch.isfinite(opt.flat_p).all()


def test_packed_weight_images_follow_the_optimizer():
    """The panel GEMM's packed images (functional.wpacked) are re-packed after EVERY optimizer step and checkpoint load - eagerly
    and inside a captured hipGraph (whose replays must contain the pack launches): after each step's forward every cached image
    equals a fresh pack of the current bf16 shadow, and graph replays equal eager steps."""
    from summarymixing_amd import functional as F, ops
    from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoder
    from summarymixing_amd.trainer import FlatAdamW
    dtype, d = torch.bfloat16, 256
    old_rows = F._PANEL_MIN_ROWS
    F._PANEL_MIN_ROWS = 128                       # (the panel path from 128 rows: a small batch exercises it)

    def run(panel, graph):
        F._PANEL = panel
        torch.manual_seed(5)
        enc = ConformerEncoder(1, d, 1024, 4, kernel_size=31, activation="swish", dropout=0.0, attention_type="SummaryMixing",
                               local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], mode="SummaryMixing-fast").cuda()
        opt = FlatAdamW(enc, lr=3e-2, max_grad_norm=5.0, compute_dtype=dtype)
        x = torch.randn(4, 200, d, device="cuda").to(dtype)
        r = torch.randn(4, 200, d, device="cuda").to(dtype)

        def step():
            opt.zero_grad()
            y, _ = enc(x)
            (y.float() * r.float()).sum().backward()
            opt.step()

        def check_images():
            n = 0
            for key, ent in list(F._packed.items()):
                prm, tr = ent[0](), key[1]
                if prm is None or not any(prm is q for q in enc.parameters()):
                    continue
                W = F.wcast(prm, dtype)
                W = W.view(W.shape[0], -1) if W.dim() != 2 else W
                fresh = ops.weight_pack_slices(W, key[2], bool(tr)) if len(key) > 2 else ops.weight_pack(W, bool(tr), ent[4])
                assert torch.equal(ent[2], fresh), "stale packed image"
                n += 1
            return n

        step(); step()                             # eager: images packed singly, then by the grouped launch
        if panel:
            with torch.no_grad():
                enc(x)                             # (a forward after the update: the forward images are current again)
            assert check_images() >= 3
        if graph:
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                step()
                with torch.cuda.graph(g, stream=s):
                    step()
            torch.cuda.current_stream().wait_stream(s)
            for _ in range(3):
                g.replay()
        else:
            for _ in range(5):
                step()
        torch.cuda.synchronize()
        return torch.cat([p.detach().float().flatten() for p in enc.parameters()])

    try:
        eager = run(True, False)
        graphed = run(True, True)
        tiled_eager, tiled_graphed = run(False, False), run(False, True)
    finally:
        F._PANEL, F._PANEL_MIN_ROWS = True, old_rows
    # replays whose graph lacked the pack launches would run steps 4-7 on the images of step 3: far outside this bar at lr 3e-2
    # (the tiled path sets the bar: graph replay against eager steps of the same kernels)
    assert float((graphed - eager).abs().max()) <= max(1e-6, 2.0 * float((tiled_graphed - tiled_eager).abs().max()))


def _packed_images_current(enc, dtype):
    """Every cached packed image of `enc`'s parameters equals a fresh pack of the CURRENT bf16 shadow and bias."""
    from summarymixing_amd import functional as F, ops
    n = 0
    for key, ent in list(F._packed.items()):
        prm, tr = ent[0](), key[1]
        if prm is None or not any(prm is q for q in enc.parameters()):
            continue
        W = F.wcast(prm, dtype)
        W = W.view(W.shape[0], -1) if W.dim() != 2 else W
        fresh = ops.weight_pack_slices(W, key[2], bool(tr)) if len(key) > 2 else ops.weight_pack(W, bool(tr), ent[4])   # (key[2]: K-slice images)
        assert torch.equal(ent[2], fresh), "stale packed image"
        n += 1
    return n


def _small_panel_encoder(d=256):
    from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoder
    torch.manual_seed(5)
    return ConformerEncoder(1, d, 4 * d, 4, kernel_size=31, activation="swish", dropout=0.0, attention_type="SummaryMixing",
                            local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], mode="SummaryMixing-fast").cuda()


def test_packed_images_follow_a_second_optimizer_over_the_same_module():
    """Round-5 advisor finding: a second FlatAdamW over one module installs NEW shadow / flat buffers; the cached images must be
    packed from those, not from the first optimizer's frozen buffers."""
    from summarymixing_amd import functional as F
    from summarymixing_amd.trainer import FlatAdamW
    dtype, d = torch.bfloat16, 256
    old_rows, F._PANEL_MIN_ROWS = F._PANEL_MIN_ROWS, 128
    try:
        enc = _small_panel_encoder(d)
        x = torch.randn(4, 200, d, device="cuda").to(dtype)
        r = torch.randn(4, 200, d, device="cuda").to(dtype)

        def step(opt):
            opt.zero_grad()
            (enc(x)[0].float() * r.float()).sum().backward()
            opt.step()

        opt1 = FlatAdamW(enc, lr=3e-2, compute_dtype=dtype)
        step(opt1); step(opt1)
        opt2 = FlatAdamW(enc, lr=3e-2, compute_dtype=dtype)
        step(opt2); step(opt2)
        with torch.no_grad():
            enc(x)
        assert _packed_images_current(enc, dtype) >= 3
        for ent in F._packed.values():                 # nothing keeps the first optimizer's shadow alive through the cache
            assert ent[0]() is None or ent[3].untyped_storage().data_ptr() != opt1.shadow.untyped_storage().data_ptr()
    finally:
        F._PANEL_MIN_ROWS = old_rows


def test_packed_images_in_graphs_captured_after_a_forward_only_warm_up_and_captured_twice():
    """Round-5 advisor finding: the pack launches must be INSIDE every capture (also one whose epoch stamp happened to be current
    when it began, and a second capture of the same step), and the first eager forward after replays must re-pack."""
    from summarymixing_amd import functional as F
    from summarymixing_amd.trainer import FlatAdamW
    dtype, d = torch.bfloat16, 256
    old_rows, F._PANEL_MIN_ROWS = F._PANEL_MIN_ROWS, 128
    try:
        def run(graphs):
            enc = _small_panel_encoder(d)
            opt = FlatAdamW(enc, lr=3e-2, compute_dtype=dtype)
            opt.use_device_step_counter(True)
            x = torch.randn(4, 200, d, device="cuda").to(dtype)
            r = torch.randn(4, 200, d, device="cuda").to(dtype)

            def step():
                opt.zero_grad()
                (enc(x)[0].float() * r.float()).sum().backward()
                opt.step()

            step()
            if not graphs:
                for _ in range(6):
                    step()
            else:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    with torch.no_grad():
                        enc(x)                         # forward-only warm-up: every forward image carries the CURRENT epoch now
                    g1 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g1, stream=s):
                        step()
                    for _ in range(2):
                        opt.replay(g1)
                    with torch.no_grad():
                        enc(x)                         # eager forward after replays: re-packs
                    assert _packed_images_current(enc, dtype) >= (3 if F._PANEL else 0)
                    g2 = torch.cuda.CUDAGraph()        # a second capture of the same step
                    with torch.cuda.graph(g2, stream=s):
                        step()
                    for _ in range(4):
                        opt.replay(g2)
                torch.cuda.current_stream().wait_stream(s)
            opt.use_device_step_counter(False)
            torch.cuda.synchronize()
            return torch.cat([p.detach().float().flatten() for p in enc.parameters()])

        eager, graphed = run(False), run(True)
        F._PANEL = False
        tiled_eager, tiled_graphed = run(False), run(True)
    finally:
        F._PANEL, F._PANEL_MIN_ROWS = True, old_rows
    # replays of a graph without pack launches run on frozen images: far outside this bar at lr 3e-2
    assert float((graphed - eager).abs().max()) <= max(1e-6, 2.0 * float((tiled_graphed - tiled_eager).abs().max()))
