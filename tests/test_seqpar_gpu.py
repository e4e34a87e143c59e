"""Sequence-parallel mode (summarymixing_amd/sequence_parallel.py): the time axis of one batch sharded
over 2 ranks must reproduce the unsharded encoder - outputs, input gradients and (summed) parameter gradients.

The GPU box has ONE device, so the two ranks share cuda:0 and talk over gloo (host-staged); on a multi-GPU node the same
code runs one rank per GPU over RCCL.  Every rank first runs the unsharded encoder itself as the reference."""
import os
import subprocess
import sys
import socket

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["SMX_ROOT"])
from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoder
from summarymixing_amd import sequence_parallel as SP

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
mode, drop_ok = os.environ["SMX_MODE"], True
d, B, T = 64, 3, 96
torch.manual_seed(5)
if os.environ.get("SMX_ENC") == "branchformer":
    from summarymixing_amd.lobes.models.transformer.Branchformer import BranchformerEncoder
    enc = BranchformerEncoder(2, d, 1, kernel_size=31, activation=torch.nn.GELU, dropout=0.0, attention_type="SummaryMixing",
                              csgu_linear_units=128, local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d],
                              summary_out_dim=d, mode=mode)
    enc.eval()     # (the cell keeps its default global_dropout = 0.1 in train(), as the reference; the sharded mode is dropout-free)
else:
    enc = ConformerEncoder(2, d, 128, 4, kernel_size=31, activation="swish", dropout=0.0, attention_type="SummaryMixing",
                           local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], mode=mode)
with torch.no_grad():
    for n, p in enc.named_parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_normal_(p)
        elif "bias" in n:
            p.normal_(0, 0.05)
enc = enc.cuda()
x = torch.randn(B, T, d).cuda()
r = torch.randn(B, T, d).cuda()
lens = torch.tensor([T, 70, 40])                    # utterance 2 ends inside shard 0: shard 1 holds only its padding
pad = (torch.arange(T)[None] < lens[:, None]).cuda()
if os.environ.get("SMX_NOMASK") == "1":
    pad = None

kw_full, kw_loc = {}, {}
if os.environ.get("SMX_DYNCHUNK"):
    from summarymixing_amd import functional as F
    from summarymixing_amd.utils.dynamic_chunk_training import DynChunkTrainConfig
    cs, lc = os.environ["SMX_DYNCHUNK"].split(",")
    cs, lc = int(cs), (None if lc == "all" else int(lc))
    cfg = DynChunkTrainConfig(cs, lc)
    kw_full = dict(src_mask=F.DynChunkMask(T, cs, lc), dynchunktrain_config=cfg)
    kw_loc = dict(src_mask=F.DynChunkMask(T // world, cs, lc), dynchunktrain_config=cfg)

xf = x.clone().requires_grad_(True)
y, _ = enc(xf, src_key_padding_mask=pad, **kw_full)
(y * r).sum().backward()
gfull = {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}   # (expdecay: frozen decay constant)
dxf = xf.grad.clone()
for p in enc.parameters():
    p.grad = None

with SP.sequence_parallel():
    xl = SP.shard(x).requires_grad_(True)
    pl = SP.shard(pad) if pad is not None else None
    yl, _ = enc(xl, src_key_padding_mask=pl, **kw_loc)
    (yl * SP.shard(r)).sum().backward()
    SP.reduce_gradients(list(enc.parameters()))
    y_ref, dx_ref = SP.shard(y.detach()), SP.shard(dxf)

def close(a, b, what, tol=2e-4):
    err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-6)
    assert err < tol, f"rank {rank} {what}: rel err {err:.3e}"

close(yl.detach(), y_ref, "output")
close(xl.grad, dx_ref, "input gradient")
for n, g in gfull.items():
    close(dict(enc.named_parameters())[n].grad, g, "grad " + n, 5e-4)
dist.barrier()
print(f"rank {rank} OK")
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(mode, nomask=False, enc="conformer", dynchunk=""):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SMX_ROOT=ROOT, SMX_MODE=mode, SMX_NOMASK="1" if nomask else "0", SMX_ENC=enc, SMX_DYNCHUNK=dynchunk, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n[timeout]"
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {rank} OK" in out, f"rank {rank} failed:\n{out[-3000:]}"


@pytest.mark.parametrize("mode", ["SummaryMixing-fast", "SummaryMixing", "SummaryMixing-lite"])
def test_sequence_parallel_matches_unsharded(mode):
    _run(mode)


def test_sequence_parallel_no_padding_mask():
    _run("SummaryMixing-fast", nomask=True)


@pytest.mark.parametrize("nomask", [False, True])
def test_sequence_parallel_branchformer(nomask):
    """The Branchformer over two time shards: the CSGU's reflect-padded depthwise convolution gets its neighbours' frames as halos
    and the reflected frames at the two ends of the whole sequence (Branchformer.py:31-97); outputs, input gradients and summed
    parameter gradients equal the unsharded encoder's."""
    _run("SummaryMixing", nomask=nomask, enc="branchformer")


@pytest.mark.parametrize("nomask", [False, True])
def test_sequence_parallel_expdecay(nomask):
    """SummaryMixing-expdecay over two time shards (summary_mixing.py:316-365 without sum_mask): the two-sided exponential
    filter crosses the shard boundary through one (B, D) state per direction, the denominators use the global frame index
    (functional._expdecay_seqpar); forward and both transposed uses in the backward equal the unsharded encoder's."""
    _run("SummaryMixing-expdecay", nomask=nomask)


@pytest.mark.parametrize("dynchunk", ["8,2", "16,all", "12,1", "8,0", "24,2"])
def test_sequence_parallel_dynchunk(dynchunk):
    """Dynamic Chunk Training over two time shards of 48 frames (Conformer.py:190-313 chunked convolution: left halo of whole
    chunks, no right halo; summary_mixing.py:224-235 with the mask of TransformerASR.py:85-110: chunk sums of the previous /
    next shard, or the totals of all earlier / later ones with unlimited left context) equals the unsharded encoder."""
    _run("SummaryMixing-fast", dynchunk=dynchunk)


def test_sequence_parallel_dynchunk_full_mode_no_mask():
    _run("SummaryMixing", nomask=True, dynchunk="16,1")


DROP_WORKER = r'''
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["SMX_ROOT"])
from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoder
from summarymixing_amd import sequence_parallel as SP, ops

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
mode = os.environ["SMX_MODE"]
d, B, T = 64, 3, 96
torch.manual_seed(5)
if os.environ.get("SMX_ENC") == "branchformer":
    from summarymixing_amd.lobes.models.transformer.Branchformer import BranchformerEncoder
    enc = BranchformerEncoder(2, d, 1, kernel_size=31, activation=torch.nn.GELU, dropout=0.2, attention_type="SummaryMixing",
                              csgu_linear_units=128, local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d],
                              summary_out_dim=d, mode=mode)
else:
    enc = ConformerEncoder(2, d, 128, 4, kernel_size=31, activation="swish", dropout=0.2, attention_type="SummaryMixing",
                           local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], mode=mode)
with torch.no_grad():
    for n, p in enc.named_parameters():
        if p.dim() > 1:
            torch.nn.init.xavier_normal_(p)
enc = enc.cuda().train()
x = torch.randn(B, T, d).cuda()
r = torch.randn(B, T, d).cuda()
v = torch.randn(B, T, d).cuda()
lens = torch.tensor([T, 70, 40])
pad = (torch.arange(T)[None] < lens[:, None]).cuda()
kw = {}
if os.environ.get("SMX_DYNCHUNK"):
    from summarymixing_amd import functional as F
    from summarymixing_amd.utils.dynamic_chunk_training import DynChunkTrainConfig
    cs, lc = os.environ["SMX_DYNCHUNK"].split(",")
    cs, lc = int(cs), (None if lc == "all" else int(lc))
    kw = dict(src_mask=F.DynChunkMask(T // world, cs, lc), dynchunktrain_config=DynChunkTrainConfig(cs, lc))

def loss_of(xl, pl, rl, want_grad):
    """Local loss of this shard with the SAME dropout masks at every call (the seed counter is rewound)."""
    ops._drop_state["counter"] = 1000
    xin = xl.clone().requires_grad_(want_grad)
    yl, _ = enc(xin, src_key_padding_mask=pl, **kw)
    loss = (yl * rl).sum()
    if want_grad:
        loss.backward()
    return loss.detach(), yl.detach(), (xin.grad if want_grad else None)

with SP.sequence_parallel():
    xl, pl, rl, vl = SP.shard(x), SP.shard(pad), SP.shard(r), SP.shard(v)
    l0, y0, dx = loss_of(xl, pl, rl, True)
    l1, y1, _ = loss_of(xl, pl, rl, False)
    assert torch.equal(y0, y1), "same seeds, different outputs"
    assert torch.isfinite(y0).all() and torch.isfinite(dx).all()
    # the masks are live (training mode) ...
    enc.eval(); _, ye, _ = loss_of(xl, pl, rl, False); enc.train()
    assert (ye - y0).abs().max() > 1e-3
    # ... and the backward uses the forward's masks: directional derivative of the GLOBAL loss along v (every shard its slice)
    eps = 2e-3
    lp, _, _ = loss_of(xl + eps * vl, pl, rl, False)
    lm, _, _ = loss_of(xl - eps * vl, pl, rl, False)
    num = (lp - lm).double() / (2 * eps)
    ana = (dx * vl).sum().double()
    both = torch.stack([num, ana]).cpu()
    dist.all_reduce(both)
    rel = abs(float(both[0] - both[1])) / max(abs(float(both[1])), 1e-6)
    assert rel < 3e-2, f"directional derivative {float(both[0]):.6f} vs <dx, v> {float(both[1]):.6f}"
    # the shards draw DIFFERENT masks: the seed a site would draw next differs by rank
    seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(seeds, torch.tensor([ops.new_dropout_seed() & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64))
    assert len({int(t) for t in seeds}) == world
dist.barrier()
print(f"rank {rank} OK")
'''


def _run_drop(mode, enc="conformer", dynchunk=""):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SMX_ROOT=ROOT, SMX_MODE=mode, SMX_ENC=enc, SMX_DYNCHUNK=dynchunk, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", DROP_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n[timeout]"
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {rank} OK" in out, f"rank {rank} failed:\n{out[-3000:]}"


@pytest.mark.parametrize("mode,enc,dynchunk", [("SummaryMixing-fast", "conformer", ""), ("SummaryMixing", "branchformer", ""),
                                               ("SummaryMixing-fast", "conformer", "8,2"), ("SummaryMixing-expdecay", "conformer", "")])
def test_sequence_parallel_trains_with_dropout(mode, enc, dynchunk):
    """Round 6: the sharded mode with dropout 0.2 in train() (it refused before: every shard would have drawn the same mask from a
    seed indexed by LOCAL rows; now each rank salts the seeds it draws).  Same seeds -> bit-identical outputs; train != eval; the
    backward reuses the forward's masks (directional derivative of the global loss against <dx, v>); the ranks' seeds differ."""
    _run_drop(mode, enc, dynchunk)
