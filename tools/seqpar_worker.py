#!/usr/bin/env python3
"""One rank of a 2-rank sequence-parallel forward + backward (gloo, both ranks on cuda:0: the GPU box has one device) for
tools/seqpar_profile.sh.  env: RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT, SMX_MODE (cell mode), SMX_DYNCHUNK ("8,2" | "")."""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoder
from summarymixing_amd import sequence_parallel as SP, functional as F
from summarymixing_amd.utils.dynamic_chunk_training import DynChunkTrainConfig

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
mode = os.environ.get("SMX_MODE", "SummaryMixing-fast")
d, B, T = 256, 4, 8192                                     # 32768 frames, 16384 per rank: activation = 4.2 M elements per tensor
torch.manual_seed(5)
enc = ConformerEncoder(2, d, 1024, 4, kernel_size=31, activation="swish", dropout=0.1, attention_type="SummaryMixing",
                       local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], mode=mode).cuda().train()
x = torch.randn(B, T, d, device="cuda").bfloat16()
r = torch.randn(B, T, d, device="cuda").bfloat16()
pad = (torch.arange(T, device="cuda")[None] < torch.tensor([T, 7000, 5000, 8000], device="cuda")[:, None])
kw = {}
if os.environ.get("SMX_DYNCHUNK"):
    cs, lc = os.environ["SMX_DYNCHUNK"].split(",")
    cs, lc = int(cs), (None if lc == "all" else int(lc))
    kw = dict(src_mask=F.DynChunkMask(T // world, cs, lc), dynchunktrain_config=DynChunkTrainConfig(cs, lc))
with SP.sequence_parallel():
    xl, pl, rl = SP.shard(x).requires_grad_(True), SP.shard(pad), SP.shard(r)
    for _ in range(3):
        yl, _ = enc(xl, src_key_padding_mask=pl, **kw)
        yl.backward(rl)                                    # (no host-side elementwise pass of its own)
        SP.reduce_gradients(list(enc.parameters()))
torch.cuda.synchronize()
dist.barrier()
print(f"rank {rank} done: {B} x {T // world} frames per rank, d = {d}, mode {mode}, dynchunk {os.environ.get('SMX_DYNCHUNK') or '-'}")
