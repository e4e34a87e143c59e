#!/usr/bin/env python3
"""One eager training step of a bench.py config with the in-step records on (ops.prof_start): time per record label, and
for the labels matching argv[2] (a substring) the repo frames that issued the call.
usage: python tools/step_records.py c4 act_mask_bwd      (BT=10,375 in the environment: another batch size)"""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from summarymixing_amd import ops  # noqa: E402
from summarymixing_amd.trainer import FlatAdamW  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2b"
pat = sys.argv[2] if len(sys.argv) > 2 else None
cfg = dict(bench.CONFIGS[name])
if os.environ.get("BT"):                                   # BT=10,375: another batch of the same model
    cfg["B"], cfg["T"] = (int(v) for v in os.environ["BT"].split(","))
dev = torch.device("cuda", 0)
enc = bench.build_encoder(cfg, dev, 0.15)
opt = FlatAdamW(enc, compute_dtype=torch.bfloat16)
src, wav_len, r, _ = bench.synthetic_batch(cfg, 0, dev, torch.bfloat16)


enc_kw = {}
if os.environ.get("DYNCHUNK"):                             # DYNCHUNK=8,2: a DynChunk training batch (chunk size, left context chunks)
    from summarymixing_amd.utils.dynamic_chunk_training import DynChunkTrainConfig
    parts = [int(v) for v in os.environ["DYNCHUNK"].split(",")]
    enc_kw["dynchunktrain_config"] = DynChunkTrainConfig(parts[0], parts[1] if len(parts) > 1 else None)


def step():
    opt.zero_grad()
    enc(src, wav_len, **enc_kw).backward(r)
    opt.step()


for _ in range(2):
    step()
sites = collections.Counter()
if pat:
    orig = ops._pb

    def pb(label, nbytes, flops=0.0):
        if pat in label:
            fr = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()[:-1] if "/summarymixing_amd/" in f.filename and "ops.py" not in f.filename][-3:]
            sites[(label, " <- ".join(reversed(fr)))] += 1
        return orig(label, nbytes, flops)
    ops._pb = pb
ops.prof_start()
step()
recs = ops.prof_stop()
tot = collections.defaultdict(lambda: [0, 0.0, 0.0])
for n, b, f, ms in recs:
    t = tot[n]
    t[0] += 1; t[1] += ms; t[2] += b
allms = sum(t[1] for t in tot.values())
print(f"# {name}: {len(recs)} records, {allms:.2f} ms inside records (eager step, B={cfg['B']} x T={cfg['T']})")
for n, (c, ms, b) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{100*ms/allms:5.1f} %  {c:4d} x {1e3*ms/c:8.1f} us  {b/c/1e6:8.0f} MB  {b/ms/1e9 if ms else 0:5.2f} TB/s  {n}")
if pat:
    print(f"# call sites of '{pat}'")
    for (label, st), c in sites.most_common(30):
        print(f"{c:4d}  {label:32s} {st}")
