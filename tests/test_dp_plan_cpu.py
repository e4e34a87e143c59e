"""The N-rank data-parallel plan without GPUs (`bench.py --dry-run-ranks N`): the gradient buckets and the
reduce='rs_ag' shard map cover every element of the flat buffers exactly once, for the world sizes of BASELINE config 3."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny():
    from summarymixing_amd.lobes.models.transformer.TransformerASR import EncoderWrapper, TransformerASR
    torch.manual_seed(0)
    net = TransformerASR(tgt_vocab=10, input_size=40, d_model=64, nhead=4, num_encoder_layers=3, num_decoder_layers=0, d_ffn=128,
                         dropout=0.0, attention_type="SummaryMixing", local_proj_hid_dim=[64], local_proj_out_dim=64,
                         summary_hid_dim=[64], causal=False, kernel_size=31, encoder_module="conformer", mode="SummaryMixing-fast")
    return EncoderWrapper(net)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_buckets_and_shards_cover_every_element_exactly_once(world):
    from summarymixing_amd.trainer import FlatAdamW, plan_buckets, shard_map
    enc = _tiny()
    opt = FlatAdamW(enc, compute_dtype=torch.float32)
    ranges = [opt.param_range(list(l.parameters())) for l in enc.transformer.encoder.layers]
    buckets = plan_buckets(opt.total, ranges)
    # launch order: the layers as their backward passes finish (last first), then front and back
    assert [n for _, _, n in buckets[:3]] == ["layer 2", "layer 1", "layer 0"]
    hits = torch.zeros(opt.total, dtype=torch.int32)
    for a, b, _ in buckets:
        hits[a:b] += 1
    assert int(hits.min()) == 1 and int(hits.max()) == 1
    shards = shard_map(buckets, world)
    hits.zero_()
    for r in range(world):
        assert sum(b - a for a, b in shards[r]) == opt.total // world
        for a, b in shards[r]:
            hits[a:b] += 1
    assert int(hits.min()) == 1 and int(hits.max()) == 1
    # the shard map is the one FlatAdamW._shard uses for its update and its moments
    opt.world, opt.rank = world, world - 1
    assert [opt._shard(a, b) for a, b, _ in buckets] == shards[world - 1]


def test_shard_map_refuses_a_bucket_that_does_not_divide():
    from summarymixing_amd.trainer import shard_map
    with pytest.raises(ValueError):
        shard_map([(0, 100, "x")], 8)


@pytest.mark.timeout(300)
def test_bench_dry_run_ranks_8_prints_the_plan_without_a_gpu():
    env = dict(os.environ, HIP_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-ranks", "8", "--reduce", "rs_ag",
                          "--grad-dtype", "bf16"], capture_output=True, text=True, env=env, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["dry_run_ranks"] == 8 and d["buckets_cover_flat_buffer_exactly_once"]
    assert len(d["buckets_in_launch_order"]) in (13, 14) and d["buckets_in_launch_order"][0]["name"] == "layer 11"
    assert len(set(d["rs_ag_shard_elements_per_rank"])) == 1 and d["rs_ag_shard_elements_per_rank"][0] * 8 == d["flat_elements"]
    assert sum(b["wire_dtype_bytes"] for b in d["buckets_in_launch_order"]) == 2 * d["flat_elements"]      # bf16 on the wire
    assert d["hipgraph"] is False and "HSA_ENABLE_IPC_MODE_LEGACY" in d["rccl_env"]
