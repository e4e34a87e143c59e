"""hipGraph replay of the whole training step (bench.py --graph): the device step counter keeps AdamW's bias correction
and the dropout epoch advancing although every captured kernel argument is constant."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("ln_fuse_mode")]   # (both LayerNorm dispatches: tests/conftest.py)


def _setup(dropout):
    from summarymixing_amd.lobes.models.transformer.TransformerASR import EncoderWrapper, TransformerASR
    from summarymixing_amd.trainer import FlatAdamW
    torch.manual_seed(11)
    net = TransformerASR(tgt_vocab=50, input_size=64, d_model=64, nhead=4, num_encoder_layers=2, num_decoder_layers=0,
                         d_ffn=128, dropout=dropout, activation=torch.nn.GELU, encoder_module="conformer",
                         attention_type="SummaryMixing", mode="SummaryMixing-fast", local_proj_hid_dim=[64],
                         local_proj_out_dim=64, summary_hid_dim=[64], summary_out_dim=64, causal=False)
    enc = EncoderWrapper(net).cuda().train()
    opt = FlatAdamW(enc, lr=1e-3, compute_dtype=torch.float32)
    g = torch.Generator().manual_seed(5)
    src = torch.randn(4, 48, 64, generator=g).cuda()
    wav_len = torch.tensor([1.0, 0.6, 0.8, 0.5]).cuda()
    r = (torch.randn(4, 48, 64, generator=g) / 100).cuda()

    def step():
        opt.zero_grad()
        enc(src, wav_len).backward(r)
        opt.step()
    return enc, opt, step


def _capture(opt, step):
    opt.use_device_step_counter(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    return graph


def test_graph_replay_matches_eager_steps():
    enc_a, opt_a, step_a = _setup(0.0)
    for _ in range(5):
        step_a()
    ref = opt_a.flat_p.clone()
    enc_b, opt_b, step_b = _setup(0.0)
    try:
        step_b()                       # eager step 1
        graph = _capture(opt_b, step_b)  # eager step 2 on the side stream, then capture (not executed)
        for _ in range(3):             # steps 3..5
            graph.replay()
        torch.cuda.synchronize()
        assert int(opt_b._dev_step.item()) == 5
        err = (opt_b.flat_p - ref).abs().max().item() / ref.abs().max().item()
        assert err < 1e-5, err
    finally:
        opt_b.use_device_step_counter(False)


def test_graph_replay_draws_fresh_dropout_masks():
    from summarymixing_amd import ops
    x = torch.ones(256, 64, device="cuda")
    y = torch.empty_like(x)
    counter = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.set_step_counter(counter)
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.dropout(x, 0.5, 77, out=y)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            ops.dropout(x, 0.5, 77, out=y)
            ops.step_counter_add(counter, 1)
        graph.replay(); torch.cuda.synchronize(); m1 = y.clone()
        graph.replay(); torch.cuda.synchronize(); m2 = y.clone()
        assert not torch.equal(m1, m2)                       # same captured seed, different epoch
        assert abs((m2 != 0).float().mean().item() - 0.5) < 0.02
    finally:
        ops.set_step_counter(None)
    assert torch.equal(ops.dropout(x, 0.5, 77), ops.dropout(x, 0.5, 77))   # counter cleared: by-value seeds again

