"""State that must survive checkpoint / resume, checked on the CPU (parameter holders only, no kernels):
SpeechBrain-compatible keys of the CNN front-end and the InputNormalization statistics."""
import torch


def test_convolution_frontend_keys_match_speechbrain_layout():
    from summarymixing_amd.lobes.models.convolution import ConvolutionFrontEnd
    fe = ConvolutionFrontEnd((None, None, 80), num_blocks=2, out_channels=(64, 32))
    keys = sorted(fe.state_dict().keys())
    want = sorted(f"convblock_{i}.convs.{m}.{p}" for i in (0, 1) for m, p in
                  (("conv_0", "conv.weight"), ("conv_0", "conv.bias"), ("norm_0", "norm.weight"), ("norm_0", "norm.bias")))
    assert keys == want
    # a checkpoint written with upstream's key layout (the `CNN` recoverable of the recipes) loads strictly
    ckpt = {"convblock_0.convs.conv_0.conv.weight": torch.randn(64, 1, 3, 3), "convblock_0.convs.conv_0.conv.bias": torch.randn(64),
            "convblock_0.convs.norm_0.norm.weight": torch.randn(40, 64), "convblock_0.convs.norm_0.norm.bias": torch.randn(40, 64),
            "convblock_1.convs.conv_0.conv.weight": torch.randn(32, 64, 3, 3), "convblock_1.convs.conv_0.conv.bias": torch.randn(32),
            "convblock_1.convs.norm_0.norm.weight": torch.randn(20, 32), "convblock_1.convs.norm_0.norm.bias": torch.randn(20, 32)}
    fe.load_state_dict(ckpt, strict=True)
    assert torch.equal(fe.blocks[1].conv.weight, ckpt["convblock_1.convs.conv_0.conv.weight"])
    assert set(fe.state_dict_for_oracle()) == {f"convblock_{i}.{m}.{p}" for i in (0, 1) for m in ("conv", "norm") for p in ("weight", "bias")}


def test_input_normalization_statistics_persist(tmp_path):
    from summarymixing_amd.lobes.features import InputNormalization
    n = InputNormalization(norm_type="global")
    assert "glob_mean" not in n.state_dict()                       # nothing learned yet
    n.glob_mean, n.glob_std, n.count = torch.arange(5.0), torch.arange(5.0) + 1, 17
    sd = n.state_dict()
    assert torch.equal(sd["glob_mean"], torch.arange(5.0)) and sd["_extra_state"]["count"] == 17
    m = InputNormalization(norm_type="global")
    m.load_state_dict(sd, strict=True)                             # fresh module: buffers are created from the checkpoint
    assert m.count == 17 and torch.equal(m.glob_std, torch.arange(5.0) + 1)
    # SpeechBrain checkpointer hooks (normalizer.ckpt format)
    path = str(tmp_path / "normalizer.ckpt")
    n._save(path)
    k = InputNormalization(norm_type="global")
    k._load(path)
    assert k.count == 17 and torch.equal(k.glob_mean, torch.arange(5.0))
    stats = torch.load(path)
    assert set(stats) >= {"count", "glob_mean", "glob_std", "spk_dict_mean", "spk_dict_std", "spk_dict_count"}
