#!/usr/bin/env python3
"""Generate the golden fixtures tests/golden/*.npz by running the UNMODIFIED reference.

Runs ONLY in the dev container (needs /root/reference).  The reference files are imported in place
under the SpeechBrain stand-in in tests/golden/sb_standin (own code: upstream containers/Linear/
LayerNorm/FFN/CSGU/length_to_mask).  Outputs are data only: inputs, the module's
state_dict, outputs and gradients.  No reference source or bytecode is written anywhere.

    python tests/golden/make_golden.py            # rewrites every fixture (deterministic seeds)

Fixture groups: G1 cell modes x nhead x mask, G2 DynChunk sum_mask, G3 ParallelLinear,
G4 gradients, G5 encoder layers + BASELINE config-1 model (stand-in dependent), G6 quirks.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "sb_standin"))

import speechbrain  # noqa: E402,F401  (the stand-in)
from speechbrain.lobes.models.VanillaNN import ParallelLinear  # noqa: E402
from speechbrain.lobes.models.transformer.Branchformer import BranchformerEncoderLayer  # noqa: E402
from speechbrain.lobes.models.transformer.Conformer import ConformerEncoderLayer  # noqa: E402
from speechbrain.lobes.models.transformer.TransformerASR import (  # noqa: E402
    EncoderWrapper, TransformerASR, make_transformer_src_mask)
from speechbrain.nnet.activations import Swish  # noqa: E402
from speechbrain.nnet.summary_mixing import SummaryMixing  # noqa: E402
from speechbrain.utils.dynamic_chunk_training import DynChunkTrainConfig  # noqa: E402

ACTS = {"gelu": torch.nn.GELU, "swish": Swish, "leaky_relu": torch.nn.LeakyReLU}


def randomize(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("decay_constant"):
                continue
            if p.dim() == 1 and ("norm" in name and name.endswith("weight")):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                fan = p.shape[-1] if p.dim() > 1 else 16
                if name.endswith(".weights"):
                    fan = p.shape[1]
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / np.sqrt(fan)) * 1.5)


def save(name, meta, arrays, sd):
    out = {"meta": np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)}
    for k, v in arrays.items():
        if v is None:
            continue
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    for k, v in sd.items():
        out["sd/" + k] = v.detach().cpu().numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: tuple(v.shape) for k, v in out.items() if not k.startswith("sd/") and k != "meta"})


def ragged_mask(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(max(1, T // 3), T + 1, (B,), generator=g)
    lens[0] = T  # one full-length row
    return torch.arange(T)[None, :] < lens[:, None]


def cell_case(name, mode, nhead, B, T, d, hid, out, shid, sout, act, masked, seed, sum_mask=None, grads=True,
              extra_meta=None):
    torch.manual_seed(seed)
    m = SummaryMixing(enc_dim=d, nhead=nhead, local_proj_hid_dim=hid, local_proj_out_dim=out,
                      summary_hid_dim=shid, summary_out_dim=sout, activation=ACTS[act], global_dropout=0.1,
                      mode=mode).eval()
    randomize(m, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.randn(B, T, d, generator=g, requires_grad=True)
    pad = ragged_mask(B, T, seed + 3) if masked else None
    y = m(x, sum_mask=sum_mask, src_padding_mask=pad)
    arrays = {"x": x, "y": y, "pad_mask": pad, "sum_mask": sum_mask}
    meta = dict(kind="cell", mode=mode, nhead=nhead, act=act, local_proj_out_dim=out, masked=masked,
                y_stride=list(y.stride()))
    meta.update(extra_meta or {})
    if grads and mode != "SummaryMixing-lite" or (grads and mode == "SummaryMixing-lite"):
        r = torch.randn(y.shape, generator=g)
        (y * r).sum().backward()
        arrays["r"] = r
        arrays["gx"] = x.grad
        for k, p in m.named_parameters():
            if p.grad is not None:
                arrays["g/" + k] = p.grad
    save(name, meta, arrays, m.state_dict())


def main():
    # ---- G1 + G4: every mode x nhead x mask at (3,17,16) --------------------------------------
    i = 0
    for mode in ["SummaryMixing", "SummaryMixing-fast", "SummaryMixing-lite", "SummaryMixing-expdecay"]:
        for nhead in ([1] if mode == "SummaryMixing-fast" else [1, 4]):
            for masked in [False, True]:
                act = ["gelu", "swish"][i % 2]
                i += 1
                tag = mode.replace("SummaryMixing", "sm").replace("-", "_")
                cell_case(f"g1_{tag}_h{nhead}_{'mask' if masked else 'nomask'}", mode, nhead, 3, 17, 16,
                          [16], 16, [16], 16, act, masked, 100 + i)
    # the reference shape-test configuration (test_summary_mixing.py:5-57): (8,10,64), hid [32], out 32
    cell_case("g1_shape_full_h4", "SummaryMixing", 4, 8, 10, 64, [32], 32, [512], 64, "gelu", False, 666)
    cell_case("g1_shape_lite_h1", "SummaryMixing-lite", 1, 8, 10, 64, [32], 32, [512], 64, "gelu", False, 667)
    # two hidden layers + leaky relu (VanillaNN default act) + local_out != summary_out
    cell_case("g1_full_h2_deep", "SummaryMixing", 2, 2, 9, 12, [8, 20], 6, [24, 8], 10, "leaky_relu", True, 31)

    # ---- G2: DynChunk sum_mask built by the reference's own mask builder -----------------------
    for mode in ["SummaryMixing", "SummaryMixing-fast"]:
        for left in [None, 1]:
            T = 18
            sm = make_transformer_src_mask(torch.zeros(1, T, 1), causal=False, masked_false_or_true=False,
                                           dynchunktrain_config=DynChunkTrainConfig(4, left))
            tag = mode.replace("SummaryMixing", "sm").replace("-", "_")
            cell_case(f"g2_{tag}_chunk4_left{left}", mode, 1, 3, T, 16, [16], 16, [16], 16, "swish", True,
                      200 + (left or 0) + (7 if "fast" in mode else 0), sum_mask=sm,
                      extra_meta=dict(chunk_size=4, left_context=left))

    # ---- G3: ParallelLinear 3-D and 4-D inputs --------------------------------------------------
    torch.manual_seed(5)
    pl = ParallelLinear(n_neurons=24, input_size=16, n_split=4, combine_out_dims=True)
    pl4 = ParallelLinear(n_neurons=24, input_size=16, n_split=4, combine_out_dims=False)
    x3 = torch.randn(2, 7, 16)
    x4 = torch.randn(2, 7, 4, 4)
    save("g3_parallel_linear", dict(kind="parallel_linear"),
         {"x3": x3, "y3": pl(x3), "x4": x4, "y4_nocombine": pl4(x4), "y3_nocombine": pl4(x3)},
         {**{"a." + k: v for k, v in pl.state_dict().items()}, **{"b." + k: v for k, v in pl4.state_dict().items()}})

    # ---- G5: encoder layers (STAND-IN DEPENDENT: LayerNorm/FFN/CSGU come from the stand-in) -----
    for act, mode, seed in [("swish", "SummaryMixing-fast", 41), ("gelu", "SummaryMixing", 42)]:
        torch.manual_seed(seed)
        d = 32
        layer = ConformerEncoderLayer(d_model=d, d_ffn=64, nhead=4, kernel_size=7, activation=ACTS[act],
                                      dropout=0.0, attention_type="SummaryMixing", local_proj_hid_dim=[d],
                                      local_proj_out_dim=d, summary_hid_dim=[d], mode=mode).eval()
        randomize(layer, seed)
        x = torch.randn(3, 21, d, requires_grad=True)
        pad = ragged_mask(3, 21, seed)
        y, _ = layer(x, src_mask=None, src_key_padding_mask=pad)
        r = torch.randn(y.shape)
        (y * r).sum().backward()
        arrays = {"x": x, "y": y, "pad_mask": pad, "r": r, "gx": x.grad}
        for k, p in layer.named_parameters():
            arrays["g/" + k] = p.grad
        save(f"g5_conformer_layer_{act}", dict(kind="conformer_layer", act=act, mode=mode, nhead=4,
                                               local_proj_out_dim=d, standin_dependent=True),
             arrays, layer.state_dict())

    torch.manual_seed(43)
    d = 32
    bl = BranchformerEncoderLayer(d_model=d, nhead=1, kernel_size=7, activation=torch.nn.GELU, dropout=0.0,
                                  attention_type="SummaryMixing", csgu_linear_units=96, local_proj_hid_dim=[d],
                                  local_proj_out_dim=d, summary_hid_dim=[d], summary_out_dim=d,
                                  mode="SummaryMixing").eval()
    randomize(bl, 43)
    x = torch.randn(3, 21, d, requires_grad=True)
    pad = ragged_mask(3, 21, 43)
    y, _ = bl(x, src_mask=None, src_key_padding_mask=pad)
    r = torch.randn(y.shape)
    (y * r).sum().backward()
    arrays = {"x": x, "y": y, "pad_mask": pad, "r": r, "gx": x.grad}
    for k, p in bl.named_parameters():
        arrays["g/" + k] = p.grad
    save("g5_branchformer_layer", dict(kind="branchformer_layer", act="gelu", mode="SummaryMixing", nhead=1,
                                       local_proj_out_dim=d, standin_dependent=True), arrays, bl.state_dict())

    # BASELINE config-1 plumbing model (reduced d_ffn to keep the fixture small): 2-layer Conformer-SM
    # d=144, nhead 4, fast, input (2,50,640) + wav_len [1.0,0.6] through EncoderWrapper; also DynChunk.
    for tag, dyn in [("", None), ("_dynchunk", DynChunkTrainConfig(8, 2))]:
        torch.manual_seed(3407)
        net = TransformerASR(tgt_vocab=10, input_size=640, d_model=144, nhead=4, num_encoder_layers=2,
                             num_decoder_layers=0, d_ffn=576, dropout=0.0, activation=torch.nn.GELU,
                             encoder_module="conformer", conformer_activation=Swish,
                             attention_type="SummaryMixing", mode="SummaryMixing-fast", local_proj_out_dim=144,
                             local_proj_hid_dim=[144], summary_hid_dim=[144], normalize_before=True,
                             causal=False)
        enc = EncoderWrapper(net).eval()
        src = torch.randn(2, 50, 20, 32)
        wav_len = torch.tensor([1.0, 0.6])
        kw = {"dynchunktrain_config": dyn} if dyn is not None else {}
        y = enc(src, wav_len, **kw)
        sd = {k: v for k, v in net.state_dict().items() if k != "positional_encoding.pe"}
        save("g5_config1_encoder" + tag,
             dict(kind="asr_encode", encoder_module="conformer", act="swish", mode="SummaryMixing-fast",
                  local_proj_out_dim=144, nhead=4, standin_dependent=True,
                  dynchunk=None if dyn is None else [dyn.chunk_size, dyn.left_context_size]),
             {"src": src, "wav_len": wav_len, "y": y}, sd if tag == "" else {})

    torch.manual_seed(77)
    net = TransformerASR(tgt_vocab=10, input_size=80, d_model=32, nhead=1, num_encoder_layers=2,
                         num_decoder_layers=0, d_ffn=64, dropout=0.0, activation=torch.nn.GELU,
                         encoder_module="branchformer", branchformer_activation=torch.nn.GELU,
                         attention_type="SummaryMixing", mode="SummaryMixing", local_proj_out_dim=32,
                         local_proj_hid_dim=[32], summary_hid_dim=[32], summary_out_dim=32, csgu_linear_units=96,
                         kernel_size=7,
                         normalize_before=True, causal=False)
    randomize(net, 78)
    enc = EncoderWrapper(net).eval()
    src = torch.randn(3, 25, 80)
    wav_len = torch.tensor([1.0, 0.5, 0.8])
    y = enc(src, wav_len)
    sd = {k: v for k, v in net.state_dict().items() if k != "positional_encoding.pe"}
    save("g5_branchformer_encoder", dict(kind="asr_encode", encoder_module="branchformer", act="gelu",
                                         mode="SummaryMixing", local_proj_out_dim=32, nhead=1,
                                         standin_dependent=True, dynchunk=None),
         {"src": src, "wav_len": wav_len, "y": y}, sd)

    # ---- G6: quirks ---------------------------------------------------------------------------
    # (a) all-padding row -> 0/0 = NaN in that row's summary (summary_mixing.py:264-266)
    torch.manual_seed(9)
    m = SummaryMixing(enc_dim=8, nhead=1, local_proj_hid_dim=[8], local_proj_out_dim=8, summary_hid_dim=[8],
                      summary_out_dim=8, mode="SummaryMixing-fast").eval()
    randomize(m, 9)
    x = torch.randn(2, 5, 8)
    pad = torch.tensor([[True] * 5, [False] * 5])
    y = m(x, src_padding_mask=pad)
    save("g6_allpad_row", dict(kind="cell", mode="SummaryMixing-fast", nhead=1, act="gelu", local_proj_out_dim=8,
                               masked=True, quirk="row 1 has zero valid frames -> non-finite output"),
         {"x": x, "y": y, "pad_mask": pad, "isnan": torch.isnan(y)}, m.state_dict())
    # (b) padded-frame CONTENT changes valid-frame outputs at layer level (conv sees padded frames)
    torch.manual_seed(10)
    layer = ConformerEncoderLayer(d_model=16, d_ffn=32, nhead=1, kernel_size=5, activation=Swish, dropout=0.0,
                                  attention_type="SummaryMixing", local_proj_hid_dim=[16], local_proj_out_dim=16,
                                  summary_hid_dim=[16], mode="SummaryMixing-fast").eval()
    randomize(layer, 10)
    xa = torch.randn(2, 12, 16)
    pad = torch.arange(12)[None, :] < torch.tensor([12, 7])[:, None]
    xb = xa.clone()
    xb[1, 7:] = torch.randn(5, 16) * 3
    ya, _ = layer(xa, src_key_padding_mask=pad)
    yb, _ = layer(xb, src_key_padding_mask=pad)
    save("g6_padded_content", dict(kind="conformer_layer", act="swish", mode="SummaryMixing-fast", nhead=1,
                                   local_proj_out_dim=16, standin_dependent=True,
                                   quirk="valid frames differ between ya and yb"),
         {"x": xa, "y": ya, "xb": xb, "yb": yb, "pad_mask": pad}, layer.state_dict())


if __name__ == "__main__":
    main()
