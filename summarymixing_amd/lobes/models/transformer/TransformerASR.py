"""The lobes-level encoder API on MI355X: TransformerASR.encode / EncoderWrapper, mask builders, abs-sine PE.

Mirrors speechbrain/lobes/models/transformer/TransformerASR.py (:50-180 masks, :281-360 ctor, :501-560 encode,
:687-741 EncoderWrapper) and Transformer.py:284-335 (PositionalEncoding), :201-259 (encoder dispatch) for
attention_type="SummaryMixing" with encoder_module in {"conformer", "branchformer"}.  The MHA decoder, the
vanilla-Transformer encoder (broken for SummaryMixing in the reference) and streaming are out of scope.
"""
import math
from typing import Optional

import torch
from torch import nn

from .... import functional as F
from .... import sequence_parallel as SP
from ....nnet.activations import Swish
from ....utils.dynamic_chunk_training import DynChunkTrainConfig  # noqa: F401
from ...models.VanillaNN import Linear
from .Branchformer import BranchformerEncoder
from .Conformer import ConformerEncoder


def length_to_mask(length, max_len=None):
    """arange(max_len)[None] < length[:, None]  (speechbrain.dataio.dataio.length_to_mask)."""
    if max_len is None:
        max_len = int(length.max().item())
    return torch.arange(max_len, device=length.device)[None, :] < length[:, None]


def make_transformer_src_mask(src, causal=False, masked_false_or_true=True, dynchunktrain_config=None):
    """TransformerASR.py:50-110.  Returns None or a functional.DynChunkMask (closed form of the (T,T) mask; call
    .dense() for the boolean matrix).  masked_false_or_true=True inverts the dense matrix like the reference."""
    if causal:
        raise NotImplementedError("causal masks make SummaryMixing non-finite in the reference (SURVEY §8a A5)")
    if dynchunktrain_config is None:
        return None
    m = F.DynChunkMask(src.size(1), dynchunktrain_config.chunk_size, dynchunktrain_config.left_context_size)
    if masked_false_or_true:
        return ~m.dense(src.device)
    return m


def make_transformer_src_tgt_masks(src, tgt=None, wav_len=None, pad_idx=0, causal=False, masked_false_or_true=True,
                                   dynchunktrain_config=None):
    """TransformerASR.py:113-180 (encoder side)."""
    if tgt is not None:
        raise NotImplementedError("the MHA decoder is outside the SummaryMixing hot path")
    src_key_padding_mask = None
    if wav_len is not None:
        abs_len = torch.round(wav_len * src.shape[1])
        # the reference sizes the mask by abs_len.max() (a host sync) and only works when that equals T (the longest
        # utterance of a batch has wav_len == 1); max_len = T gives the same mask without the sync (hipGraph-capturable)
        valid = length_to_mask(abs_len, src.shape[1])
        src_key_padding_mask = ~valid if masked_false_or_true else valid
    src_mask = make_transformer_src_mask(src, causal, masked_false_or_true, dynchunktrain_config)
    return src_key_padding_mask, None, src_mask, None


class PositionalEncoding(nn.Module):
    """Absolute sinusoidal table (Transformer.py:306-335); ``forward`` returns pe[:, :T]."""

    def __init__(self, input_size, max_len=2500):
        super().__init__()
        if input_size % 2 != 0:
            raise ValueError(f"Cannot use sin/cos positional encoding with odd channels (got channels={input_size})")
        self.max_len = max_len
        pe = torch.zeros(max_len, input_size)
        pos = torch.arange(0, max_len).unsqueeze(1).float()
        den = torch.exp(torch.arange(0, input_size, 2).float() * -(math.log(10000.0) / input_size))
        pe[:, 0::2] = torch.sin(pos * den)
        pe[:, 1::2] = torch.cos(pos * den)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        return self.pe[:, : x.size(1)].clone().detach()


class _SrcModule(nn.Module):
    """Key layout of the reference's custom_src_module: layers.0 = Linear holder (.w), layers.1 = Dropout."""

    def __init__(self, input_size, d_model, dropout):
        super().__init__()
        self.layers = nn.ModuleList([Linear(d_model, input_size), nn.Dropout(dropout)])


class TransformerASR(nn.Module):
    def __init__(self, tgt_vocab, input_size, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6,
                 d_ffn=2048, dropout=0.1, activation=nn.ReLU, positional_encoding="fixed_abs_sine",
                 normalize_before=False, kernel_size: Optional[int] = 31, bias: Optional[bool] = True,
                 encoder_module: Optional[str] = "transformer", conformer_activation=Swish,
                 branchformer_activation=nn.GELU, attention_type: Optional[str] = "SummaryMixing",
                 max_length: Optional[int] = 2500, causal: Optional[bool] = True,
                 csgu_linear_units: Optional[int] = 3072, gate_activation=nn.Identity,
                 use_linear_after_conv: Optional[bool] = False, local_proj_hid_dim: Optional[list] = [512],
                 local_proj_out_dim: Optional[int] = 512, summary_hid_dim: Optional[list] = [1024],
                 summary_out_dim: Optional[int] = 1024, mode: Optional[str] = "SummaryMixing",
                 masked_false_or_true: Optional[bool] = True):
        super().__init__()
        if attention_type != "SummaryMixing":
            raise NotImplementedError("summarymixing_amd implements attention_type='SummaryMixing' only")
        if causal:
            raise NotImplementedError("causal=True is unsupported with SummaryMixing (non-finite in the reference)")
        self.causal, self.attention_type = causal, attention_type
        self.positional_encoding_type = positional_encoding
        self.num_encoder_layers, self.num_decoder_layers = num_encoder_layers, num_decoder_layers
        self.masked_false_or_true = False                                    # TransformerASR.py:344-347
        self.p_drop = float(dropout)
        if positional_encoding == "fixed_abs_sine":
            self.positional_encoding = PositionalEncoding(d_model, max_length)
        elif positional_encoding is not None:
            raise NotImplementedError(f"positional_encoding={positional_encoding}")
        if encoder_module == "conformer":
            self.encoder = ConformerEncoder(nhead=nhead, num_layers=num_encoder_layers, d_ffn=d_ffn, d_model=d_model,
                                            dropout=dropout, activation=conformer_activation, kernel_size=kernel_size,
                                            bias=bias, causal=causal, attention_type=attention_type,
                                            local_proj_hid_dim=local_proj_hid_dim,
                                            local_proj_out_dim=local_proj_out_dim, summary_hid_dim=summary_hid_dim,
                                            mode=mode)
        elif encoder_module == "branchformer":
            self.encoder = BranchformerEncoder(nhead=nhead, num_layers=num_encoder_layers, d_model=d_model,
                                               dropout=dropout, activation=branchformer_activation,
                                               kernel_size=kernel_size, attention_type=attention_type,
                                               csgu_linear_units=csgu_linear_units, gate_activation=gate_activation,
                                               use_linear_after_conv=use_linear_after_conv,
                                               local_proj_hid_dim=local_proj_hid_dim,
                                               local_proj_out_dim=local_proj_out_dim, summary_hid_dim=summary_hid_dim,
                                               summary_out_dim=summary_out_dim, mode=mode)
        else:
            raise NotImplementedError("encoder_module='transformer' + SummaryMixing is broken in the reference "
                                      "(SURVEY §2 row 8); use 'conformer' or 'branchformer'")
        self.custom_src_module = _SrcModule(input_size, d_model, dropout)
        for p in self.parameters():                                           # _init_params, :681-684
            if p.dim() > 1:
                torch.nn.init.xavier_normal_(p)

    def forward(self, src, tgt=None, wav_len=None, pad_idx=0):
        raise NotImplementedError("the seq2seq decoder is outside the SummaryMixing hot path; call .encode()")

    def encode(self, src, wav_len=None, pad_idx=0, dynchunktrain_config=None, masked_false_or_true: Optional[bool] = True):
        """TransformerASR.py:501-560.  masked_false_or_true is honoured exactly like the reference: the SummaryMixing cell
        and conv module read the padding mask as True = VALID (masked_false_or_true=False, what EncoderWrapper and the
        reference's own forward() pass, :344-347,:720-729); a direct call with the signature's default True hands them the
        inverted mask - in the reference too."""
        if SP.enabled():
            raise NotImplementedError("sequence-parallel mode enters at the ConformerEncoder stack: the positional table, "
                                      "the wav_len masks and the input dropout counters here are not offset per shard")
        if src.dim() == 4:
            bz, t, ch1, ch2 = src.shape
            src = src.reshape(bz, t, ch1 * ch2)
        B, T, _ = src.shape
        key_padding_mask, _, src_mask, _ = make_transformer_src_tgt_masks(
            src, None, wav_len, pad_idx=pad_idx, causal=self.causal, dynchunktrain_config=dynchunktrain_config,
            masked_false_or_true=masked_false_or_true)
        lin = self.custom_src_module.layers[0].w
        if self.positional_encoding_type == "fixed_abs_sine":
            if T > self.positional_encoding.max_len:
                raise ValueError(f"sequence length {T} exceeds max_length {self.positional_encoding.max_len}")
            pe = self.positional_encoding.pe[0, :T]
        else:
            pe = torch.zeros((T, lin.weight.shape[0]), device=src.device)
        x = F.input_proj_pe(src, lin.weight, lin.bias, pe, T, self.p_drop if self.training else 0.0)
        # (x is the residual stream: float32 for a bf16 model by default - the encoder is told the GEMM dtype)
        kw = {"_compute_dtype": src.dtype}
        out, _ = self.encoder(src=x, src_mask=src_mask, src_key_padding_mask=key_padding_mask, pos_embs=None,
                              dynchunktrain_config=dynchunktrain_config, **kw)
        return out


class EncoderWrapper(nn.Module):
    """forward() = transformer.encode() (TransformerASR.py:715-729)."""

    def __init__(self, transformer, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.transformer = transformer

    def forward(self, x, wav_lens=None, pad_idx=0, **kwargs):
        return self.transformer.encode(x, wav_lens, pad_idx, **kwargs,
                                       masked_false_or_true=self.transformer.masked_false_or_true)
