"""Conformer encoder with the SummaryMixing cell on MI355X.

Mirrors the constructor / forward signatures and state-dict keys of the reference
(speechbrain/lobes/models/transformer/Conformer.py: ConvolutionModule :72-331, ConformerEncoderLayer :334-537,
ConformerEncoder :623-786) for attention_type="SummaryMixing".  The torch.nn sub-modules below are parameter
HOLDERS only (same names => same keys => reference checkpoints load); all arithmetic runs in libsmx.so:

  layer(x):  x1 = x  + 1/2 FFN1(x)              LN -> GEMM(+bias,act) -> GEMM(+bias, residual, alpha=1/2)
             x2 = x1 + SummaryMixing(LN(x1))    cell with the skip fused as the merge-GEMM residual
             x3 = x2 + mask * ConvModule(x2)    LN -> GEMM -> fused GLU+dwconv -> LN+act -> GEMM(+mask,residual)
             y  = LN(x3 + 1/2 FFN2(x3))
Attention types other than SummaryMixing, causal convolution and the streaming context are out of scope
 and raise NotImplementedError.
"""
from typing import Optional

import torch
import torch.nn as nn

from .... import functional as F
from .... import ops
from ....nnet.activations import Swish, act_code
from ....nnet.summary_mixing import SummaryMixing


class _LayerNorm(nn.Module):
    """Key layout of speechbrain.nnet.normalization.LayerNorm (``.norm`` is an nn.LayerNorm)."""

    def __init__(self, d, eps=1e-5):
        super().__init__()
        self.norm = nn.LayerNorm(d, eps=eps)


class _FFN(nn.Module):
    """Key layout of PositionalwiseFeedForward: ``ffn.0`` Linear(d,f), ``ffn.3`` Linear(f,d)."""

    def __init__(self, d_ffn, input_size, dropout, activation):
        super().__init__()
        self.ffn = nn.Sequential(nn.Linear(input_size, d_ffn), nn.Identity(), nn.Dropout(dropout),
                                 nn.Linear(d_ffn, input_size))


def _ffn_params(seq):
    return {"ln_w": seq[0].weight, "ln_b": seq[0].bias, "W1": seq[1].ffn[0].weight, "b1": seq[1].ffn[0].bias,
            "W2": seq[1].ffn[3].weight, "b2": seq[1].ffn[3].bias}


class ConvolutionModule(nn.Module):
    def __init__(self, input_size, kernel_size=31, bias=True, activation=Swish, dropout=0.0, causal=False,
                 dilation=1, masked_false_or_true=True):
        super().__init__()
        if causal or dilation != 1:
            raise NotImplementedError("causal / dilated convolution is outside the SummaryMixing hot path")
        self.kernel_size, self.causal, self.dilation = kernel_size, causal, dilation
        self.masked_false_or_true = masked_false_or_true
        self.padding = (kernel_size - 1) // 2
        self.act = act_code(activation)
        self.p_drop = float(dropout)
        self.layer_norm = nn.LayerNorm(input_size)
        self.bottleneck = nn.Sequential(nn.Conv1d(input_size, 2 * input_size, kernel_size=1, bias=bias), nn.Identity())
        self.conv = nn.Conv1d(input_size, input_size, kernel_size, padding=self.padding, groups=input_size, bias=bias)
        self.after_conv = nn.Sequential(nn.LayerNorm(input_size), nn.Identity(), nn.Linear(input_size, input_size, bias=bias),
                                        nn.Dropout(dropout))

    def params(self):
        return {"ln1_w": self.layer_norm.weight, "ln1_b": self.layer_norm.bias, "Wp": self.bottleneck[0].weight,
                "bp": self.bottleneck[0].bias, "wd": self.conv.weight, "bd": self.conv.bias,
                "ln2_w": self.after_conv[0].weight, "ln2_b": self.after_conv[0].bias, "Wo": self.after_conv[2].weight,
                "bo": self.after_conv[2].bias}

    def forward(self, x, mask: Optional[torch.Tensor] = None, dynchunktrain_config=None):
        """Returns conv_module(x) (without the residual), mask (B,T,1) multiplies the output when
        masked_false_or_true is False (the SummaryMixing convention, Conformer.py:327-331)."""
        B, T, d = x.shape
        if mask is not None and self.masked_false_or_true:
            mask = ~mask.bool()
        m8 = F.mask_u8(mask.reshape(B, T) if mask is not None else None, B, T, x.device)
        P, act = self.params(), self.act
        chunk = dynchunktrain_config.chunk_size if dynchunktrain_config is not None else 0
        pd = self.p_drop if self.training else 0.0

        def run(xin, need_bwd):
            x2 = ops.rows2d(xin)
            y, bwd = F.conv_module_fwd(x2, P, act, m8, B, T, need_bwd, xin.dtype, chunk, residual=False, p=pd)
            return y.view(B, T, d), ((lambda dy: bwd(ops.rows2d(dy.contiguous())).view(B, T, d)) if need_bwd else None)
        return F.block(x, run, list(self.parameters()))


class ConformerEncoderLayer(nn.Module):
    def __init__(self, d_model, d_ffn, nhead, kernel_size=31, kdim=None, vdim=None, activation=Swish, bias=True,
                 dropout=0.0, causal=False, attention_type="RelPosMHAXL", local_proj_hid_dim=[512],
                 local_proj_out_dim=512, summary_hid_dim=[1024], mode="SummaryMixing"):
        super().__init__()
        if attention_type != "SummaryMixing":
            raise NotImplementedError("summarymixing_amd implements attention_type='SummaryMixing' only")
        self.attention_type, self.mode = attention_type, mode
        self.masked_false_or_true = False                     # Conformer.py:447
        self.act = act_code(activation)
        self.p_drop = float(dropout)
        self.mha_layer = SummaryMixing(enc_dim=d_model, nhead=nhead, local_proj_hid_dim=local_proj_hid_dim,
                                       local_proj_out_dim=local_proj_out_dim, summary_hid_dim=summary_hid_dim,
                                       summary_out_dim=d_model, activation=activation, global_dropout=dropout, mode=mode)
        self.convolution_module = ConvolutionModule(d_model, kernel_size, bias, activation, dropout, causal=causal,
                                                    masked_false_or_true=False)
        self.ffn_module1 = nn.Sequential(nn.LayerNorm(d_model), _FFN(d_ffn, d_model, dropout, activation), nn.Dropout(dropout))
        self.ffn_module2 = nn.Sequential(nn.LayerNorm(d_model), _FFN(d_ffn, d_model, dropout, activation), nn.Dropout(dropout))
        self.norm1 = _LayerNorm(d_model)
        self.norm2 = _LayerNorm(d_model)
        self.drop = nn.Dropout(dropout)

    def make_run(self, B, T, m8, src_mask, chunk, compute_dtype=None, next_layer=None):
        """compute_dtype: dtype of the GEMM operands when the incoming stream x3 is the float32 residual stream of a bf16
        model (functional.RESIDUAL_F32); None = everything in x3.dtype.
        next_layer: the layer that consumes this one's output inside an encoder stack.  Its first LayerNorm (ffn_module1's) then
        runs in the same pass as this layer's norm2 where the shapes allow (ops.layernorm_fwd_pair: one read of the float32 stream
        for both); run(..., with_post=True) then returns a third value, (LN(y), stats) | None, which the stack hands to the next
        layer's run as `pre_ln`."""
        d_act = self.act
        P1, P2 = _ffn_params(self.ffn_module1), _ffn_params(self.ffn_module2)
        Pc = self.convolution_module.params()
        n1, n2 = self.norm1.norm, self.norm2.norm
        pd = self.p_drop if self.training else 0.0
        cell = F.cell_run(self.mha_layer._params(), self.mha_layer._cfg(), B, T, m8, src_mask,
                          self.mha_layer.global_dropout if self.training else 0.0)

        Pn = _ffn_params(next_layer.ffn_module1) if next_layer is not None else None

        def run(x3, need, pre_ln=None, with_post=False):
            dtype = compute_dtype or x3.dtype
            x = ops.rows2d(x3)
            post_next = None
            # every LayerNorm that follows a Linear of width d_model = 256 runs in that GEMM's epilogue (`post` = its output and
            # statistics, None when the shape does not qualify and the consumer runs the LayerNorm kernel itself)
            y1, b1, post1 = F.ffn_module_fwd(x, P1, d_act, need, dtype, p=pd, pre_ln=pre_ln, ln_next=(n1.weight, n1.bias, n1.eps))   # :507
            h, bn1 = F.ln_fwd(y1, n1.weight, n1.bias, n1.eps, need, pre=post1, out_dtype=dtype)         # :510
            y2_3, bcell, post2 = cell(h.view(B, T, -1), need, res=y1, ln_next=(Pc["ln1_w"], Pc["ln1_b"], 1e-5))   # :512-530
            y2 = ops.rows2d(y2_3)
            y3, bconv, post3 = F.conv_module_fwd(y2, Pc, d_act, m8, B, T, need, dtype, chunk, p=pd, pre_ln=post2,
                                                 ln_next=(P2["ln_w"], P2["ln_b"], 1e-5))                          # :532-534
            # (norm2's output is the layer output = the next layer's residual stream: stream dtype, 4th element of ln_next)
            # (inside a stack: where norm2 rides in the second FFN's down-projection, the NEXT layer's first LayerNorm can ride with it)
            y4, bf2, post4, post_next = F.ffn_module_fwd(y3, P2, d_act, need, dtype, p=pd, pre_ln=post3, ln_next=(n2.weight, n2.bias, n2.eps, True),
                                                         ln_pair=(Pn["ln_w"], Pn["ln_b"], 1e-5) if Pn is not None else ())
            if (post4 is None and Pn is not None and y4.dtype != dtype and
                    ops.layernorm_pair_ok(y4, dtype, n2.weight, n2.bias, Pn["ln_w"], Pn["ln_b"])):
                # norm2 and the next layer's first LayerNorm in one pass over the float32 stream (equal to two launches to an ulp)
                y5_, st1, hn, st2 = ops.layernorm_fwd_pair(y4, n2.weight.detach(), n2.bias.detach(), n2.eps, Pn["ln_w"].detach(),
                                                           Pn["ln_b"].detach(), 1e-5, need, dtype)
                post4, post_next = (y5_, st1), (hn, st2)
            y5, bn2 = F.ln_fwd(y4, n2.weight, n2.bias, n2.eps, need, pre=post4)        # :536
            if not need:
                return (y5.view(B, T, -1), None, post_next) if with_post else (y5.view(B, T, -1), None)

            def bwd(dy3):
                dy = ops.rows2d(dy3 if dy3.is_contiguous() else dy3.contiguous())
                if dy.dtype != dtype:                      # (a float32 gradient from autograd for the float32 stream: gradients run in the compute dtype)
                    dy = ops.cast(dy, dtype)
                # each LayerNorm backward also writes what the NEXT block applies first to its gradient (`pre`: the FFN's
                # 1/2 * dropout, the conv module's dropout * padding mask) - no separate elementwise pass per module
                pre_c = getattr(bconv, "pre", None)
                d4, d4z = bn2(dy, second=bf2.pre)
                # the cell's own first step, dy * act'(zm), as the second output of the conv module's fused LayerNorm backward
                pre_cell = getattr(bcell, "pre", None) if getattr(bconv, "ln1_fused", False) else None
                d3z = None
                if pre_c is not None:
                    d3, d3z = bf2(d4, dz_in=d4z, second=pre_c)
                else:
                    d3 = bf2(d4, dz_in=d4z)
                d2z = None
                if pre_cell is not None:
                    d2, d2z = bconv(d3, dz_in=d3z, second=pre_cell)
                else:
                    d2 = bconv(d3, dz_in=d3z)
                if getattr(bcell, "can_fuse_ln", False) and F.ln_fusable(bn1.spec, d2.shape[0], d2.shape[1], dtype, bcell.ln_reduce, bcell.ln_W):
                    # norm1's backward (+ the skip gradient, + FFN1's 1/2 * dropout) in the epilogue of the cell's input dgrad
                    d1, d1z = bcell(d2.view(B, T, -1), ln=bn1.spec, ln_res=d2, ln_second=b1.pre, dz_in=d2z)
                else:
                    dh = ops.rows2d(bcell(d2.view(B, T, -1), dz_in=d2z))
                    d1, d1z = bn1(dh, res=d2, second=b1.pre)                           # skip gradient fused
                return b1(d1, dz_in=d1z).view(B, T, -1)
            return (y5.view(B, T, -1), bwd, post_next) if with_post else (y5.view(B, T, -1), bwd)
        return run

    def forward(self, x, src_mask: Optional[torch.Tensor] = None, src_key_padding_mask: Optional[torch.Tensor] = None,
                pos_embs: torch.Tensor = None, dynchunktrain_config=None):
        B, T, _ = x.shape
        m8 = F.mask_u8(src_key_padding_mask, B, T, x.device)
        chunk = dynchunktrain_config.chunk_size if dynchunktrain_config is not None else 0
        stream = F.stream_dtype(x.dtype)
        if stream == x.dtype:
            return F.block(x, self.make_run(B, T, m8, src_mask, chunk), list(self.parameters())), None
        # a bf16 layer called on its own: the same float32 residual stream as inside the encoder stack (cast in, cast out)
        inner = self.make_run(B, T, m8, src_mask, chunk, compute_dtype=x.dtype)
        d = x.shape[2]

        def run(xin, need):
            y, b = inner(ops.cast(ops.rows2d(xin), stream).view(B, T, d), need)
            return ops.cast(ops.rows2d(y), xin.dtype).view(B, T, d), b
        return F.block(x, run, list(self.parameters())), None

    def forward_streaming(self, *a, **k):
        raise NotImplementedError("streaming inference is broken for SummaryMixing in the reference (SURVEY §2 row 5)")


class ConformerEncoder(nn.Module):
    def __init__(self, num_layers, d_model, d_ffn, nhead, kernel_size=31, kdim=None, vdim=None, activation=Swish,
                 bias=True, dropout=0.0, causal=False, attention_type="RelPosMHAXL", local_proj_hid_dim=[512],
                 local_proj_out_dim=512, summary_hid_dim=[1024], mode="SummaryMixing"):
        super().__init__()
        self.layers = nn.ModuleList([
            ConformerEncoderLayer(d_ffn=d_ffn, nhead=nhead, d_model=d_model, kdim=kdim, vdim=vdim, dropout=dropout,
                                  activation=activation, kernel_size=kernel_size, bias=bias, causal=causal,
                                  attention_type=attention_type, local_proj_hid_dim=local_proj_hid_dim,
                                  local_proj_out_dim=local_proj_out_dim, summary_hid_dim=summary_hid_dim, mode=mode)
            for _ in range(num_layers)])
        self.norm = _LayerNorm(d_model, eps=1e-6)
        self.attention_type = attention_type

    def forward(self, src, src_mask: Optional[torch.Tensor] = None, src_key_padding_mask: Optional[torch.Tensor] = None,
                pos_embs: Optional[torch.Tensor] = None, dynchunktrain_config=None, _compute_dtype=None):
        B, T, d = src.shape
        m8 = F.mask_u8(src_key_padding_mask, B, T, src.device)
        chunk = dynchunktrain_config.chunk_size if dynchunktrain_config is not None else 0
        # compute dtype: what the caller says (TransformerASR.encode hands over the float32 stream of a bf16 model), else the input's
        out = F.encoder_stack(src, list(self.layers),
                              lambda layer, compute, nxt=None: layer.make_run(B, T, m8, src_mask, chunk, compute_dtype=compute, next_layer=nxt),
                              self.norm.norm, list(self.parameters()), _compute_dtype, pair_next=True)
        return out, [None] * len(self.layers)
