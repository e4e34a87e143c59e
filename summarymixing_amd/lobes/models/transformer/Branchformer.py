"""Branchformer encoder with the SummaryMixing cell on MI355X.

Mirrors speechbrain/lobes/models/transformer/Branchformer.py (ConvolutionBranch :31-97,
BranchformerEncoderLayer :100-334, BranchformerEncoder :337-491) for attention_type="SummaryMixing":

  layer(x) = x + merge_proj(cat[ SummaryMixing(LN(x)), cgMLP(LN(x)) ])
  cgMLP(h) = Linear(csgu/2 -> d)( CSGU( act(Linear(d -> csgu)(h)) ) ),
  CSGU(u)  = u1 * dwconv_reflect(LN(u2))                      (upstream ConvolutionalSpatialGatingUnit)
The concatenation is never materialised: both branches write into the two column halves of one buffer.
"""
import os
from typing import Optional

import torch
import torch.nn as nn

from .... import _lib as L
from .... import functional as F
from .... import ops
from .... import sequence_parallel as SP
from ....nnet.activations import act_code
from ....nnet.summary_mixing import SummaryMixing
from ...models.VanillaNN import VanillaNN
from .Conformer import _LayerNorm

_SPLIT_MERGE_DGRAD = True   # A/B knob: the merge's input gradient as two GEMMs with fused first steps of their consumers
_PREACT_LN = True   # A/B knob: channel_proj1's activation backward inside the CSGU LayerNorm backward


class _CSGUConv(nn.Module):
    def __init__(self, n, k):
        super().__init__()
        self.conv = nn.Conv1d(n, n, k, groups=n)


class ConvolutionalSpatialGatingUnit(nn.Module):
    """Parameter holder with upstream's keys (norm.norm.*, conv.conv.*) and init (w ~ N(0,1e-6), bias = 1)."""

    def __init__(self, input_size, kernel_size=31, dropout=0.0, use_linear_after_conv=False, activation=nn.Identity):
        super().__init__()
        if input_size % 2 != 0:
            raise ValueError("Input size must be divisible by 2!")
        if use_linear_after_conv or act_code(activation) != L.ACT_NONE:
            raise NotImplementedError("CSGU linear-after-conv / gate activation are not used by the SummaryMixing recipes")
        n = input_size // 2
        self.norm = _LayerNorm(n)
        self.conv = _CSGUConv(n, kernel_size)
        nn.init.normal_(self.conv.conv.weight, std=1e-6)
        nn.init.ones_(self.conv.conv.bias)


class ConvolutionBranch(nn.Module):
    def __init__(self, input_size, linear_units=3072, kernel_size=31, activation=nn.GELU, gate_activation=nn.Identity,
                 dropout=0.0, use_linear_after_conv=False):
        super().__init__()
        self.pre_channel_proj = nn.Linear(input_size, linear_units)
        self.post_channel_proj = nn.Linear(linear_units // 2, input_size)
        self.act = act_code(activation)
        self.p_drop = float(dropout)                       # (the CSGU's own dropout on x1 * conv(x2), upstream CSGU.forward)
        self.csgu = ConvolutionalSpatialGatingUnit(linear_units, kernel_size, dropout, use_linear_after_conv,
                                                   gate_activation)

    def forward(self, x):
        """x (B, T, D) -> post_channel_proj(CSGU(act(pre_channel_proj(x)))) (reference Branchformer.py:86-97), standalone: the
        same kernels the fused layer path runs (projection GEMMs with fused bias / activation, LayerNorm, the rolling CSGU
        depthwise-conv kernel with its gate and dropout), forward and backward.  BranchformerEncoderLayer does NOT call this - it
        takes params() and fuses the branch with its neighbours (make_run)."""
        B, T, d = x.shape
        Pb = self.params()
        act = self.act
        pd = self.p_drop if self.training else 0.0

        def run(x3, need):
            dtype = x3.dtype
            xr = ops.rows2d(x3 if x3.is_contiguous() else x3.contiguous())
            Wpre, Wpost = F.wcast(Pb["Wpre"], dtype), F.wcast(Pb["Wpost"], dtype)
            u, zu = F.linear_fwd(xr, Wpre, Pb["bpre"], act, None, save_z=need, wparam=Pb["Wpre"])       # (N, linear_units)
            n = u.shape[1] // 2
            u1, u2 = u[:, :n], u[:, n:]
            v, bnv = F.ln_fwd(u2, Pb["ln_w"], Pb["ln_b"], 1e-5, need)
            k = Pb["wd"].shape[-1]
            wd = Pb["wd"].detach().reshape(n, k)
            sd = ops.new_dropout_seed() if pd > 0.0 else None
            g = ops.dwconv_fwd(v, wd, Pb["bd"].detach(), B, T, n, k, False, L.PAD_REFLECT, 0, gate=u1,
                               drop=(pd, sd) if pd > 0.0 else None)
            y, _ = F.linear_fwd(g, Wpost, Pb["bpost"])
            if not need:
                return y.view(B, T, d), None

            def bwd(dy3):
                dy = ops.rows2d(dy3 if dy3.is_contiguous() else dy3.contiguous())
                dg, _ = F.linear_bwd(dy, g, Wpost, None, L.ACT_NONE, None, 1.0, F.gacc(Pb["Wpost"]), F.gacc(Pb["bpost"]),
                                     dx_drop=(pd, sd) if pd > 0.0 else None)
                du = torch.empty_like(u)                   # [d gate | d LN input]
                dv, _ = F.dwconv_bwd_deferred(dg, v, wd, Pb["bd"].detach(), F.gacc(Pb["wd"]).view(n, k), F.gacc(Pb["bd"]), B, T,
                                              n, k, False, L.PAD_REFLECT, 0, gate=u1, dgate_out=du[:, :n])
                bnv(dv, out=du[:, n:])
                dx, _ = F.linear_bwd(du, xr, Wpre, zu, act, None, 1.0, F.gacc(Pb["Wpre"]), F.gacc(Pb["bpre"]))
                return dx.view(B, T, d)
            return y.view(B, T, d), bwd
        return F.block(x, run, list(self.parameters()))

    def params(self):
        return {"Wpre": self.pre_channel_proj.weight, "bpre": self.pre_channel_proj.bias,
                "Wpost": self.post_channel_proj.weight, "bpost": self.post_channel_proj.bias,
                "ln_w": self.csgu.norm.norm.weight, "ln_b": self.csgu.norm.norm.bias,
                "wd": self.csgu.conv.conv.weight, "bd": self.csgu.conv.conv.bias}


class BranchformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, kernel_size=31, kdim=None, vdim=None, activation=nn.GELU, dropout=0.0,
                 attention_type="SummaryMixing", csgu_linear_units=3072, gate_activation=nn.Identity,
                 use_linear_after_conv=False, local_proj_hid_dim=[512], local_proj_out_dim=512, summary_hid_dim=[1024],
                 summary_out_dim=1024, mode="SummaryMixing"):
        super().__init__()
        if attention_type != "SummaryMixing":
            raise NotImplementedError("summarymixing_amd implements attention_type='SummaryMixing' only")
        self.attention_type, self.mode = attention_type, mode
        self.act = act_code(activation)
        self.p_drop = float(dropout)
        self.mha_layer = SummaryMixing(enc_dim=d_model, nhead=nhead, local_proj_hid_dim=local_proj_hid_dim,
                                       local_proj_out_dim=local_proj_out_dim, summary_hid_dim=summary_hid_dim,
                                       summary_out_dim=summary_out_dim, activation=activation, mode=mode)
        # (global_dropout stays at the cell's default 0.1 whatever the layer dropout is, as in the reference:
        #  Branchformer.py:209-218 does not pass it)
        self.merge_dnn_blocks = list(summary_hid_dim) + [d_model]
        self.merge_proj = VanillaNN(input_shape=[None, None, local_proj_out_dim + summary_out_dim],
                                    dnn_blocks=len(self.merge_dnn_blocks), dnn_neurons=self.merge_dnn_blocks,
                                    activation=activation)
        self.norm_mhsa = _LayerNorm(d_model)
        self.convolution_branch = ConvolutionBranch(d_model, csgu_linear_units, kernel_size, activation, gate_activation,
                                                    dropout, use_linear_after_conv)
        self.norm_conv = _LayerNorm(d_model)
        self.dropout = nn.Dropout(dropout)

    def make_run(self, B, T, m8, src_mask, compute_dtype=None):
        sp = SP.enabled()       # time axis sharded over the ranks: the CSGU's depthwise conv gets halos (below), the cell its all-reduce
        act = self.act
        pd = self.p_drop if self.training else 0.0
        cell = F.cell_run(self.mha_layer._params(), self.mha_layer._cfg(), B, T, m8, src_mask,
                          self.mha_layer.global_dropout if self.training else 0.0)
        Pb = self.convolution_branch.params()
        merge = self.merge_proj.specs()
        nm, nc = self.norm_mhsa.norm, self.norm_conv.norm
        s_out = self.mha_layer.summary_out_dim if self.mode != "SummaryMixing-lite" else self.mha_layer.summary_out_dim

        def run(x3, need):
            dtype = compute_dtype or x3.dtype              # (x3 may be the float32 residual stream of a bf16 model)
            x = ops.rows2d(x3)
            N, d = x.shape
            dev = x.device
            # branch 1: SummaryMixing(LN(x))
            h1, bn1 = F.ln_fwd(x, nm.weight, nm.bias, nm.eps, need, out_dtype=dtype)
            # training: the dropout of the cell's output (:279) and its placement in the merge input ride in the cell's last
            # GEMM epilogue (and the dropout backward in the cell's first backward pass)
            fuse_y1 = pd > 0.0 and self.mode != "SummaryMixing-lite"
            cat = sd1 = None
            if fuse_y1:
                c1 = self.mha_layer._params()["summary_local_merging"][0]["W"].shape[0]
                cat = torch.empty((N, c1 + d), dtype=dtype, device=dev)
                sd1 = ops.new_dropout_seed()
                y1_3, bcell = cell(h1.view(B, T, d), need, out=cat[:, :c1], out_drop=(pd, sd1))
            else:
                y1_3, bcell = cell(h1.view(B, T, d), need)
            y1 = ops.rows2d(y1_3.contiguous() if y1_3.stride(1) == 0 else y1_3)
            c1 = y1.shape[1]
            # branch 2: cgMLP(LN(x))
            h2, bn2 = F.ln_fwd(x, nc.weight, nc.bias, nc.eps, need, out_dtype=dtype)
            Wpre, Wpost = F.wcast(Pb["Wpre"], dtype), F.wcast(Pb["Wpost"], dtype)
            u, zu = F.linear_fwd(h2, Wpre, Pb["bpre"], act, None, save_z=need, wparam=Pb["Wpre"])           # (N, csgu)
            n = u.shape[1] // 2
            u1, u2 = u[:, :n], u[:, n:]
            v, bnv = F.ln_fwd(u2, Pb["ln_w"], Pb["ln_b"], 1e-5, need)
            k = Pb["wd"].shape[-1]
            wd = Pb["wd"].detach().reshape(n, k)
            # (the CSGU's own dropout on x1 * conv(x2), upstream CSGU.forward, rides in the kernel where it can)
            sd4 = ops.new_dropout_seed() if pd > 0.0 else None
            if sp:
                # sequence-parallel: the conv input v = LN(u2) of the (k-1)/2 frames either side comes from the neighbour ranks
                # (one small all-gather, sequence_parallel.exchange_halos); at the two ends of the WHOLE sequence the halo rows
                # hold the reflected frames (Branchformer.py:31-97 pads by reflection), so a zero-padded conv over the extended
                # shard equals the reflect-padded conv over the whole sequence; only the centre T rows go on
                H = (k - 1) // 2
                if T <= H:
                    raise ValueError(f"sequence-parallel shards need more than {H} frames per rank (got {T})")
                Te = T + 2 * H
                first_rank, last_rank = SP.rank() == 0, SP.rank() == SP.world() - 1
                v3 = v.view(B, T, n)
                lh, rh = SP.exchange_halos(v3[:, :H], v3[:, T - H:])
                if first_rank:
                    lh = v3[:, 1:H + 1].flip(1)            # frame -j = frame j
                if last_rank:
                    rh = v3[:, T - 1 - H:T - 1].flip(1)    # frame T-1+j = frame T-1-j
                ve = torch.cat([lh, v3, rh], 1).contiguous().view(B * Te, n)
                u1e = torch.zeros((B, Te, n), dtype=u.dtype, device=dev)
                u1e[:, H:H + T] = u1.reshape(B, T, n)
                u1e = u1e.view(B * Te, n)
                ge = ops.dwconv_fwd(ve, wd, Pb["bd"].detach(), B, Te, n, k, False, L.PAD_ZERO, 0, gate=u1e)
                g = ge.view(B, Te, n)[:, H:H + T].contiguous().view(N, n)
            else:
                g = ops.dwconv_fwd(v, wd, Pb["bd"].detach(), B, T, n, k, False, L.PAD_REFLECT, 0, gate=u1,
                                   drop=(pd, sd4) if pd > 0.0 else None)
            # both branches land in one (N, c1 + d) buffer = the merge input (no torch.cat)
            if cat is None:
                cat = torch.empty((N, c1 + d), dtype=dtype, device=dev)
            sd2 = sd3 = None
            if pd > 0.0:                                    # dropout on both branches and on the merge (:279,295,334)
                if not fuse_y1:
                    sd1 = ops.new_dropout_seed()
                    ops.dropout(y1, pd, sd1, out=cat[:, :c1])
                sd2, sd3 = (ops.new_dropout_seed() for _ in range(2))
            else:
                ops.axpby(1.0, y1, out=cat[:, :c1])
            F.linear_fwd(g, Wpost, Pb["bpost"], out=cat[:, c1:], drop=(pd, sd2) if pd > 0.0 else None)   # Linear + dropout
            mdrop = (pd, sd3) if pd > 0.0 else None
            if merge[-1]["kind"] == "linear":              # x + dropout(merge(cat)) (:279) in the last Linear's epilogue
                y, sv_m = F.mlp_fwd(cat, merge, act, None, need, dtype, last_res=x, last_drop=mdrop)
            else:
                m_out, sv_m = F.mlp_fwd(cat, merge, act, None, need, dtype)
                if pd > 0.0:
                    ops.dropout(m_out, pd, sd3, out=m_out)
                y = ops.axpby(1.0, x, 1.0, ops.cast(m_out, x.dtype))
            if not need:
                return y.view(B, T, d), None

            def bwd(dy3):
                dy = ops.rows2d(dy3 if dy3.is_contiguous() else dy3.contiguous())
                if dy.dtype != dtype:
                    dy = ops.cast(dy, dtype)
                cpre = getattr(bcell, "pre", None)         # (1, None, None, zm, act): the cell's first backward step is dy * act'(zm)
                split = (_SPLIT_MERGE_DGRAD and merge[0]["kind"] == "linear" and merge[-1]["kind"] == "linear" and cpre is not None
                         and (pd == 0.0 or fuse_y1) and c1 % 8 == 0 and d % 8 == 0)
                d1z = None
                if split:
                    # the merge's input gradient as two GEMMs over the column halves of its first weight: the half that goes to
                    # the cell leaves the epilogue as the cell's dZ_m = D(.) * act'(z_m) (dropout of the cell output + the cell's
                    # own first backward step), the cgMLP half with its dropout backward applied - no elementwise pass over either
                    e1 = dict(act=cpre[4], act_grad_z=cpre[3], drop=(pd, sd1) if pd > 0.0 else None)
                    e2 = dict(drop=(pd, sd2)) if pd > 0.0 else None
                    d1z, dgz = F.mlp_bwd(dy, merge, act, sv_m, dtype, last_drop=mdrop, dx_split=[(0, c1, e1), (c1, c1 + d, e2)])
                    dg, _ = F.linear_bwd(dgz, g, Wpost, None, L.ACT_NONE, None, 1.0, F.gacc(Pb["Wpost"]), F.gacc(Pb["bpost"]),
                                         dz_ready=True, dx_drop=(pd, sd4) if pd > 0.0 else None, wparam=Pb["Wpost"])
                else:
                    if merge[-1]["kind"] == "linear":
                        dcat = F.mlp_bwd(dy, merge, act, sv_m, dtype, last_drop=mdrop)
                    else:
                        dcat = F.mlp_bwd(ops.dropout(dy, pd, sd3) if pd > 0.0 else dy, merge, act, sv_m, dtype)
                    # the cell's gradient: a contiguous (N, c1) copy either way, the dropout backward rides in it
                    d1 = dcat[:, :c1] if fuse_y1 else (ops.dropout(dcat[:, :c1], pd, sd1) if pd > 0.0 else dcat[:, :c1].contiguous())
                    # branch 2 backward (the dropout backward of its half rides in linear_bwd's activation/mask pass)
                    dg, _ = F.linear_bwd(dcat[:, c1:], g, Wpost, None, L.ACT_NONE, None, 1.0, F.gacc(Pb["Wpost"]), F.gacc(Pb["bpost"]),
                                         drop=(pd, sd2) if pd > 0.0 else None, dx_drop=(pd, sd4) if pd > 0.0 else None, wparam=Pb["Wpost"])
                du = torch.empty_like(u)                   # [d gate | d LN input]: both kernels write their half directly
                if sp:
                    # the transposed exchange: gradients of the halo rows go back to the ranks that own those frames, gradients
                    # of the reflected rows at the sequence ends fold back onto their source frames
                    dge = torch.zeros((B, Te, n), dtype=dg.dtype, device=dev)     # (the halo OUTPUTS were dropped: zero gradient)
                    dge[:, H:H + T] = dg.view(B, T, n)
                    dve, dgate_e = ops.dwconv_bwd(dge.view(B * Te, n), ve, wd, Pb["bd"].detach(), F.gacc(Pb["wd"]).view(n, k),
                                                  F.gacc(Pb["bd"]), B, Te, n, k, False, L.PAD_ZERO, 0, gate=u1e)
                    dve3 = dve.view(B, Te, n)
                    dv3 = dve3[:, H:H + T].contiguous()
                    gl, gr = dve3[:, :H], dve3[:, Te - H:]
                    if first_rank:
                        dv3[:, 1:H + 1] += gl.flip(1)
                        gl = torch.zeros_like(gl)
                    if last_rank:
                        dv3[:, T - 1 - H:T - 1] += gr.flip(1)
                        gr = torch.zeros_like(gr)
                    g_first, g_last = SP.return_halo_grads(gl.contiguous(), gr.contiguous())
                    dv3[:, :H] += g_first
                    dv3[:, T - H:] += g_last
                    dv = dv3.view(N, n)
                    du[:, :n] = dgate_e.view(B, Te, n)[:, H:H + T].reshape(N, n)
                else:
                    dv, _ = F.dwconv_bwd_deferred(dg, v, wd, Pb["bd"].detach(), F.gacc(Pb["wd"]).view(n, k), F.gacc(Pb["bd"]), B, T,
                                                  n, k, False, L.PAD_REFLECT, 0, gate=u1, dgate_out=du[:, :n])
                if _PREACT_LN and zu is not None and act != L.ACT_NONE and bnv.spec["act"] == L.ACT_NONE and ops.layernorm_bwd_preact_ok(dv, u2, zu[:, n:], du[:, n:]):
                    # the activation backward of channel_proj1 rides in the CSGU LayerNorm's backward for the normalised half
                    # (dZ = act'(z) * LNbwd), and runs in place on the gate half only: no pass over the whole (N, csgu) gradient
                    bnv(dv, out=du[:, n:], preact=(zu[:, n:], act))
                    ops.act_mask_bwd(du[:, :n], zu[:, :n], None, act, 1.0, du[:, :n], None)
                    dh2, _ = F.linear_bwd(du, h2, Wpre, None, act, None, 1.0, F.gacc(Pb["Wpre"]), F.gacc(Pb["bpre"]), dz_ready=True)
                else:
                    bnv(dv, out=du[:, n:])
                    dh2, _ = F.linear_bwd(du, h2, Wpre, zu, act, None, 1.0, F.gacc(Pb["Wpre"]), F.gacc(Pb["bpre"]))
                dx = bn2(dh2, res=dy)
                # branch 1 backward
                dh1 = ops.rows2d(bcell(d1z.view(B, T, c1), dz_in=d1z) if d1z is not None else bcell(d1.view(B, T, c1)))
                dx = bn1(dh1, res=dx)
                return dx.view(B, T, d)
            return y.view(B, T, d), bwd
        return run

    def forward(self, x, src_mask: Optional[torch.Tensor] = None, src_key_padding_mask: Optional[torch.Tensor] = None,
                pos_embs: Optional[torch.Tensor] = None):
        B, T, d = x.shape
        m8 = F.mask_u8(src_key_padding_mask, B, T, x.device)
        stream = F.stream_dtype(x.dtype)
        if stream == x.dtype:
            return F.block(x, self.make_run(B, T, m8, src_mask), list(self.parameters())), None
        inner = self.make_run(B, T, m8, src_mask, compute_dtype=x.dtype)     # a bf16 layer on its own: float32 stream inside

        def run(xin, need):
            y, b = inner(ops.cast(ops.rows2d(xin), stream).view(B, T, d), need)
            return ops.cast(ops.rows2d(y), xin.dtype).view(B, T, d), b
        return F.block(x, run, list(self.parameters())), None


class BranchformerEncoder(nn.Module):
    def __init__(self, num_layers, d_model, nhead, kernel_size=31, kdim=None, vdim=None, activation=nn.GELU, dropout=0.0,
                 attention_type="SummaryMixing", csgu_linear_units=3072, gate_activation=nn.Identity,
                 use_linear_after_conv=False, local_proj_hid_dim=[512], local_proj_out_dim=512, summary_hid_dim=[1024],
                 summary_out_dim=1024, mode="SummaryMixing"):
        super().__init__()
        self.layers = nn.ModuleList([
            BranchformerEncoderLayer(nhead=nhead, d_model=d_model, kdim=kdim, vdim=vdim, dropout=dropout,
                                     activation=activation, kernel_size=kernel_size, attention_type=attention_type,
                                     csgu_linear_units=csgu_linear_units, gate_activation=gate_activation,
                                     use_linear_after_conv=use_linear_after_conv, local_proj_hid_dim=local_proj_hid_dim,
                                     local_proj_out_dim=local_proj_out_dim, summary_hid_dim=summary_hid_dim,
                                     summary_out_dim=summary_out_dim, mode=mode) for _ in range(num_layers)])
        self.norm = _LayerNorm(d_model, eps=1e-6)
        self.attention_type = attention_type

    def forward(self, src, src_mask: Optional[torch.Tensor] = None, src_key_padding_mask: Optional[torch.Tensor] = None,
                pos_embs: Optional[torch.Tensor] = None, dynchunktrain_config=None, _compute_dtype=None):
        if dynchunktrain_config is not None:
            raise NotImplementedError("Dynamic chunk training is not supported for the Branchformer (as the reference)")
        B, T, _ = src.shape
        m8 = F.mask_u8(src_key_padding_mask, B, T, src.device)
        out = F.encoder_stack(src, list(self.layers), lambda layer, compute: layer.make_run(B, T, m8, src_mask, compute_dtype=compute),
                              self.norm.norm, list(self.parameters()), _compute_dtype)
        return out, [None] * len(self.layers)
