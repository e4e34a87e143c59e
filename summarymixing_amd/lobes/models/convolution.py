"""Conv2d subsampling front-end on MI355X (surface of speechbrain.lobes.models.convolution.ConvolutionFrontEnd as the
recipes use it: 2 blocks, out_channels (64, 32), 3x3 kernels, strides (2, 2), ...transducer.yaml:247-254):

    block:  Conv2d(3x3, stride 2, reflect pad 1) -> LayerNorm over (F', C) -> LeakyReLU(0.01) -> Dropout

Each conv is im2col (channels-last gather kernel) + the MFMA GEMM with bias epilogue; LayerNorm+LeakyReLU is one
kernel; the backward runs wgrad / dgrad GEMMs and a gather-form col2im.  Output (B, ceil(T/4), 20, 32) feeds
TransformerASR.encode, which flattens it to 640 features (TransformerASR.py:528-530).  Arithmetic spec:
oracle/smx_oracle.py::conv_frontend (parity unpinned: upstream SpeechBrain code)."""
import torch
from torch import nn

from ... import _lib as L
from ... import functional as F
from ... import ops


import os
_FUSED_BLOCK1 = True      # A/B knob: smx_conv1_ln_fwd / _bwd instead of im2col + Linear + LayerNorm
_DIRECT_CONV2 = True            # A/B knob: forward / wgrad GEMMs gather from the input, no im2col
_DIRECT_DGRAD = True      # A/B knob: smx_conv2d_s2_dgrad instead of GEMM + col2im


class _Conv2dHolder(nn.Module):
    """Key layout of speechbrain.nnet.CNN.Conv2d: ``.conv`` is the nn.Conv2d."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.conv = nn.Conv2d(c_in, c_out, 3, stride=2)            # parameter holder (Cout, Cin, 3, 3)


class _NormHolder(nn.Module):
    """Key layout of speechbrain.nnet.normalization.LayerNorm: ``.norm`` is the nn.LayerNorm."""

    def __init__(self, shape):
        super().__init__()
        self.norm = nn.LayerNorm(shape)                             # affine over (F', C) like upstream's LayerNorm


class _ConvBlock(nn.Module):
    """Parameter holder with the state-dict keys of upstream's ConvBlock (one layer per block):
    ``convs.conv_0.conv.{weight,bias}``, ``convs.norm_0.norm.{weight,bias}`` - a recipe checkpoint's ``CNN`` recoverable
    (...transducer.yaml:247-254,407-413) loads with strict=True."""

    def __init__(self, c_in, c_out, f_in):
        super().__init__()
        self.f_out = (f_in + 1) // 2
        self.convs = nn.ModuleDict({"conv_0": _Conv2dHolder(c_in, c_out), "norm_0": _NormHolder((self.f_out, c_out))})
        self.c_in, self.c_out = c_in, c_out
        self.kp = (9 * c_in + 7) // 8 * 8

    @property
    def conv(self):
        return self.convs["conv_0"].conv

    @property
    def norm(self):
        return self.convs["norm_0"].norm


class ConvolutionFrontEnd(nn.Module):
    def __init__(self, input_shape, num_blocks=2, num_layers_per_block=1, out_channels=(64, 32), kernel_sizes=(3, 3),
                 strides=(2, 2), dilations=(1, 1), residuals=(False, False), conv_module=None, activation=nn.LeakyReLU,
                 norm=None, dropout=0.1, conv_bias=True, padding="same", conv_init=None):
        super().__init__()
        if (num_layers_per_block != 1 or tuple(kernel_sizes) != (3,) * num_blocks or tuple(strides) != (2,) * num_blocks or
                any(residuals) or any(d != 1 for d in dilations)):
            raise NotImplementedError("only the recipes' configuration is built: 3x3 kernels, stride 2, one layer per block")
        f, c = input_shape[-1], 1
        self.num_blocks = num_blocks
        for i in range(num_blocks):                                 # upstream: Sequential layers named convblock_{i}
            blk = _ConvBlock(c, out_channels[i], f)
            self.add_module(f"convblock_{i}", blk)
            f, c = blk.f_out, out_channels[i]
        self.p_drop = float(dropout)

    @property
    def blocks(self):
        return [getattr(self, f"convblock_{i}") for i in range(self.num_blocks)]

    def state_dict_for_oracle(self):
        sd = {}
        for i, b in enumerate(self.blocks):
            sd[f"convblock_{i}.conv.weight"], sd[f"convblock_{i}.conv.bias"] = b.conv.weight.detach(), b.conv.bias.detach()
            sd[f"convblock_{i}.norm.weight"], sd[f"convblock_{i}.norm.bias"] = b.norm.weight.detach(), b.norm.bias.detach()
        return sd

    def forward(self, x):
        """x (B, T, F) log-mel features (GPU, float32 or bfloat16) -> (B, ceil(T/4), F/4, C_last)."""
        B = x.shape[0]
        pd = self.p_drop if self.training else 0.0
        blocks = list(self.blocks)

        def run(xin, need):
            dtype = xin.dtype
            h = xin.reshape(B, xin.shape[1], xin.shape[2], 1).contiguous()
            saved = []
            for bi, blk in enumerate(blocks):
                _, T_, F_, C = h.shape
                T2, F2 = (T_ + 1) // 2, (F_ + 1) // 2
                if bi == 0 and C == 1 and _FUSED_BLOCK1 and ops.conv1_ln_ok(h.view(B, T_, F_), blk.c_out):
                    # first block in ONE pass (conv + LayerNorm + LeakyReLU; the backward recomputes the conv from x)
                    x3 = h.view(B, T_, F_)
                    w9 = blk.conv.weight.detach().reshape(blk.c_out, 9).float().contiguous()
                    a, st = ops.conv1_ln_fwd(x3, w9, blk.conv.bias.detach(), blk.norm.weight.detach().view(-1),
                                             blk.norm.bias.detach().view(-1), blk.norm.eps, L.ACT_LEAKY_RELU, need)
                    seed = None
                    if pd > 0.0:
                        seed = ops.new_dropout_seed()
                        ops.dropout(a, pd, seed, out=a)
                    saved.append(("fused1", x3, w9, st, seed, blk, bi))
                    h = a.view(B, T2, F2, blk.c_out)
                    continue
                # GEMM-layout weight (Cout, Kp): column (dt*3+df)*Cin + c  <- conv.weight (Cout, Cin, 3, 3)
                wg = torch.zeros((blk.c_out, blk.kp), dtype=torch.float32, device=h.device)
                wg[:, :9 * C] = blk.conv.weight.detach().permute(0, 2, 3, 1).reshape(blk.c_out, 9 * C)
                wgc = ops.cast(wg, dtype)
                nocol = _DIRECT_CONV2 and ops.conv2d_s2_direct_ok(h, blk.c_out) and ops.conv2d_s2_dgrad_ok(h, C, blk.c_out, T_, F_)
                if nocol:
                    # second block without the (rows, 9 C) patch matrix: the GEMMs gather from h (forward, wgrad; dgrad direct)
                    col = h
                    y = ops.conv2d_s2_fwd(h, wgc, blk.conv.bias.detach(), blk.c_out)
                else:
                    col = ops.im2col_s2(h, blk.kp)
                    y = ops.linear_k16(col, wgc, blk.conv.bias.detach()) if blk.kp == 16 else None   # first block: 9 taps, VALU
                    if y is None:
                        y, _ = F.linear_fwd(col, wgc, blk.conv.bias.detach())                     # (B*T2*F2, Cout)
                yr = y.view(B * T2, F2 * blk.c_out)
                a, ln_b = F.ln_fwd(yr, blk.norm.weight.view(-1), blk.norm.bias.view(-1), blk.norm.eps, need,
                                   L.ACT_LEAKY_RELU, wp=blk.norm.weight, bp=blk.norm.bias)
                seed = None
                if pd > 0.0:
                    seed = ops.new_dropout_seed()
                    ops.dropout(a, pd, seed, out=a)
                saved.append((h.shape, col, wgc, ln_b, seed, blk, bi if not nocol else -bi))
                h = a.view(B, T2, F2, blk.c_out)
            if not need:
                return h, None

            def bwd(dout):
                d = dout.contiguous()
                for shape, col, wgc, ln_b, seed, blk, bi in reversed(saved):
                    if isinstance(shape, str):                       # the fused first block: gradients from dA and x alone
                        x3, w9, st = col, wgc, ln_b
                        T2, F2 = (x3.shape[1] + 1) // 2, x3.shape[2] // 2
                        da = d.reshape(B * T2, F2 * blk.c_out)
                        if seed is not None:
                            da = ops.dropout(da, pd, seed)
                        gr = ops.conv1_ln_bwd(da, x3, w9, blk.conv.bias.detach(), blk.norm.weight.detach().view(-1),
                                              blk.norm.bias.detach().view(-1), st, L.ACT_LEAKY_RELU)
                        Dn = F2 * blk.c_out
                        for prm, piece in ((blk.norm.weight, gr[:Dn]), (blk.norm.bias, gr[Dn:2 * Dn]),
                                           (blk.conv.weight, gr[2 * Dn:2 * Dn + 9 * blk.c_out]),
                                           (blk.conv.bias, gr[2 * Dn + 9 * blk.c_out:])):
                            g = F.gacc(prm)
                            if g is not None:
                                g.add_(piece.view(g.shape))
                        return None
                    _, T_, F_, C = shape
                    T2, F2 = (T_ + 1) // 2, (F_ + 1) // 2
                    da = d.reshape(B * T2, F2 * blk.c_out)
                    if seed is not None:
                        da = ops.dropout(da, pd, seed)
                    dy = ln_b(da).view(B * T2 * F2, blk.c_out)
                    gw = torch.zeros((blk.c_out, blk.kp), dtype=torch.float32, device=dy.device)
                    if bi < 0:                                       # (saved without a patch matrix: col IS the block input)
                        gb = F.gacc(blk.conv.bias)
                        ops.conv2d_s2_wgrad(dy, col, gw, gb if gb is not None else torch.zeros(blk.c_out, device=dy.device))
                        g = F.gacc(blk.conv.weight)
                        if g is not None:
                            g.add_(gw[:, :9 * C].view(blk.c_out, 3, 3, C).permute(0, 3, 1, 2))
                        d = ops.conv2d_s2_dgrad(dy, wgc, B, T_, F_, C)
                        continue
                    first = bi == 0
                    # the second block's input gradient comes from the direct kernel (no (rows, 9 C) gradient matrix, no col2im)
                    direct = (not first) and _DIRECT_DGRAD and ops.conv2d_s2_dgrad_ok(dy, C, blk.c_out, T_, F_)
                    dcol, _ = F.linear_bwd(dy, col, wgc, None, L.ACT_NONE, None, 1.0, gw, F.gacc(blk.conv.bias),
                                           need_dx=not first and not direct)
                    g = F.gacc(blk.conv.weight)                       # fold the GEMM-layout gradient back (9*Cin*Cout values)
                    F.flush_deferred()                                # gw is a temporary: its slab reduction must have run
                    if g is not None:
                        g.add_(gw[:, :9 * C].view(blk.c_out, 3, 3, C).permute(0, 3, 1, 2))
                    if first:
                        return None
                    d = ops.conv2d_s2_dgrad(dy, wgc, B, T_, F_, C) if direct else ops.col2im_s2(dcol, B, T_, F_, C)
                return None
            return h, bwd
        return F.block(x, run, list(self.parameters()))
