"""SummaryMixing cell on MI355X (reference surface: speechbrain/nnet/summary_mixing.py:76-87,161).

Same constructor, same ``forward(x, sum_mask=None, src_padding_mask=None)``, same state-dict keys; the
arithmetic runs in libsmx.so (MFMA projection GEMMs with fused bias/activation/mask epilogues, the split-T
masked-mean kernel, the per-utterance summary folded into the merge GEMM as a side input so that neither the
``repeat`` (:222,:267) nor the ``cat`` (:238,:283) is ever materialised).
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import functional as F
from ..lobes.models.VanillaNN import VanillaNN
from .activations import act_code

MODES = ("SummaryMixing", "SummaryMixing-lite", "SummaryMixing-expdecay", "SummaryMixing-fast")


class SummaryMixing(nn.Module):
    def __init__(self, enc_dim, nhead, local_proj_hid_dim: Optional[list] = [512],
                 local_proj_out_dim: Optional[int] = 512, summary_hid_dim: Optional[list] = [512],
                 summary_out_dim: Optional[int] = 512, activation: Optional[nn.Module] = nn.GELU,
                 global_dropout: Optional[float] = 0.1, mode: Optional[str] = "SummaryMixing"):
        super().__init__()
        if mode not in MODES:
            raise ValueError("The SummaryMixing mode should either be 'SummaryMixing', 'SummaryMixing-lite', "
                             "'SummaryMixing-fast' or 'SummaryMixing-expdecay'")
        self.enc_dim, self.mode = enc_dim, mode
        self.local_proj_hid_dim, self.local_proj_out_dim = local_proj_hid_dim, local_proj_out_dim
        self.summary_hid_dim, self.summary_out_dim = summary_hid_dim, summary_out_dim
        self.act = act_code(activation)
        self.global_dropout = float(global_dropout)
        local_blocks = list(local_proj_hid_dim) + [local_proj_out_dim]
        summary_blocks = list(summary_hid_dim) + [summary_out_dim]
        shape = [None, None, enc_dim]
        # construction order follows the reference (:112-157) so that a shared RNG seed draws the same init
        if mode in ("SummaryMixing", "SummaryMixing-expdecay"):
            self.local_proj = VanillaNN(shape, activation, len(local_blocks), local_blocks, n_split=nhead)
            self.summary_local_merging = VanillaNN([None, None, local_proj_out_dim + summary_out_dim], activation, 1,
                                                   [summary_out_dim])
        if mode == "SummaryMixing-fast":
            self.global_proj = VanillaNN(shape, activation, 1, local_proj_out_dim * 2, n_split=1)
            self.summary_local_merging = VanillaNN([None, None, local_proj_out_dim * 2], activation, 1,
                                                   [summary_out_dim])
        else:
            self.summary_proj = VanillaNN(shape, activation, len(summary_blocks), summary_blocks, n_split=nhead)
        if mode == "SummaryMixing-expdecay":
            self.decay_constant = nn.Parameter(data=torch.tensor(0.995), requires_grad=False)
        for m in self.modules():                        # reference :159,:312-314: zero every nn.Linear bias
            if isinstance(m, nn.Linear):
                nn.init.zeros_(m.bias)

    def _params(self):
        P = {}
        for name in ("local_proj", "summary_proj", "global_proj", "summary_local_merging"):
            if hasattr(self, name):
                P[name] = getattr(self, name).specs()
        if hasattr(self, "decay_constant"):
            P["decay_constant"] = self.decay_constant
        return P

    def drop_p(self):
        """Dropout probability in effect (reference: nn.Dropout(global_dropout), active in train() only)."""
        return self.global_dropout if self.training else 0.0

    def _decay(self):
        """The frozen decay constant (summary_mixing.py:154-157) as a host float, read ONCE per parameter version: a
        .cpu() per forward would be a blocking device sync (and would break hipGraph capture)."""
        p = self.decay_constant
        key = (p._version, p.data_ptr())
        if getattr(self, "_decay_cache", (None, None))[0] != key:
            self._decay_cache = (key, float(p.detach().float().cpu()))
        return self._decay_cache[1]

    def _cfg(self):
        cfg = {"mode": self.mode, "act": self.act, "local_proj_out_dim": self.local_proj_out_dim}
        if hasattr(self, "decay_constant"):
            cfg["decay"] = self._decay
        return cfg

    def forward(self, x, sum_mask=None, src_padding_mask=None):
        """x (B,T,enc_dim) on the GPU, float32 or bfloat16; src_padding_mask (B,T) True = valid frame;
        sum_mask (T,T) tensor or functional.DynChunkMask."""
        B, T, _ = x.shape
        mask = F.mask_u8(src_padding_mask, B, T, x.device)
        run = F.cell_run(self._params(), self._cfg(), B, T, mask, sum_mask, self.drop_p())
        return F.block(x, run, list(self.parameters()))
