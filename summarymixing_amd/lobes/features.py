"""Log-mel filterbank front-end on MI355X (surface of speechbrain.lobes.features.Fbank as the recipes use it:
``Fbank(sample_rate=16000, n_fft=512, n_mels=80, win_length=32)``, ...transducer.yaml:171-175).

frame+window kernel -> DFT as an exact-fp32 MFMA GEMM against a [cos | -sin] basis -> fused power/mel/dB kernel ->
per-utterance top_db clamp.  No parameters, no backward.  Arithmetic spec: oracle/smx_oracle.py::fbank (the reference
pins nothing here: upstream SpeechBrain code)."""
import math

import torch
from torch import nn

from .. import _lib as L
from .. import ops


_FOLDED_DFT = True              # A/B knob: cosine / sine halves of half the length
_IMPLICIT_FRAMES = True   # A/B knob: the DFT GEMM reads the waveform in place


class Fbank(nn.Module):
    def __init__(self, deltas=False, context=False, requires_grad=False, sample_rate=16000, f_min=0, f_max=None, n_fft=400,
                 n_mels=40, filter_shape="triangular", param_change_factor=1.0, param_rand_factor=0.0, left_frames=5,
                 right_frames=5, win_length=25, hop_length=10, amin=1e-10, top_db=80.0):
        super().__init__()
        if deltas or context or requires_grad or filter_shape != "triangular":
            raise NotImplementedError("only the plain triangular log-mel filterbank of the SummaryMixing recipes is built")
        self.sample_rate, self.n_fft, self.n_mels = sample_rate, n_fft, n_mels
        self.win = int(round(sample_rate / 1000.0 * win_length))
        self.hop = int(round(sample_rate / 1000.0 * hop_length))
        if self.win != n_fft:
            raise NotImplementedError("win_length must equal n_fft samples (recipe: 32 ms @ 16 kHz = 512)")
        self.amin, self.top_db = amin, top_db
        n_bins = n_fft // 2 + 1
        self.im_off = (n_bins + 3) // 4 * 4                                # 260 for 257 bins: 16-byte aligned halves
        k = torch.arange(n_bins, dtype=torch.float64)[:, None] * torch.arange(n_fft, dtype=torch.float64)[None, :]
        ang = 2.0 * math.pi * k / n_fft
        basis = torch.zeros(2 * self.im_off, n_fft, dtype=torch.float64)
        basis[:n_bins] = torch.cos(ang)
        basis[self.im_off:self.im_off + n_bins] = -torch.sin(ang)
        self.register_buffer("basis", basis.float(), persistent=False)
        self.register_buffer("window", torch.hamming_window(self.win, dtype=torch.float32), persistent=False)
        # the analysis window folded into the DFT basis: frames are then plain overlapping rows of the (zero-padded) waveform
        # and the GEMM reads them in place (leading dimension = hop), no (B*T, n_fft) frame matrix is written
        win64 = torch.hamming_window(self.win, dtype=torch.float64)
        self.register_buffer("basis_w", (basis * win64[None, :]).float(), persistent=False)
        # ... and, the window being symmetric (w[j] = w[n - j]), the real DFT in two halves of half the length: the cosine part on
        # x[j] + x[n - j] (j = 0 .. n/2), the sine part on x[j] - x[n - j]; both sums are formed inside the GEMM's operand loader
        half = n_fft // 2
        self.fold = bool(torch.allclose(win64[1:], win64[1:].flip(0))) and n_fft % 8 == 0
        # HTK-mel triangular filters (n_mels, n_bins)
        f_max = sample_rate / 2 if f_max is None else f_max
        to_mel = lambda hz: 2595.0 * math.log10(1.0 + hz / 700.0)
        mel = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2, dtype=torch.float64)
        hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
        band = (hz[1:] - hz[:-1])[:-1]
        freqs = torch.linspace(0, sample_rate // 2, n_bins, dtype=torch.float64)
        slope = (freqs[None, :] - hz[1:-1][:, None]) / band[:, None]
        self.register_buffer("fb", torch.clamp(torch.minimum(slope + 1.0, -slope + 1.0), min=0.0).float().contiguous(),
                             persistent=False)
        # the Nyquist bin carries no weight in any mel filter (the last triangle ends exactly at f_max = sample_rate / 2): without
        # it the bases have n_fft / 2 = 256 rows = exactly two 128-row GEMM tiles instead of three
        fbm = torch.clamp(torch.minimum(slope + 1.0, -slope + 1.0), min=0.0)
        self.drop_nyquist = bool(self.fold and n_bins == half + 1 and half % 128 == 0 and float(fbm[:, -1].abs().max()) == 0.0)
        rows = half if self.drop_nyquist else self.im_off
        bc = torch.zeros(rows, half + 4, dtype=torch.float64)
        bc[:min(n_bins, rows), :half + 1] = (torch.cos(ang[:, :half + 1]) * win64[None, :half + 1])[:rows]
        bs = torch.zeros(rows, half, dtype=torch.float64)
        bs[:min(n_bins, rows)] = (-torch.sin(ang[:, :half]) * win64[None, :half])[:rows]
        self.register_buffer("basis_cos", bc.float().contiguous(), persistent=False)
        self.register_buffer("basis_sin", bs.float().contiguous(), persistent=False)
        self.register_buffer("fb_nn", fbm[:, :half].float().contiguous(), persistent=False)     # filters without the Nyquist column

    def forward(self, wav, out_dtype=torch.float32):
        """wav (B, L) float32 on the GPU -> (B, 1 + L // hop, n_mels)."""
        assert wav.dim() == 2 and wav.dtype == torch.float32 and wav.is_cuda
        B, Lw = wav.shape
        T = 1 + Lw // self.hop
        spec = torch.empty((B * T, self.basis.shape[0]), dtype=torch.float32, device=wav.device)
        if _IMPLICIT_FRAMES and self.hop % 4 == 0:
            # center=True: n_fft/2 zeros on both sides; frame t of utterance b = samples [t*hop, t*hop + n_fft) of the padded row
            half = self.n_fft // 2
            Lp = (Lw + self.n_fft + 4 + 3) // 4 * 4
            wp = torch.empty((B, Lp), dtype=torch.float32, device=wav.device)   # (only the two margins are zero-filled)
            wp[:, :half] = 0.0
            wp[:, half + Lw:] = 0.0
            wp[:, half:half + Lw] = wav
            M = self.basis.shape[0]
            if self.fold and _FOLDED_DFT:
                ops.dft_frames(wp, self.basis_cos, self.basis_sin, spec, self.im_off, B, T, self.n_fft, self.hop)
                if self.drop_nyquist:                           # (column n_fft / 2 of spec was not computed and is not read)
                    return ops.mel_db(spec, self.im_off, self.fb_nn, B, T, self.amin, self.top_db, out_dtype)
            else:
                ops.gemm(L.GEMM_NT, wp[0, :self.n_fft].view(1, -1), self.basis_w, spec[:T], T, M, self.n_fft, batch=B, sa=Lp, sb=0,
                         sc=T * M, lda=self.hop)
        else:
            frames = ops.frame_window(wav.contiguous(), self.window, T, self.n_fft, self.hop)
            ops.gemm(L.GEMM_NT, frames, self.basis, spec, B * T, self.basis.shape[0], self.n_fft)
        return ops.mel_db(spec, self.im_off, self.fb, B, T, self.amin, self.top_db, out_dtype)


class InputNormalization(nn.Module):
    """Mean / variance normalisation of the filterbank features (speechbrain.processing.features.InputNormalization as the
    recipes instantiate it: ``norm_type: global, update_until_epoch: 4``, ...transducer.yaml:167-169).  Statistics per
    utterance over its valid frames (unbiased std, clamped at eps); "sentence" normalises each utterance by its own,
    "batch" by the batch average, "global" by a running average over all training batches seen while
    ``epoch < update_until_epoch`` (weight 1/(count+1), or ``avg_factor``).  Padded frames are normalised too, like the
    reference.  HIP kernels (smx_utt_meanstd / smx_stats_combine / smx_colnorm); arithmetic spec:
    oracle/smx_oracle.py::input_normalization (upstream-only code: parity unpinned)."""

    def __init__(self, mean_norm=True, std_norm=True, norm_type="global", avg_factor=None, requires_grad=False,
                 update_until_epoch=3):
        super().__init__()
        if norm_type not in ("global", "batch", "sentence"):
            raise NotImplementedError("norm_type 'speaker' keeps per-speaker dictionaries on the host: not built")
        self.mean_norm, self.std_norm, self.norm_type = mean_norm, std_norm, norm_type
        self.avg_factor, self.update_until_epoch = avg_factor, update_until_epoch
        self.eps = 1e-10
        self.count = 0
        # running statistics are part of the module state: in state_dict() (buffers, once they exist), moved by .to(),
        # and saved / loaded through SpeechBrain's checkpointer hooks (_save / _load, the normalizer.ckpt format)
        self.register_buffer("glob_mean", None)
        self.register_buffer("glob_std", None)

    # ---- persistence -----------------------------------------------------------------------------------------------
    def get_extra_state(self):
        return {"count": self.count}

    def set_extra_state(self, state):
        self.count = int(state.get("count", 0))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for name in ("glob_mean", "glob_std"):               # buffers that do not exist yet (None) are created from the checkpoint
            key = prefix + name
            if key in state_dict and getattr(self, name) is None:
                setattr(self, name, state_dict[key].detach().clone().float())
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def _statistics_dict(self):
        """The dictionary SpeechBrain's InputNormalization writes to normalizer.ckpt."""
        return {"count": self.count, "glob_mean": self.glob_mean, "glob_std": self.glob_std,
                "spk_dict_mean": {}, "spk_dict_std": {}, "spk_dict_count": {}}

    def _save(self, path):
        torch.save(self._statistics_dict(), path)

    def _load(self, path, end_of_epoch=False):
        del end_of_epoch
        stats = torch.load(path, map_location="cpu")
        dev = self.glob_mean.device if self.glob_mean is not None else None
        self.count = int(stats["count"])
        for name in ("glob_mean", "glob_std"):
            v = stats[name]
            if v is not None and not torch.is_tensor(v):
                v = torch.tensor(v)
            setattr(self, name, v.float().to(dev) if (v is not None and dev is not None) else (v.float() if v is not None else None))

    def forward(self, x, lengths, spk_ids=None, epoch=0):
        if not x.is_cuda:
            raise RuntimeError("summarymixing_amd kernels run on the GPU only (no CPU fallback)")
        B, T, F = x.shape
        x = x if x.is_contiguous() else x.contiguous()
        x2 = x.view(B * T, F)
        lens = torch.round(lengths.to(x.device) * T).to(torch.int32)
        mean = torch.empty((B, F), dtype=torch.float32, device=x.device)
        std = torch.empty((B, F), dtype=torch.float32, device=x.device)
        ops.utt_meanstd(x2, lens, mean, std, B, T, self.mean_norm, self.std_norm, self.eps)
        out = torch.empty_like(x2)
        if self.norm_type == "sentence":
            ops.colnorm(x2, mean, std, F, out, B, T)
            return out.view(B, T, F)
        if self.norm_type == "batch":
            gm, gs = torch.empty(F, device=x.device), torch.empty(F, device=x.device)
            ops.stats_combine(mean, std, gm, gs, 1.0)
        else:
            if self.glob_mean is None:
                self.glob_mean, self.glob_std = torch.zeros(F, device=x.device), torch.ones(F, device=x.device)
            elif self.glob_mean.device != x.device:
                self.glob_mean, self.glob_std = self.glob_mean.to(x.device), self.glob_std.to(x.device)
            if self.training:
                if self.count == 0:
                    ops.stats_combine(mean, std, self.glob_mean, self.glob_std, 1.0)
                elif epoch < self.update_until_epoch:
                    w = 1.0 / (self.count + 1) if self.avg_factor is None else self.avg_factor
                    ops.stats_combine(mean, std, self.glob_mean, self.glob_std, w)
                self.count += 1
            gm, gs = self.glob_mean, self.glob_std
        ops.colnorm(x2, gm, gs, 0, out, B, T)
        return out.view(B, T, F)
