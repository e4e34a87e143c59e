"""CPU oracle for the SummaryMixing encoder hot path.  TEST INFRASTRUCTURE — NOT A PRODUCT PATH.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline.  Nothing under ``summarymixing_amd/`` imports it.

It is a plain-PyTorch-on-CPU *functional* restatement (explicit weights in, tensors out, no nn.Module,
no SpeechBrain) of the reference algorithm.  Every function cites the reference file:line it follows
(paths relative to the reference root, SamsungLabs/SummaryMixing @ 2024_10_08).  Weights are passed as a
flat ``dict[str, Tensor]`` keyed exactly like the reference modules' ``state_dict()`` so a reference
checkpoint feeds the oracle unchanged.  The functions are dtype generic (run them in float64 for a
tight reference) and differentiable (torch autograd provides the gradient oracle).

Pinning: the reference's own known-answer test (tests/unittests/test_summary_mixing.py:60-153) does
not reproduce with the current reference code, so the oracle is pinned against outputs
of the reference itself, generated in the dev container by ``tests/golden/make_golden.py`` (which imports
the unmodified reference files under a SpeechBrain stand-in) and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every fixture.  Cell-level fixtures (G1-G4,G6) depend only on
reference code + ``nn.Linear``; encoder-level fixtures (G5) additionally depend on the stand-in's
restatement of upstream SpeechBrain (LayerNorm / FFN / CSGU) and are flagged "stand-in dependent".
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------
# activations (reference: `activation` ctor arg, summary_mixing.py:85,107; Conformer.py:443)
# ----------------------------------------------------------------------------------------------
def activation(name: str, x: Tensor) -> Tensor:
    """gelu = exact erf GELU (torch.nn.GELU default); swish = x*sigmoid(x) (speechbrain Swish, beta=1);
    leaky_relu slope 0.01 (VanillaNN.py:156 default); relu; identity."""
    if name == "gelu":
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if name == "swish":
        return x * torch.sigmoid(x)
    if name == "leaky_relu":
        return torch.where(x >= 0, x, 0.01 * x)
    if name == "relu":
        return torch.clamp_min(x, 0)
    if name in ("identity", "none"):
        return x
    raise ValueError(f"unknown activation {name}")


# ----------------------------------------------------------------------------------------------
# L1 blocks: ParallelLinear / VanillaNN
# ----------------------------------------------------------------------------------------------
def parallel_linear(x: Tensor, weights: Tensor, biases: Tensor, combine_out_dims: bool) -> Tensor:
    """VanillaNN.py:100-117.  x (B,T,F) or (B,T,H,F/H); weights (H, F/H, h); biases (H, h).
    y[b,t,m,:] = x[b,t,m,:] @ weights[m] + biases[m]; flattened to (B,T,H*h) iff combine_out_dims."""
    H = weights.shape[0]
    if x.ndim == 3:
        B, T, Fdim = x.shape
        x = x.reshape(B, T, H, Fdim // H)
    y = torch.einsum("btmf,mfh->btmh", x, weights) + biases
    if combine_out_dims:
        y = y.reshape(y.shape[0], y.shape[1], -1)
    return y


def _block_names(sd: SD, prefix: str):
    names, i = [], 0
    while True:
        n = "linear" if i == 0 else f"linear_{i - 1}"
        if (prefix + n + ".w.weight") in sd or (prefix + n + ".weights") in sd:
            names.append(n)
            i += 1
        else:
            return names


def vanilla_nn(x: Tensor, sd: SD, prefix: str, act: str) -> Tensor:
    """VanillaNN.py:153-196: [Linear|ParallelLinear -> act] x blocks; an activation follows EVERY block
    (:196); with n_split>1 only the last block recombines the head dim (:177-180).  Block names follow
    the upstream Sequential's duplicate-name rule: linear, linear_0, linear_1, ..."""
    names = _block_names(sd, prefix)
    assert names, f"no VanillaNN blocks under {prefix}"
    for i, n in enumerate(names):
        if (prefix + n + ".weights") in sd:
            x = parallel_linear(x, sd[prefix + n + ".weights"], sd[prefix + n + ".biases"],
                                combine_out_dims=(i == len(names) - 1))
        else:
            x = F.linear(x, sd[prefix + n + ".w.weight"], sd[prefix + n + ".w.bias"])
        x = activation(act, x)
    return x


# ----------------------------------------------------------------------------------------------
# L2: the SummaryMixing cell
# ----------------------------------------------------------------------------------------------
def laplace_weights(T: int, decay: Tensor, binary_mask: Optional[Tensor], dtype) -> Tensor:
    """summary_mixing.py:316-365 with normalise=False: M_ij = exp(|i-j| * log(decay)) * binary_mask_ij."""
    idx = torch.arange(T)
    dist = (idx[None, :] - idx[:, None]).abs().to(dtype)
    m = torch.exp(dist * torch.log(decay.to(dtype)))
    if binary_mask is not None:
        m = m * binary_mask.to(dtype)
    return m


def summary_mixing(x: Tensor, sd: SD, prefix: str, mode: str, act: str, local_proj_out_dim: int,
                   sum_mask: Optional[Tensor] = None, src_padding_mask: Optional[Tensor] = None) -> Tensor:
    """summary_mixing.py:161-310 (forward dispatch + the four modes), dropout = identity (eval / p=0).

    x (B,T,d); src_padding_mask (B,T) bool/float, True/1 = VALID frame (:175-178: None -> all ones);
    sum_mask (T,T) bool/float, row i = frames summarised for step i (:180-181 casts to float).
    """
    B, T, _ = x.shape
    dt = x.dtype
    if src_padding_mask is None:
        m = torch.ones((B, T, 1), dtype=dt)
    else:
        m = src_padding_mask.to(dt).unsqueeze(-1)
    if sum_mask is not None:
        sum_mask = sum_mask.to(dt)

    def pool(s):  # :218-222 / :233-235 / :264-267 / :278-280
        if sum_mask is None:
            sbar = s.sum(dim=1) / m.sum(dim=1)          # denominator = number of valid frames
            return sbar.unsqueeze(1).expand(B, T, sbar.shape[-1])
        # denominator = rowsum(sum_mask): ignores padding (comment :226-231)
        return torch.matmul(sum_mask, s) / sum_mask.sum(dim=1).unsqueeze(-1)

    if mode in ("SummaryMixing", "SummaryMixing-expdecay"):       # :190-239
        local = vanilla_nn(x, sd, prefix + "local_proj.", act) * m
        s = vanilla_nn(x, sd, prefix + "summary_proj.", act) * m
        if mode == "SummaryMixing-expdecay":
            sum_mask = laplace_weights(T, sd[prefix + "decay_constant"], sum_mask, dt)
        return vanilla_nn(torch.cat([local, pool(s)], dim=-1), sd, prefix + "summary_local_merging.", act)
    if mode == "SummaryMixing-fast":                               # :241-284
        g = vanilla_nn(x, sd, prefix + "global_proj.", act) * m
        local, s = g[..., :local_proj_out_dim], g[..., local_proj_out_dim:]
        return vanilla_nn(torch.cat([local, pool(s)], dim=-1), sd, prefix + "summary_local_merging.", act)
    if mode == "SummaryMixing-lite":                               # :286-310 (sum_mask ignored there)
        s = vanilla_nn(x, sd, prefix + "summary_proj.", act) * m
        sbar = s.sum(dim=1) / m.sum(dim=1)
        return sbar.unsqueeze(1).expand(B, T, sbar.shape[-1])
    raise ValueError("The SummaryMixing mode should either be 'SummaryMixing', 'SummaryMixing-lite', "
                     "'SummaryMixing-fast' or 'SummaryMixing-expdecay'")


# ----------------------------------------------------------------------------------------------
# masks (L4)
# ----------------------------------------------------------------------------------------------
def length_to_mask(abs_len: Tensor, max_len: Optional[int] = None) -> Tensor:
    """upstream speechbrain.dataio.dataio.length_to_mask: arange(max_len)[None] < len[:,None]."""
    if max_len is None:
        max_len = int(abs_len.max().item())
    return torch.arange(max_len)[None, :] < abs_len[:, None]


def padding_mask_from_wav_len(wav_len: Tensor, T: int) -> Tensor:
    """TransformerASR.py:157-162 with masked_false_or_true=False: True = valid frame."""
    return length_to_mask(torch.round(wav_len * T)).bool()


def dynchunk_sum_mask(T: int, chunk_size: int, left_context_chunks: Optional[int]) -> Tensor:
    """TransformerASR.py:85-110 with masked_false_or_true=False: (T,T) bool, True = visible.
    Frame t in chunk c sees frames < (c+1)*chunk and, with finite left context L, >= (c-L)*chunk."""
    num_chunks = T // chunk_size
    t = torch.arange(T)
    mask_idx = torch.arange(chunk_size, chunk_size * (num_chunks + 2), chunk_size).repeat_interleave(chunk_size)[:T]
    src_mask = t[None] < mask_idx[:, None]
    if left_context_chunks is not None:
        lo = mask_idx - chunk_size * (left_context_chunks + 1)
        src_mask = src_mask & (t[None] >= lo[:, None])
    return src_mask


def positional_encoding(T: int, d: int, dtype=torch.float32) -> Tensor:
    """Transformer.py:306-335: PE[pos,2i]=sin(pos*w_i), PE[pos,2i+1]=cos(pos*w_i), w_i=exp(-2i*ln(1e4)/d).
    Built in float32 like the reference buffer, then cast."""
    pe = torch.zeros(T, d)
    pos = torch.arange(0, T).unsqueeze(1).float()
    den = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * den)
    pe[:, 1::2] = torch.cos(pos * den)
    return pe.to(dtype)


# ----------------------------------------------------------------------------------------------
# L3: Conformer encoder (stand-in dependent pieces: LayerNorm eps, FFN layout)
# ----------------------------------------------------------------------------------------------
def layer_norm(x: Tensor, sd: SD, wkey: str, bkey: str, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[wkey], sd[bkey], eps)


def depthwise_conv_chunked(u: Tensor, w: Tensor, b: Tensor, chunk_size: int) -> Tensor:
    """Dynamic Chunk Convolution, Conformer.py:190-313, restated tap by tap: the ordinary zero-padded
    depthwise conv, except that for output frame t (in chunk c = t // chunk_size) every input frame at
    or beyond (c+1)*chunk_size -- the future outside t's own chunk -- reads as zero (the reference
    gets this by unfolding into left-context chunks and right-padding each chunk with zeros).
    u (B,T,d); w (d,1,k); b (d)."""
    B, T, d = u.shape
    k = w.shape[-1]
    pad = (k - 1) // 2
    t = torch.arange(T)
    limit = (t // chunk_size + 1) * chunk_size          # exclusive right limit per output frame
    up = F.pad(u, (0, 0, pad, pad))                     # zero pad time
    y = b.to(u.dtype).expand(B, T, d).clone()
    for j in range(k):
        src = t + j - pad
        ok = (src < limit).to(u.dtype)[None, :, None]
        y = y + up[:, j:j + T, :] * ok * w[:, 0, j]
    return y


def conformer_conv_module(x: Tensor, sd: SD, p: str, act: str, mask: Optional[Tensor],
                          chunk_size: Optional[int] = None) -> Tensor:
    """Conformer.py:314-331 (non causal; :190-313 when a DynChunk config is given):
    LN -> pointwise Conv1d(d->2d) -> GLU(channel) -> depthwise Conv1d(k, zero pad (k-1)/2) -> LN -> act
    -> Linear(d,d) -> (dropout) -> * mask (B,T,1) (masked_false_or_true=False for SummaryMixing, :327-331)."""
    d = x.shape[-1]
    out = layer_norm(x, sd, p + "layer_norm.weight", p + "layer_norm.bias")
    out = F.linear(out, sd[p + "bottleneck.0.weight"].squeeze(-1), sd[p + "bottleneck.0.bias"])
    out = out[..., :d] * torch.sigmoid(out[..., d:])
    w = sd[p + "conv.weight"]                       # (d,1,k)
    k = w.shape[-1]
    if chunk_size is not None:
        out = depthwise_conv_chunked(out, w, sd[p + "conv.bias"], chunk_size)
    else:
        out = F.conv1d(out.transpose(1, 2), w, sd[p + "conv.bias"], padding=(k - 1) // 2, groups=d).transpose(1, 2)
    out = layer_norm(out, sd, p + "after_conv.0.weight", p + "after_conv.0.bias")
    out = activation(act, out)
    out = F.linear(out, sd[p + "after_conv.2.weight"], sd[p + "after_conv.2.bias"])
    if mask is not None:
        out = out * mask.to(out.dtype).unsqueeze(-1)
    return out


def ffn_module(x: Tensor, sd: SD, p: str, act: str) -> Tensor:
    """Conformer.py:458-472: LayerNorm -> PositionalwiseFeedForward(Linear d->f, act, Linear f->d)."""
    h = layer_norm(x, sd, p + "0.weight", p + "0.bias")
    h = activation(act, F.linear(h, sd[p + "1.ffn.0.weight"], sd[p + "1.ffn.0.bias"]))
    return F.linear(h, sd[p + "1.ffn.3.weight"], sd[p + "1.ffn.3.bias"])


def conformer_layer(x: Tensor, sd: SD, p: str, act: str, mode: str, local_proj_out_dim: int,
                    src_mask: Optional[Tensor], pad_mask: Optional[Tensor],
                    chunk_size: Optional[int] = None) -> Tensor:
    """Conformer.py:479-537 (attention_type == 'SummaryMixing')."""
    x = x + 0.5 * ffn_module(x, sd, p + "ffn_module1.", act)                       # :507
    skip = x
    h = layer_norm(x, sd, p + "norm1.norm.weight", p + "norm1.norm.bias")          # :510
    h = summary_mixing(h, sd, p + "mha_layer.", mode, act, local_proj_out_dim, src_mask, pad_mask)  # :512-515
    x = h + skip                                                                    # :530
    x = x + conformer_conv_module(x, sd, p + "convolution_module.", act, pad_mask, chunk_size)  # :532-534
    x = x + 0.5 * ffn_module(x, sd, p + "ffn_module2.", act)
    return layer_norm(x, sd, p + "norm2.norm.weight", p + "norm2.norm.bias")       # :536


def _num_layers(sd: SD, p: str) -> int:
    n = 0
    while any(k.startswith(f"{p}layers.{n}.") for k in sd):
        n += 1
    return n


def conformer_encoder(x: Tensor, sd: SD, p: str, act: str, mode: str, local_proj_out_dim: int,
                      src_mask: Optional[Tensor] = None, pad_mask: Optional[Tensor] = None,
                      chunk_size: Optional[int] = None) -> Tensor:
    """Conformer.py:741-786: layer loop + final LayerNorm(eps=1e-6) (:738,784)."""
    for i in range(_num_layers(sd, p)):
        x = conformer_layer(x, sd, f"{p}layers.{i}.", act, mode, local_proj_out_dim, src_mask, pad_mask, chunk_size)
    return layer_norm(x, sd, p + "norm.norm.weight", p + "norm.norm.bias", eps=1e-6)


# ----------------------------------------------------------------------------------------------
# L3: Branchformer encoder (stand-in dependent: CSGU)
# ----------------------------------------------------------------------------------------------
def csgu(x: Tensor, sd: SD, p: str) -> Tensor:
    """upstream ConvolutionalSpatialGatingUnit: split channels; x2 -> LN -> depthwise conv
    (k, 'same', REFLECT pad) -> identity gate act; return x1 * x2."""
    n = x.shape[-1] // 2
    x1, x2 = x[..., :n], x[..., n:]
    x2 = layer_norm(x2, sd, p + "norm.norm.weight", p + "norm.norm.bias")
    w = sd[p + "conv.conv.weight"]
    k = w.shape[-1]
    x2 = F.pad(x2.transpose(1, 2), ((k - 1) // 2, (k - 1) // 2), mode="reflect")
    x2 = F.conv1d(x2, w, sd[p + "conv.conv.bias"], groups=n).transpose(1, 2)
    return x1 * x2


def branchformer_layer(x: Tensor, sd: SD, p: str, act: str, mode: str, local_proj_out_dim: int,
                       src_mask: Optional[Tensor], pad_mask: Optional[Tensor]) -> Tensor:
    """Branchformer.py:243-334: x + merge_proj(cat[SM(LN(x)), cgMLP(LN(x))]) (dropout = identity)."""
    x1 = layer_norm(x, sd, p + "norm_mhsa.norm.weight", p + "norm_mhsa.norm.bias")
    x1 = summary_mixing(x1, sd, p + "mha_layer.", mode, act, local_proj_out_dim, src_mask, pad_mask)
    x2 = layer_norm(x, sd, p + "norm_conv.norm.weight", p + "norm_conv.norm.bias")
    cb = p + "convolution_branch."
    x2 = activation(act, F.linear(x2, sd[cb + "pre_channel_proj.weight"], sd[cb + "pre_channel_proj.bias"]))
    x2 = csgu(x2, sd, cb + "csgu.")
    x2 = F.linear(x2, sd[cb + "post_channel_proj.weight"], sd[cb + "post_channel_proj.bias"])
    return x + vanilla_nn(torch.cat([x1, x2], dim=-1), sd, p + "merge_proj.", act)


def branchformer_encoder(x: Tensor, sd: SD, p: str, act: str, mode: str, local_proj_out_dim: int,
                         src_mask: Optional[Tensor] = None, pad_mask: Optional[Tensor] = None) -> Tensor:
    """Branchformer.py:447-491: layer loop + final LayerNorm(eps=1e-6) (:444,489)."""
    for i in range(_num_layers(sd, p)):
        x = branchformer_layer(x, sd, f"{p}layers.{i}.", act, mode, local_proj_out_dim, src_mask, pad_mask)
    return layer_norm(x, sd, p + "norm.norm.weight", p + "norm.norm.bias", eps=1e-6)


# ----------------------------------------------------------------------------------------------
# L4: TransformerASR.encode / EncoderWrapper.forward
# ----------------------------------------------------------------------------------------------
def asr_encode(src: Tensor, wav_len: Optional[Tensor], sd: SD, encoder_module: str, act: str, mode: str,
               local_proj_out_dim: int, dynchunk: Optional[tuple] = None) -> Tensor:
    """TransformerASR.py:501-560 via EncoderWrapper.forward (:720-729), attention_type='SummaryMixing':
    4-D -> 3-D reshape (:528-530); key-padding mask True=valid from round(wav_len*T) (:157-162);
    custom_src_module = Linear(input_size->d) (+dropout) (:349-354,542); src += abs-sine PE (:547-549);
    encoder stack.  dynchunk = (chunk_size, left_context_chunks|None)."""
    if src.ndim == 4:
        src = src.reshape(src.shape[0], src.shape[1], -1)
    T = src.shape[1]
    pad_mask = padding_mask_from_wav_len(wav_len, T) if wav_len is not None else None
    src_mask = dynchunk_sum_mask(T, dynchunk[0], dynchunk[1]) if dynchunk is not None else None
    x = F.linear(src, sd["custom_src_module.layers.0.w.weight"], sd["custom_src_module.layers.0.w.bias"])
    x = x + positional_encoding(T, x.shape[-1], x.dtype)
    if encoder_module == "conformer":
        return conformer_encoder(x, sd, "encoder.", act, mode, local_proj_out_dim, src_mask, pad_mask,
                                 dynchunk[0] if dynchunk is not None else None)
    return branchformer_encoder(x, sd, "encoder.", act, mode, local_proj_out_dim, src_mask, pad_mask)


# ----------------------------------------------------------------------------------------------
# Front-end.  PARITY UNPINNED: this arithmetic lives in un-vendored SpeechBrain
# (speechbrain.lobes.features.Fbank, speechbrain.lobes.models.convolution.ConvolutionFrontEnd; call sites
# recipes/LibriSpeech/ASR/transducer/hparams/conformer_summarymixing_transducer.yaml:167-175,247-254) and no
# reference test touches it.  The functions below are this repo's own CPU definition (SURVEY §2.1 formulas).
# ----------------------------------------------------------------------------------------------
def mel_filterbank(n_mels: int = 80, n_fft: int = 512, sample_rate: int = 16000, f_min: float = 0.0,
                   f_max: Optional[float] = None) -> Tensor:
    """Triangular HTK-mel filters (n_mels, n_fft//2+1): mel = 2595 log10(1 + f/700)."""
    f_max = sample_rate / 2 if f_max is None else f_max
    to_mel = lambda hz: 2595.0 * math.log10(1.0 + hz / 700.0)
    mel = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2, dtype=torch.float64)
    hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
    band = hz[1:] - hz[:-1]
    f_central, band = hz[1:-1], band[:-1]
    all_freqs = torch.linspace(0, sample_rate // 2, n_fft // 2 + 1, dtype=torch.float64)
    slope = (all_freqs[None, :] - f_central[:, None]) / band[:, None]
    return torch.clamp(torch.minimum(slope + 1.0, -slope + 1.0), min=0.0).float()


def fbank(wav: Tensor, sample_rate: int = 16000, n_fft: int = 512, win_length_ms: float = 32, hop_length_ms: float = 10,
          n_mels: int = 80, amin: float = 1e-10, top_db: float = 80.0) -> Tensor:
    """wav (B, L) -> log-mel (B, T, n_mels), T = 1 + L // hop: STFT(hamming, center, zero pad) -> |X|^2 -> mel ->
    10 log10(max(., amin)) -> clamp at (per-utterance max - top_db)."""
    win = int(round(sample_rate / 1000.0 * win_length_ms))
    hop = int(round(sample_rate / 1000.0 * hop_length_ms))
    window = torch.hamming_window(win, dtype=wav.dtype)
    X = torch.stft(wav, n_fft, hop, win, window, center=True, pad_mode="constant", normalized=False, onesided=True,
                   return_complex=True)
    power = (X.real ** 2 + X.imag ** 2).transpose(1, 2)                 # (B, T, n_bins)
    mel = power @ mel_filterbank(n_mels, n_fft, sample_rate).to(wav.dtype).t()
    db = 10.0 * torch.log10(torch.clamp(mel, min=amin))
    return torch.maximum(db, db.amax(dim=(-2, -1), keepdim=True) - top_db)


def conv_frontend(x: Tensor, sd: SD, prefix: str = "") -> Tensor:
    """x (B, T, F) -> (B, ceil(T/4), ceil(F/4) * C_last): per block Conv2d(3x3, stride 2, reflect pad 1) over (time, freq)
    -> LayerNorm over (F', C) -> LeakyReLU(0.01).  Weights: {prefix}convblock_i.conv.weight (Cout, Cin, 3, 3), .bias,
    {prefix}convblock_i.norm.weight/.bias (F', Cout)."""
    h = x.unsqueeze(1)                                                   # (B, C=1, T, F)
    i = 0
    while f"{prefix}convblock_{i}.conv.weight" in sd:
        p = f"{prefix}convblock_{i}."
        h = F.conv2d(F.pad(h, (1, 1, 1, 1), mode="reflect"), sd[p + "conv.weight"], sd[p + "conv.bias"], stride=2)
        hl = h.permute(0, 2, 3, 1)                                       # (B, T', F', C)
        hl = F.layer_norm(hl, hl.shape[2:], sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-5)
        h = F.leaky_relu(hl, 0.01).permute(0, 3, 1, 2)
        i += 1
    h = h.permute(0, 2, 3, 1)
    return h.reshape(h.shape[0], h.shape[1], -1)


# ----------------------------------------------------------------------------------------------------
# CTC head.  Arithmetic is upstream-only (SpeechBrain v1.0 `speechbrain.nnet.losses.ctc_loss`
# -> torch.nn.functional.ctc_loss, and `speechbrain.nnet.activations.Softmax(apply_log=True)`; call sites
# …/LibriSpeech/ASR/transducer/hparams/conformer_summarymixing_transducer.yaml:297-298,331): **parity unpinned** by the
# reference tree, pinned here against torch's own CPU implementation, which is what the recipes execute.
# ----------------------------------------------------------------------------------------------------
def log_softmax(x):
    return torch.log_softmax(x, dim=-1)


def ctc_loss(log_probs, targets, input_lens, target_lens, blank_index, reduction="mean"):
    """speechbrain.nnet.losses.ctc_loss restated: relative lengths -> absolute, (B,T,V) -> (T,B,V), zero_infinity."""
    B, T, _ = log_probs.shape
    in_len = (input_lens * T).round().int()
    tgt_len = (target_lens * targets.shape[1]).round().int()
    red = {"batchmean": "sum", "batch": "none"}.get(reduction, reduction)
    loss = torch.nn.functional.ctc_loss(log_probs.transpose(0, 1), targets, in_len, tgt_len, blank_index,
                                        zero_infinity=True, reduction=red)
    if reduction == "batchmean":
        return loss / B
    if reduction == "batch":
        return loss / tgt_len.to(loss.dtype)
    return loss


# ----------------------------------------------------------------------------------------------------
# InputNormalization (speechbrain.processing.features.InputNormalization v1.0, recipe key `normalize`,
# …transducer.yaml:167-169).  Upstream-only arithmetic: **parity unpinned**; restated from the SpeechBrain semantics:
# per-utterance statistics over the valid frames (torch.mean / unbiased torch.std, std clamped at eps=1e-10), the
# "global" mode keeps a running average of the per-batch averages while training and epoch < update_until_epoch.
# ----------------------------------------------------------------------------------------------------
class InputNormalizationState:
    def __init__(self):
        self.count, self.glob_mean, self.glob_std = 0, None, None


def input_normalization(x, lengths, state, norm_type="global", mean_norm=True, std_norm=True, avg_factor=None,
                        update_until_epoch=3, epoch=0, training=True, eps=1e-10):
    B, T, _ = x.shape
    means, stds = [], []
    for b in range(B):
        n = int(torch.round(lengths[b] * T))
        seg = x[b, :n]
        m = seg.mean(0) if mean_norm else torch.zeros(1, dtype=x.dtype)
        s = seg.std(0) if std_norm else torch.ones(1, dtype=x.dtype)
        means.append(m)
        stds.append(torch.max(s, eps * torch.ones_like(s)))
    if norm_type == "sentence":
        return torch.stack([(x[b] - means[b]) / stds[b] for b in range(B)])
    cm, cs = torch.stack(means).mean(0), torch.stack(stds).mean(0)
    if norm_type == "batch":
        return (x - cm) / cs
    if training:
        if state.count == 0:
            state.glob_mean, state.glob_std = cm, cs
        elif epoch < update_until_epoch:
            w = 1.0 / (state.count + 1) if avg_factor is None else avg_factor
            state.glob_mean = (1 - w) * state.glob_mean + w * cm
            state.glob_std = (1 - w) * state.glob_std + w * cs
        state.count += 1
    return (x - state.glob_mean) / state.glob_std
