"""CTC loss of the recipes' multitask head (recipe key ``ctc_cost``: speechbrain.nnet.losses.ctc_loss, which wraps
``torch.nn.functional.ctc_loss(..., zero_infinity=True)``; SpeechBrain is not vendored in the reference tree, the call
site is …/LibriSpeech/ASR/transducer/hparams/conformer_summarymixing_transducer.yaml:297-298).  The forward/backward
variables and the gradient run in the HIP kernels of csrc/ctc.hip; no CPU fallback."""
import torch

from .. import ops


class _CTC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_probs, targets, in_len, tgt_len, blank):
        B, T, V = log_probs.shape
        lp2 = ops.rows2d(log_probs if log_probs.is_contiguous() else log_probs.contiguous())
        nll, ws = ops.ctc_fwd(lp2, targets, in_len, tgt_len, B, T, blank)
        ctx.save_for_backward(lp2, targets, in_len, tgt_len, nll, ws)
        ctx.meta = (B, T, V, blank)
        return torch.where(torch.isinf(nll), torch.zeros_like(nll), nll)      # zero_infinity=True

    @staticmethod
    def backward(ctx, gnll):
        lp2, targets, in_len, tgt_len, nll, ws = ctx.saved_tensors
        B, T, V, blank = ctx.meta
        g = ops.ctc_bwd(lp2, targets, in_len, tgt_len, B, T, blank, nll, gnll.float().contiguous(), ws)
        return g.view(B, T, V), None, None, None, None


def ctc_loss(log_probs, targets, input_lens, target_lens, blank_index, reduction="mean"):
    """speechbrain.nnet.losses.ctc_loss.  log_probs (B, T, V) log-softmax outputs (CUDA, fp32 or bf16); targets
    (B, S) integer tokens (padded); input_lens / target_lens RELATIVE lengths in (0, 1] as everywhere in SpeechBrain."""
    if not log_probs.is_cuda:
        raise RuntimeError("summarymixing_amd.nnet.losses.ctc_loss runs on the GPU only (no CPU fallback)")
    B, T, V = log_probs.shape
    in_len = (input_lens.to(log_probs.device) * T).round().to(torch.int32)
    tgt_len = (target_lens.to(log_probs.device) * targets.shape[1]).round().to(torch.int32)
    tg = targets.to(device=log_probs.device, dtype=torch.int32).contiguous()
    nll = _CTC.apply(log_probs, tg, in_len, tgt_len, int(blank_index))
    if reduction == "mean":                          # torch: each loss / target length (>= 1), then the batch mean
        return (nll / tgt_len.clamp(min=1).to(nll.dtype)).mean()
    if reduction == "sum":
        return nll.sum()
    if reduction == "batchmean":
        return nll.sum() / B
    if reduction == "batch":
        return nll / tgt_len.to(nll.dtype)           # SpeechBrain: per-utterance loss over its target length
    if reduction == "none":
        return nll
    raise ValueError(f"unknown reduction {reduction!r}")
