"""Sequence-parallel (long-context) mode: one utterance batch, the TIME axis sharded over the ranks of a group.

SURVEY §8(e) "optional sequence sharding" / §8(f) rank 4.  The encoder is per-frame everywhere except in two places,
and those are the only exchange steps (one process per GPU; RCCL when the group's backend is "nccl"):

  * the cell's per-utterance mean (summary_mixing.py:218-222, 264-267): every rank sums its own frames, ONE all-reduce of
    a (B, D_s + 1) fp32 buffer [partial sums | valid-frame counts] gives the global mean.  Backward: one all-reduce of
    the (B, D_s) gradient of the mean.
  * the depthwise convolution of the Conformer conv module (k = 31, Conformer.py:126): (k-1)/2 halo frames from the left
    and right neighbour in the forward, the halo frames' input gradients back to their owners in the backward.  The halo
    is taken on the module INPUT, the per-frame LN / pointwise conv / GLU are recomputed on the 2 x 15 halo frames
    (their parameter-gradient contributions are linear in the frame gradients, so per-rank partial sums stay exact).

Rank r of the group holds frames [r*T_loc, (r+1)*T_loc) of every utterance (equal T_loc >= (k-1)/2 on every rank; pad the
tail and mark it in the padding mask).  Parameter gradients come out as per-rank PARTIAL SUMS over the rank's frames:
reduce them with reduce_gradients() (SUM over the group) before the optimizer.

  * (round 4) the CSGU's depthwise convolution of the Branchformer's cgMLP branch (k = 31, REFLECT padding, Branchformer.py:31-97):
    the halo is taken on the conv INPUT LN(x2) itself (no per-frame recomputation), and at the two ends of the WHOLE sequence
    the halo rows are filled with the reflected frames, so a zero-padded conv over the extended shard equals the reflect-padded
    conv over the whole sequence; backward: halo gradients back to their owners, reflected rows' gradients folded onto their
    source frames.

  * (round 4) SummaryMixing-expdecay without sum_mask (summary_mixing.py:316-365): the two-sided exponential filter crosses a
    shard boundary through ONE (B, D) state per direction - an all-gather of (2, B, D) per rank, rank-1 corrections
    decay^(t+1) f_in + decay^(T-t) g_in on the local numerators, denominators from the global frame index
    (functional._expdecay_seqpar; forward and the backward's transposed operator).

  * (round 4) Dynamic Chunk Training (Conformer.py:190-313, TransformerASR.py:85-110; shards of whole chunks, T_loc % chunk == 0):
    the chunked convolution never reads beyond a frame's own chunk, so it needs no right halo and a LEFT halo rounded up to whole
    chunks (the chunk grid of the extended shard is the global one); the chunk-window mean crosses a shard boundary through chunk
    SUMS only - the last `left` chunk sums of the previous shard, or the totals of all earlier shards with unlimited left
    context (functional._chunk_mean_seqpar; the transposed operator sends the same sums the other way).

Supported: ConformerEncoder(Layer) and BranchformerEncoder(Layer) with the per-utterance mean (modes SummaryMixing, -fast,
-lite), the mask-free expdecay summary or a DynChunk mask (Conformer), with or without dropout (per-rank seeds: round 6).  A dense
(T, T) sum_mask raises NotImplementedError.  Host-side plumbing only: the
arithmetic stays in libsmx.
"""
import contextlib

import torch
import torch.distributed as dist


class _State:
    group = None
    active = False
    rank = 0
    world = 1


def enabled():
    return _State.active and _State.world > 1


def rank():
    return _State.rank


def world():
    return _State.world


@contextlib.contextmanager
def sequence_parallel(group=None):
    """with sequence_parallel(group): out, _ = encoder(x_local, src_key_padding_mask=mask_local); loss.backward()"""
    if not dist.is_initialized():
        raise RuntimeError("sequence_parallel needs an initialised torch.distributed process group")
    from . import ops
    prev = (_State.group, _State.active, _State.rank, _State.world)
    _State.group, _State.active = group, True
    _State.rank, _State.world = dist.get_rank(group), dist.get_world_size(group)
    # dropout (round 6): the fused masks are functions of (seed, LOCAL frame row, column); every shard salts the seeds it draws with
    # its rank, so the shards' masks are independent - forward and backward of a site share the drawn seed as everywhere else
    prev_salt = ops.set_seed_salt(_State.rank + 1)
    try:
        yield
    finally:
        ops.set_seed_salt(prev_salt)
        _State.group, _State.active, _State.rank, _State.world = prev


def _host_staged():
    # gloo moves device tensors through the host anyway; doing it here keeps the path independent of how the
    # torch build configured gloo (the CPU-only / single-GPU test rigs use it; production uses "nccl" = RCCL)
    return dist.get_backend(_State.group) == "gloo"


def all_reduce_sum(t):
    """In-place SUM of a small fp32 tensor over the sequence group."""
    if not enabled():
        return t
    if _host_staged() and t.is_cuda:
        h = t.detach().cpu()
        dist.all_reduce(h, group=_State.group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=_State.group)
    return t


def all_gather(t):
    """Every rank's copy of a small tensor (list indexed by rank; host-staged under gloo)."""
    return _all_gather(t)


def _all_gather(t):
    w = _State.world
    if _host_staged() and t.is_cuda:
        h = t.detach().cpu().contiguous()
        outs = [torch.empty_like(h) for _ in range(w)]
        dist.all_gather(outs, h, group=_State.group)
        return [o.to(t.device) for o in outs]
    t = t.contiguous()
    outs = [torch.empty_like(t) for _ in range(w)]
    dist.all_gather(outs, t, group=_State.group)
    return outs


def exchange_halos(first, last):
    """first / last: this rank's first and last h frames, (B, h, d).  Returns (left_halo, right_halo): the last h frames
    of rank-1 and the first h frames of rank+1 (zeros at the two ends of the sequence).  One all-gather of (2, B, h, d):
    the edges are a few hundred KB, the ring neighbours are two of the (at most 8) ranks."""
    r, w = _State.rank, _State.world
    edges = _all_gather(torch.stack([first, last]))
    left = edges[r - 1][1] if r > 0 else torch.zeros_like(first)
    right = edges[r + 1][0] if r < w - 1 else torch.zeros_like(last)
    return left, right


def return_halo_grads(g_left_halo, g_right_halo):
    """The transposed exchange: this rank's gradients w.r.t. its LEFT halo (frames owned by rank-1, its last h) and RIGHT
    halo (owned by rank+1, its first h).  Returns (g_first, g_last): what the neighbours computed for this rank's own
    first / last h frames (zeros at the ends)."""
    r, w = _State.rank, _State.world
    g = _all_gather(torch.stack([g_left_halo, g_right_halo]))
    g_first = g[r - 1][1] if r > 0 else torch.zeros_like(g_left_halo)      # rank-1's right-halo gradient
    g_last = g[r + 1][0] if r < w - 1 else torch.zeros_like(g_right_halo)  # rank+1's left-halo gradient
    return g_first, g_last


def shard(x, dim=1):
    """This rank's contiguous slice of a full-length tensor along the time axis (length divisible by the group size)."""
    w, r = _State.world, _State.rank
    T = x.shape[dim]
    if T % w != 0:
        raise ValueError(f"time axis {T} is not divisible by the sequence group size {w}: pad the tail")
    return x.narrow(dim, r * (T // w), T // w).contiguous()


def reduce_gradients(params):
    """SUM the per-rank partial parameter gradients over the sequence group (flat fp32 bucket, one all-reduce)."""
    if not enabled():
        return
    gs = [p.grad for p in params if p.grad is not None]
    if not gs:
        return
    flat = torch.cat([g.reshape(-1).float() for g in gs])
    all_reduce_sum(flat)
    o = 0
    for g in gs:
        n = g.numel()
        g.copy_(flat[o:o + n].view_as(g))
        o += n
