"""Host-side orchestration of the SummaryMixing encoder hot path on top of the libsmx.so kernels.

Every block is a (forward, backward-closure) pair written against ``ops``; a single generic
``torch.autograd.Function`` (``_BlockFn``) chains the blocks, so autograd only ever sees one activation in and
one activation out per block: residual adds, gradient fan-in sums and parameter-gradient accumulation all
happen inside the HIP kernels (GEMM epilogues, LayerNorm-backward residual input, split-K slabs / partial rows
folded into ``param.grad`` by a fixed-order reduction - no atomics).  Parameter gradients are therefore written by
the kernels directly into ``param.grad`` (created on demand, fp32, accumulate semantics identical to autograd's);
autograd itself never sees them: AccumulateGrad hooks do not fire, so torch DistributedDataParallel must NOT wrap
these modules (trainer.FlatAdamW buckets, or an explicit all-reduce of the ``.grad`` tensors, do the reduction).

Reference behaviour mirrored here (paths relative to the reference root):
  cell                  speechbrain/nnet/summary_mixing.py:161-310
  VanillaNN/ParallelLinear  speechbrain/lobes/models/VanillaNN.py:26-196
  Conformer layer       speechbrain/lobes/models/transformer/Conformer.py:314-331,479-537
  Branchformer layer    speechbrain/lobes/models/transformer/Branchformer.py:31-97,243-334
"""
import weakref

import ctypes
import os

import torch

from . import _lib as L
from . import ops
from . import sequence_parallel as SP

# ----------------------------------------------------------------------------------------------------
# parameter plumbing: compute-dtype shadows of fp32 master weights, gradient buffers
# ----------------------------------------------------------------------------------------------------
_shadow = {}   # id(param) -> (weakref(param), version | "managed", data_ptr, shadow tensor)


def register_shadow(param, shadow):
    """Trainer hook: `shadow` (bf16 view of a flat buffer) is kept up to date by smx_adamw_step.  A parameter that changes hands
    (a second FlatAdamW over the same module) gets new shadow / flat buffers: its packed images point at the old ones - drop them."""
    _shadow[id(param)] = (weakref.ref(param), "managed", None, shadow)
    for key in [k for k in _packed if k[0] == id(param)]:
        del _packed[key]
    weights_changed()


def wcast(param, dtype):
    """fp32 master parameter -> tensor in the compute dtype (bf16 shadows are cached per parameter version)."""
    if dtype == torch.float32:
        return param.detach()
    ent = _shadow.get(id(param))
    if ent is not None and ent[0]() is param and (
            ent[1] == "managed" or (ent[1] == param._version and ent[2] == param.data_ptr())):
        return ent[3]
    sh = ops.cast(param.detach(), dtype)
    if len(_shadow) > 4096:                      # drop entries of dead parameters
        for k in [k for k, v in _shadow.items() if v[0]() is None]:
            del _shadow[k]
    _shadow[id(param)] = (weakref.ref(param), param._version, param.data_ptr(), sh)
    return sh


# ---- packed weight images of the panel-resident GEMM (ops.gemm_panel / smx_weight_pack): per (parameter, orientation), re-packed
# when the weight changed.  Unmanaged parameters: torch's version counter (+ the bias's).  Trainer-managed bf16 shadows are rewritten
# behind torch's back by smx_adamw_step, so the trainer bumps an epoch (weights_changed) after every update / checkpoint load.
# The image is re-packed INTO THE SAME BUFFER: its address stays valid for captured hipGraphs.
# hipGraph rules (round-5 advisor findings): a managed image's stamp carries the id of the capture it was packed in (0 = eager), so
#  * EVERY capture contains its own pack launches, whatever the epoch was when it began (a second capture, or one taken right after
#    a forward-only warm-up, used to contain none and its replays trained against frozen images);
#  * the first eager use after a capture re-packs.  Replays advance the shadows without running this Python code: whoever replays a
#    graph that contains an optimizer update calls weights_changed() before the next EAGER forward (FlatAdamW.replay does).
_WEPOCH = [0]
# (id(param), transposed) -> (weakref(param), stamp, packed image, W (compute-dtype view), bias tensor | None, M, K, transposed)
_packed = {}


def weights_changed():
    """Trainer hook: the managed bf16 shadows / fp32 biases were rewritten in place (optimizer step, checkpoint load, graph replay)."""
    _WEPOCH[0] += 1


# device job tables of the trainer-managed packed images, per device and per signature.  A table whose launch was captured stays
# alive for the life of the process (the graph holds its raw pointer); the others are dropped when the set of images changes.
_pack_tables = {}   # (device, signature) -> [device table, blocks, captured?]


def _purge_packed():
    for k in [k for k, v in _packed.items() if v[0]() is None]:
        del _packed[k]


def _pack_jobs_of(e):
    """(W ptr, ldw, bias ptr, image ptr, M, K, transposed) per pack job of a cache entry: one, or one per K-slice (entry[8] slices of
    K = entry[6] reduce elements each: column ranges of a forward weight, row ranges of the same weight seen from its dgrad)."""
    W, img, M, K, tr = e[3], e[2], e[5], e[6], e[7]
    ns = e[8] if len(e) > 8 else 1
    if ns == 1:
        return [(W.data_ptr(), W.stride(0), 0 if e[4] is None else e[4].data_ptr(), img.data_ptr(), M, K, tr)]
    per = L.lib().smx_weight_pack_bytes(M, K)
    step = K * W.stride(0) * 2 if tr else K * 2
    return [(W.data_ptr() + s_ * step, W.stride(0), 0, img.data_ptr() + s_ * per, M, K, tr) for s_ in range(ns)]


def _repack_managed(device, stamp):
    """ONE launch re-packs every trainer-managed image of `device` (they all went stale together: the optimizer rewrote every shadow)."""
    _purge_packed()
    keys = [k for k, v in _packed.items() if v[1][0] == "m" and v[2].device == device]
    ents = [_packed[k] for k in keys]
    sig = tuple(j for e in ents for j in _pack_jobs_of(e))
    capturing = torch.cuda.is_current_stream_capturing()
    tab = _pack_tables.get((device, sig))
    if tab is None:
        if capturing:
            return False                                 # (no host-to-device table upload inside a capture: the caller packs singly)
        for k in [k for k, t in _pack_tables.items() if k[0] == device and not t[2]]:
            del _pack_tables[k]
        lib, arr, nb = L.lib(), (L.PackJob * len(sig))(), 0
        for j, (W_, ldw, bias_, out_, M, K, tr) in zip(arr, sig):
            j.W, j.ldw, j.bias, j.packed, j.M, j.K, j.transposed, j.block_start = W_, ldw, bias_ or None, out_, M, K, tr, nb
            nb += lib.smx_weight_pack_job_blocks(M, K)
        tab = _pack_tables[(device, sig)] = [torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device), nb, False]
    tab[2] = tab[2] or capturing
    ops.weight_pack_jobs(tab[0], len(sig), tab[1], sum(j[4] * j[5] for j in sig))
    for k, v in zip(keys, ents):
        _packed[k] = (v[0], stamp) + v[2:]
    return True


def wpacked(param, dtype, transposed=False, bias=None, kslice=0):
    """Packed image (ops.weight_pack) of a Linear's weight parameter [+ fp32 bias parameter] in the compute dtype.
    kslice > 0: the image of gemm_panel_slabs instead - the weight's K-slices of `kslice` reduce elements packed one after the other,
    no bias (ops.weight_pack_slices)."""
    W = wcast(param, dtype)
    if W.dim() != 2:
        W = W.view(W.shape[0], -1)               # (a Conv1d(k = 1) weight (out, in, 1) seen as a Linear's)
    assert kslice == 0 or bias is None
    ent = _shadow.get(id(param))
    managed = ent is not None and ent[0]() is param and ent[1] == "managed"
    bt = None if bias is None else bias.detach()
    if managed:
        stamp = ("m", _WEPOCH[0], ops.capture_id())
    else:
        stamp = ("v", param._version, param.data_ptr(), None if bias is None else (bias._version, bias.data_ptr()))
    key = (id(param), transposed) if kslice == 0 else (id(param), transposed, kslice)
    c = _packed.get(key)
    if c is not None and c[0]() is not param:
        del _packed[key]                           # (a dead parameter's id re-used)
        c = None
    # the image was packed FROM these tensors: another shadow view (a new optimizer), another bias or none (tied weights) -> pack anew
    same_src = c is not None and c[3].data_ptr() == W.data_ptr() and c[3].stride() == W.stride() and \
        (None if c[4] is None else c[4].data_ptr()) == (None if bt is None else bt.data_ptr())
    if same_src and c[1] == stamp:
        return c[2]
    if same_src and managed and c[1][0] == "m" and _repack_managed(c[2].device, stamp):
        return c[2]                                # (stale with every other managed image: one grouped launch re-packed them all)
    if len(_packed) > 256:
        _purge_packed()
    out = c[2] if c is not None else None          # (same parameter and orientation = same shape: the buffer is re-used)
    M, K = (W.shape[1], W.shape[0]) if transposed else (W.shape[0], W.shape[1])
    if kslice:
        out = ops.weight_pack_slices(W, kslice, transposed, out)
        _packed[key] = (weakref.ref(param), stamp, out, W, None, M, kslice, int(transposed), K // kslice)
    else:
        out = ops.weight_pack(W, transposed, bt, out)
        _packed[key] = (weakref.ref(param), stamp, out, W, bt, M, K, int(transposed))
    return out


# The panel-resident GEMM takes the output-bound Linears with a short reduction (K = 256 / 512: FFN up-projection, the act-grad
# dgrad of the down-projection, the conv module's pointwise Linear, the cgMLP projections) from this many rows on.  One 128-row
# panel per CU; below ~190 panels the kernel deals a panel's chunk rounds to 2 or 4 workgroups.  Same-box A/B of the steps
# (tools/experiments/ab_panel_rows.sh): 16 000 rows C2b 8.17 -> 7.83 ms, C2a 16.42 -> 15.81; 8 000 rows a tie; 32 000 rows -3 %.
_PANEL = os.environ.get("SMX_PANEL", "1") != "0"
_PANEL_MIN_ROWS = int(os.environ.get("SMX_PANEL_MIN_ROWS", "2048"))   # (round 6: 64- / 32-row panels below 12 288 rows, chosen by the library)
_PANEL_ACTS = (L.ACT_NONE, L.ACT_SWISH, L.ACT_GELU, L.ACT_RELU)


# Split-K over workgroups for the Linears of a SMALL batch (smx_gemm_panel_slabs + smx_slab_epilogue / smx_layernorm_bwd2_slabs; round 6):
# the reduction is cut into panel-sized K-slices, the slabs are summed by the row kernel that follows the Linear anyway (its epilogue +
# the LayerNorm, or the LayerNorm backward behind a dgrad).  Measured (tools/experiments/r06_smalln/splitk_bench.py, us, GEMM + LayerNorm
# pair, tiled | split-K): d_model 256 at 500 frames K = 1024: 13.2 | 10.2, 512: 10.8 | 9.3, 256: 9.5 | 8.8; 2000 frames: 14.1 | 13.8, 11.6 | 11.2;
# d_model 512 at 500 frames: 18.4 | 14.3, 13.3 | 12.6; 2000 frames 21.9 | 20.8; at 3750 frames the slab traffic (S x N x M x 4 bytes
# written and read back) eats the gain (27.5 | 26.1 at K = 2048, a loss below) - hence up to 2048 frames only.
_SPLITK = os.environ.get("SMX_SPLITK", "1") != "0"
_SPLITK_MAX_ROWS = int(os.environ.get("SMX_SPLITK_MAX_ROWS", "2048"))


def splitk_cfg(N, M, K, dtype):
    """(K-slice, number of slices) when the Linear (N x K) -> (N x M) should run as split-K slabs, else None."""
    if not (_SPLITK and _PANEL and dtype == torch.bfloat16 and 64 <= N <= _SPLITK_MAX_ROWS and M % 64 == 0 and 64 <= M <= 512 and not SP.enabled()):
        return None
    ks = 256 if (M <= 256 or K % 512 != 0) else 512
    if K % ks != 0 or K // ks > 16 or (K // ks == 1 and N > 1024):
        return None
    return (ks, K // ks) if L.lib().smx_gemm_panel_slabs_ok(L.BF16, N, M, ks, K // ks) == 1 else None


def _span_ok(*ts):
    """Every 2-D view spans less than 2 GB WITH its leading dimension (the panel kernel addresses with 32-bit buffer offsets)."""
    return all(t is None or ((t.shape[0] - 1) * t.stride(0) + t.shape[1]) * t.element_size() < (1 << 31) for t in ts)


def panel_ok(x, M, K, act):
    return (_PANEL and x.dtype == torch.bfloat16 and act in _PANEL_ACTS and x.shape[0] >= _PANEL_MIN_ROWS and _vec_ok(x) and
            _span_ok(x) and ops.gemm_panel_ok(x, M, K))


_POOL_FUSE = os.environ.get("SMX_POOL_FUSE", "1") != "0"   # small batches: smx_pool_bcast instead of masked mean + broadcast (3 launches)
_WGRAD_BIAS = True   # (round 4: the SMX_NO_WGRAD_BIAS A/B knob is gone)     # A/B knob: bias gradients as a by-product of the wgrad GEMM

# Residual-stream dtype of a bf16 model.  "fp32" (default) = torch autocast semantics, what the reference's `precision: bf16`
# means (…transducer.yaml:61): Linear inputs / outputs in bf16, the residual adds (Conformer.py:507,530,532-536) and the
# LayerNorm inputs in float32 - the stream tensors x, x + 1/2 FFN, + cell, + conv, and the layer-final LayerNorm output are
# stored in float32, everything a GEMM reads or writes stays bf16.  "bf16" (SMX_RESIDUAL=bf16, the round-1/2 behaviour) stores
# the stream in bf16 too: ~7 % faster, 2e-2 instead of < 1e-2 max-rel on the 12-layer forward.  Gradients are bf16 in both.
RESIDUAL_F32 = os.environ.get("SMX_RESIDUAL", "fp32").lower() not in ("bf16", "bfloat16")


def stream_dtype(compute_dtype):
    """dtype of the residual stream for a model computing in `compute_dtype`."""
    return torch.float32 if (compute_dtype == torch.bfloat16 and RESIDUAL_F32 and not SP.enabled()) else compute_dtype


def gacc(param):
    """fp32 gradient accumulator of a parameter (kernels add into it)."""
    if param is None or not param.requires_grad:
        return None
    if param.grad is None:
        param.grad = torch.zeros_like(param, dtype=torch.float32)
    return param.grad


def mask_u8(mask, B, T, device):
    """(B,T) bool/float padding mask (True/1 = valid frame) -> flat uint8 (B*T) or None."""
    if mask is None:
        return None
    m = mask.to(device=device)
    if m.dtype != torch.bool:
        m = m != 0
    return m.reshape(B * T).contiguous().view(torch.uint8)


# ----------------------------------------------------------------------------------------------------
# Deferred parameter-gradient reductions: wgrad slabs, bias partials and LayerNorm dgamma/dbeta partial rows stay in
# persistent per-parameter workspaces and ONE smx_reduce_jobs launch per block backward (= per encoder layer) folds
# them into the gradients - instead of ~300 tiny latency-bound launches per step.  The job tables are cached on the
# device (workspace and gradient addresses are stable), so steady-state steps - and hipGraph capture - copy nothing.
# ----------------------------------------------------------------------------------------------------
class _Deferred:
    enabled = os.environ.get("SMX_DEFER_REDUCE", "1") != "0"
    jobs = []        # (src_ptr, dst_ptr, src_stride, ldd, nsrc, rows, cols, alpha)
    pending = set()  # workspace keys written since the last flush
    ws = {}          # key -> persistent uint8 workspace
    cache = {}       # tuple(jobs) -> (jobs_dev, starts_dev, njobs, total_blocks)
    # grouped wgrad: the bf16 weight gradients of a block (= encoder layer) whose dims are multiples of 256 are recorded
    # and computed by ONE smx_wgrad_group launch at the end of the block's backward (SMX_WGRAD_GROUP=0: one slab GEMM each)
    group_enabled = os.environ.get("SMX_WGRAD_GROUP", "1") != "0"
    group = []       # (dz, x, gW, dbias, N, M, K)
    # (from the kernel's minimum of 64 frames: below 2048 the eight slab GEMMs + reductions it replaces are eight latency-bound
    #  launches - C2b training step at B = 1 x 500: 5.36 -> 4.00 ms, C2a at 2 x 375: 7.11 -> 5.43 ms)
    group_min_rows = 64
    # up to this many frames the grouped wgrad runs WITHOUT split-K slabs (smx_wgrad_group_direct: 128 x 128 tiles over all the frames,
    # added into the gradients; SMX_WGRAD_DIRECT_MAX_ROWS=0 switches it off): measured in tools/experiments/r06_smalln/wgrad_direct_bench.py
    group_direct_max_rows = int(os.environ.get("SMX_WGRAD_DIRECT_MAX_ROWS", "8192"))


def _evict_workspaces():
    """Gradient buffers keep being re-allocated at new addresses: drop the stale workspaces (and the tables that name them).
    Only called between producers - never while a group launch is being assembled (an eviction from inside
    _launch_groups dropped the only references to workspaces already attached to the launch's items)."""
    flush_deferred()
    _Deferred.ws.clear()
    _Deferred.cache.clear()


def deferred_ws(key, nbytes, device, check=True):
    """Persistent workspace of one producer call site (keyed by the gradient buffer it feeds).  check=False (the grouped
    wgrad assembling its items): no flush, no eviction - the caller did both before it started."""
    if check and key in _Deferred.pending:
        flush_deferred()                       # the same parameter twice inside one block: reduce the first use now
    t = _Deferred.ws.get(key)
    if check and t is None and len(_Deferred.ws) >= 2048:
        _evict_workspaces()
        _Deferred.pending.add(key)
    if t is None or t.numel() < nbytes or t.device != device:
        t = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        _Deferred.ws[key] = t
    _Deferred.pending.add(key)
    return t


def defer(src_ptr, dst, src_stride, nsrc, rows, cols, alpha=1.0):
    """dst (a (rows, cols) fp32 view or a flat (cols,) one) += alpha * sum_s src[s*src_stride + i*cols + j]."""
    ldd = dst.stride(0) if dst.dim() == 2 else cols
    _Deferred.jobs.append((src_ptr, dst.data_ptr(), src_stride, ldd, nsrc, rows, cols, alpha))


def _launch_groups():
    """The recorded weight gradients of this block: one smx_wgrad_group launch per (frame count, <= 16 weights); slabs / bias
    partials become reduction jobs."""
    recs, _Deferred.group = _Deferred.group, []
    if len(_Deferred.ws) + len(recs) >= 2048:             # eviction check ONCE, before any workspace is attached to an item
        _evict_workspaces()                               # (the group list is already detached: this folds the earlier producers' jobs only)
    by_n = {}
    for r in recs:
        by_n.setdefault(r[4], []).append(r)
    lib = L.lib()
    for N, rs in by_n.items():
        n64 = N                                 # (the kernel stages the N % 64 tail frames itself)
        # (measured, us per layer, slabs + reduce_jobs | direct: d_model 512 - 368 tiles of 128 x 128 - 500 frames 42 | 20, 3750: 83 | 70,
        #  8000: 146 | 135, 12000: 201 | 222; d_model 256 - 100 tiles, less than half a workgroup per CU - 500: 30 | 12, 2000: 38 | 29,
        #  3750: 44 | 46, 8000: 55 | 87)
        tiles128 = sum((M // 128) * (K // 128) for _, _, _, _, _, M, K in rs)
        if (N <= _Deferred.group_direct_max_rows and (tiles128 >= 256 or N <= 2560) and
                all(g.stride(1) == 1 and g.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0 for _, _, g, _, _, _, _ in rs)):
            # small batches (round 6): no K-slices, no slabs, no reduction job - every tile adds its product into the gradient
            for c0 in range(0, len(rs), L.WGRAD_GROUP_MAX):
                ops.wgrad_group_direct([(dz, x, gW, dbias, M, K) for dz, x, gW, dbias, _, M, K in rs[c0:c0 + L.WGRAD_GROUP_MAX]], N)
            continue
        for c0 in range(0, len(rs), L.WGRAD_GROUP_MAX):
            chunk = rs[c0:c0 + L.WGRAD_GROUP_MAX]
            items = (L.WgradItem * len(chunk))()
            for it, (dz, x, gW, dbias, _, M, K) in zip(items, chunk):
                it.dZ, it.lddz, it.X, it.ldx = dz.data_ptr(), dz.stride(0), x.data_ptr(), x.stride(0)
                it.M, it.K, it.want_bias = M, K, int(dbias is not None)
            splits = lib.smx_wgrad_group_splits(n64, items, len(chunk))
            for it, (dz, x, gW, dbias, _, M, K) in zip(items, chunk):
                ws = deferred_ws(gW.data_ptr(), lib.smx_wgrad_group_workspace(M, K, splits), dz.device, check=False)
                it.workspace = ws.data_ptr()
            ops.wgrad_group(items, len(chunk), n64, splits)
            for it, (dz, x, gW, dbias, _, M, K) in zip(items, chunk):
                defer(it.workspace, gW, M * K, splits, M, K)
                if dbias is not None:
                    defer(it.workspace + 4 * splits * M * K, dbias, M, splits, 1, M)


def flush_deferred():
    if _Deferred.group:
        _launch_groups()
    if not _Deferred.jobs:
        _Deferred.pending.clear()
        return
    key = tuple(_Deferred.jobs)
    ent = _Deferred.cache.get(key)
    if ent is None:
        arr = (L.ReduceJob * len(key))()
        starts = [0]
        for i, job in enumerate(key):
            src, dstp, st, ldd, nsrc, rows, cols, alpha = job[:8]
            src_ld = job[8] if len(job) > 8 else 0           # source row stride (0 = cols)
            vec = int(cols % 4 == 0 and ldd % 4 == 0 and st % 4 == 0 and src_ld % 4 == 0 and src % 16 == 0 and dstp % 16 == 0)
            arr[i] = L.ReduceJob(src, dstp, st, ldd, nsrc, rows, cols, alpha, vec, src_ld)
            starts.append(starts[-1] + L.lib().smx_reduce_job_blocks(ctypes.byref(arr[i])))
        dev = next(iter(_Deferred.ws.values())).device
        jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        starts_dev = torch.tensor(starts, dtype=torch.int32).to(dev)
        ent = (jobs_dev, starts_dev, len(key), starts[-1])
        _Deferred.cache[key] = ent
    ops.reduce_jobs(ent[0], ent[1], ent[2], ent[3])
    _Deferred.jobs = []
    _Deferred.pending.clear()


def _wgrad(dz, x, gW, N, M, K, dbias):
    """gW (M, K) += dz^T x (and dbias += column sums of dz): immediately, or as slabs + a deferred reduction job."""
    if not _Deferred.enabled or K % 4 != 0:
        ops.wgrad(dz, x, gW, N, M, K, dbias=dbias)
        return
    key = gW.data_ptr()
    if (_Deferred.group_enabled and dz.dtype == torch.bfloat16 and M % 256 == 0 and K % 256 == 0
            and N >= _Deferred.group_min_rows and dz.stride(0) % 8 == 0 and x.stride(0) % 8 == 0
            and dz.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0):
        if key in _Deferred.pending:
            flush_deferred()                   # the same parameter twice inside one block: finish the first use now
        _Deferred.pending.add(key)
        _Deferred.group.append((dz, x, gW, dbias, N, M, K))
        return
    ws = deferred_ws(key, L.lib().smx_linear_wgrad_workspace(N, M, K, 1), dz.device)
    nslabs, stride, boff = ops.wgrad_partial(dz, x, N, M, K, ws, want_bias=dbias is not None)
    defer(ws.data_ptr(), gW, stride, nslabs, M, K)
    if dbias is not None:
        defer(ws.data_ptr() + 4 * boff, dbias, M, nslabs, 1, M)


class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, run, done, *params):
        y, bwd = run(x, True)
        ctx.bwd, ctx.done = bwd, done
        ctx.n = len(params)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = ctx.bwd(dy)
        flush_deferred()        # the block's parameter gradients are final before its bucket is all-reduced
        if ctx.done is not None:
            ctx.done()          # e.g. launch this block's gradient-bucket all-reduce (trainer.FlatAdamW)
        return (dx, None, None) + (None,) * ctx.n


def block(x, run, params, on_bwd_done=None):
    """Run `run(x, need_bwd) -> (y, bwd)`; hook bwd into autograd when anything upstream needs gradients."""
    params = [p for p in params if p is not None]
    need = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    if not need:
        return run(x, False)[0]
    return _BlockFn.apply(x, run, on_bwd_done, *params)


# ----------------------------------------------------------------------------------------------------
# Linear (+bias +act +mask +residual +side input) forward / backward on 2-D row views
# ----------------------------------------------------------------------------------------------------
def _vec_ok(*ts):
    """16-byte aligned base and a leading dimension that keeps every row 16-byte aligned (what the vector kernels need)."""
    return all(t is None or (t.data_ptr() % 16 == 0 and (t.stride(0) * t.element_size()) % 16 == 0) for t in ts)


# SMX_LN_FUSE = 1 (default) both directions, 0 none, "fwd" / "bwd": only the forward (SMX_EPI_LN_FWD) / only the backward (SMX_EPI_LN_BWD)
# fusion (C2b B = 128: 19.79 both, 19.90 without the forward, 20.84 without the backward one)
_LN_FUSE_FWD = os.environ.get("SMX_LN_FUSE", "1").lower() != "bwd"
_LN_FUSE_BWD = os.environ.get("SMX_LN_FUSE", "1").lower() != "fwd"


def ln_next_ok(x, M, ln_next, W=None, res=None, training=True, wparam=None):
    """Can - and should - the LayerNorm that follows a Linear (its output: N x M) ride in that GEMM's epilogue (SMX_EPI_LN_FWD)?
    W / res: the weight (view) and residual the GEMM will be given - a column slice with an odd offset is not 16-byte aligned
    and the fused instantiation has no scalar path.
    training: at M = 512 the row-complete tile is ONE workgroup per CU, whose epilogue nothing overlaps; in a training step the
    fusion still wins (C2a 49.2 -> 48.4 ms, each direction ~0.75 ms), in a forward-only pass the standalone LayerNorm (a pure
    stream at 5.9 TB/s, paired with the next layer's) is faster (C5 forward 60.2 vs 62.0 ms fused, C2a forward 16.97 vs 17.25;
    tools/experiments/ab_lnfuse_d512.sh): not fused there."""
    if (wparam is not None and ln_next is not None and _LN_FUSE and _LN_FUSE_FWD and x.dtype == torch.bfloat16 and _vec_ok(x, res) and
            splitk_cfg(x.shape[0], M, x.shape[1], x.dtype) is not None):
        return True                                        # small batch: split-K slabs, the LayerNorm rides in the reducer (linear_fwd)
    if M > 256 and not training:
        return False
    return (_LN_FUSE and _LN_FUSE_FWD and ln_next is not None and x.dtype == torch.bfloat16 and x.shape[0] >= _LN_FUSE_MIN_ROWS and _vec_ok(x, W, res) and
            L.lib().smx_gemm_ln_fused_ok(L.BF16, x.shape[0], M, x.shape[1]) == 1)


def linear_fwd(x, W, bias=None, act=L.ACT_NONE, mask=None, res=None, alpha=1.0, c0=None, c0_mode=L.C0_NONE,
               c0_div=0, save_z=False, out=None, out_f32=False, drop=None, c0_post=False, ln_next=None, ln_post=None,
               drop_cols=0, wparam=None):
    """ln_next = (gamma, beta, eps, act, want_stats[, stream_out[, pair]]): the LayerNorm that follows this Linear runs in the GEMM
    epilogue (check ln_next_ok first); ln_post (a list) receives (LN output, stats | None).  pair = (gamma2, beta2, eps2): a second
    LayerNorm of the first one's output in the same epilogue (check ln_pair_ok); ln_post then receives a second entry,
    (LN2 output in x.dtype, stats2 | None)."""
    N, K = x.shape
    M = W.shape[0]
    if res is not None and res.dtype == torch.float32 and x.dtype != torch.float32:
        out_f32 = True                                   # fp32 residual stream: res + alpha * (bf16 GEMM) is the next stream tensor
    if out is None:
        out = torch.empty((N, M), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    z = torch.empty((N, M), dtype=x.dtype, device=x.device) if (save_z and act != L.ACT_NONE) else None
    # wparam = the fp32 parameter behind W: bias / activation / saved Z / dropout only -> the panel-resident kernel on its packed image
    if (wparam is not None and res is None and c0 is None and not out_f32 and ln_next is None and
            drop_cols % 64 == 0 and out.dtype == x.dtype and panel_ok(x, M, K, act) and _vec_ok(out, z) and _span_ok(out, z)):
        ops.gemm_panel(x, wpacked(wparam, x.dtype, False, bias), out, N, M, K,
                       ops.epilogue(act=act, z=z, drop=drop, row_mask=mask, alpha=alpha, drop_cols=drop_cols))
        return out, z
    lnf = lnf2 = None
    if ln_next is not None:
        g, b, eps, lact, want_stats = ln_next[:5]
        # ln_next[5] = True: the LayerNorm output is itself the stream (the layer-final norm2): stream dtype, else GEMM input dtype
        hy = torch.empty((N, M), dtype=out.dtype if (len(ln_next) > 5 and ln_next[5]) else x.dtype, device=x.device)
        st = torch.empty((N, 2), dtype=torch.float32, device=x.device) if want_stats else None
        lnf = (g.detach(), b.detach(), hy, st, eps, lact)
        ln_post.append((hy, st))
        if len(ln_next) > 6 and ln_next[6] is not None:
            g2, b2, eps2 = ln_next[6]
            hy2 = torch.empty((N, M), dtype=x.dtype, device=x.device)
            st2 = torch.empty((N, 2), dtype=torch.float32, device=x.device) if want_stats else None
            lnf2 = (g2.detach(), b2.detach(), hy2, st2, eps2)
            ln_post.append((hy2, st2))
    e = ops.epilogue(bias=bias, c0=c0, c0_mode=c0_mode, c0_div=c0_div, act=act, z=z, row_mask=mask, res=res,
                     alpha=alpha, out_mode=L.OUT_F32 if out_f32 else L.OUT_T, drop=drop, c0_post=c0_post, ln_fwd=lnf,
                     drop_cols=drop_cols, ln_fwd2=lnf2)
    sk = splitk_cfg(N, M, K, x.dtype) if (wparam is not None and c0 is None and drop_cols % 4 == 0) else None
    if sk is not None and sk[1] == 1 and ln_next is None:
        sk = None                                          # (one slice and no LayerNorm to absorb: the reducer would be a launch more, not less)
    if sk is not None and _vec_ok(x, out, z, res) and out.stride(0) % 4 == 0 and (res is None or res.stride(0) % 4 == 0):
        # small batch: K-slices on the panel kernel -> float32 slabs; the reducer applies this epilogue (and the LayerNorms) to their sum
        slabs = torch.empty((sk[1], N, M), dtype=torch.float32, device=x.device)
        ops.gemm_panel_slabs(x, wpacked(wparam, x.dtype, False, None, kslice=sk[0]), slabs, N, M, sk[0], sk[1])
        ops.slab_epilogue(slabs, sk[1], out, N, M, e)
        return out, z
    assert ln_next is None or L.lib().smx_gemm_ln_fused_ok(L.BF16, N, M, K) == 1, "ln_next given for a shape only the split-K path fuses"
    ops.gemm(L.GEMM_NT, x, W, out, N, M, K, e)
    return out, z


_LN_FUSE = os.environ.get("SMX_LN_FUSE", "1") != "0"   # A/B knob: LayerNorm backward / forward inside the GEMM epilogues
# The LayerNorm-fused GEMM runs one 128-row tile per workgroup (the tile holds whole rows): below ~280 tiles the grid leaves
# CUs idle and each workgroup's serial chain (K loop, four epilogue phases) IS the kernel's duration - 64 us at 16 000 frames
# against 76 us at 64 000.  Measured on the C2b step (fused / separate kernels, ms): B = 16: 8.30 / 5.97, 32: 9.77 / 8.15,
# 48: 11.33 / 10.63, 64: 12.97 / 12.56, 80: 15.20 / 15.74, 96: 16.52 / 17.49, 128: 19.11 / 20.24 -> fuse from 36 864 rows.
# Round 6, re-measured on the round-5/6 kernels (one-pass LayerNorm epilogue, DPP row sums; tools/experiments/r06_smalln/ab_lnfuse_mid*.sh,
# same box, separate | fused, ms): C2b B = 32 (16 000 rows) 7.66 | 8.28, B = 36 9.81 | 9.64, 40: 10.13 | 9.91, 48: 10.66 | 10.45, 56: 11.34 | 10.99,
# 64: 12.00 | 11.53; C2a (d_model 512) B = 32 15.87 | 17.13, 40: 20.70 | 20.55, 48: 22.26 | 21.91, 56: 24.79 | 23.49, 64: 26.85 | 25.31; the recipe's
# 4 fused micro-batches (15 000 rows) 15.40 | 16.73; C4 at 16 000 / 24 000 rows: ties -> fuse from 17 500 rows (was 36 864).
_LN_FUSE_MIN_ROWS = int(os.environ.get("SMX_LN_FUSE_MIN_ROWS", "17500"))
# norm2 of a Conformer layer + the first LayerNorm of the next layer in ONE pass over the float32 stream (smx_layernorm_fwd_pair_x32;
# tests switch it off to compare with the two-launch path)
_LN_PAIR = True


def ln_fusable(ln_spec, N, K_out, dtype, reduce, W=None):
    """Can the LayerNorm (its backward closure's `.spec`) ride in the epilogue of the dgrad GEMM dX (N x K_out) = dZ (N x reduce) W
    with K_out = the LN width?  (bf16, width 256, deferred parameter reductions on.)  `reduce` is the REAL reduce length of
    that GEMM - the producing Linear's output width (d_ffn, 2 l, 2 d): the fused instantiation has no ragged-K path, so a width
    that is not a multiple of 64 must take the standalone LayerNorm kernel (the check used a constant 64)."""
    if not (_LN_FUSE and _LN_FUSE_BWD and _Deferred.enabled and ln_spec is not None and dtype == torch.bfloat16):
        return False
    x = ln_spec["x"]
    return (x.shape[1] == K_out and x.shape[0] == N and N >= _LN_FUSE_MIN_ROWS and gacc(ln_spec["gw_param"]) is not None
            and L.lib().smx_gemm_ln_fused_ok(L.BF16, N, K_out, reduce) == 1 and _vec_ok(x, W))


def linear_bwd(dy, x, W, z, act, mask, alpha, gW, gb, need_dx=True, res_grad=None, dgroup=None, gdiv=0, dx_out=None,
               drop=None, dz_ready=False, up=None, dx_drop=None, ln=None, ln_second=None, dx_split=None, wparam=None, slabs_ok=False):
    """Backward of y = res + alpha*D(act(x W^T + b + c0))*mask (D = the forward's fused dropout, regenerated from its
    seed).  Returns (dx, dz).  gW (M,K) / gb (M) fp32 accumulate.
    dz_ready: dy already IS dZ (a downstream dgrad epilogue fused this layer's act/dropout/mask backward, see `up`).
    up = (z_up, act_up, mask_up, alpha_up, drop_up, gb_up): x is the output alpha_up*D(act_up(z_up))*mask_up of an
    upstream activation layer; the dgrad GEMM's epilogue then emits THAT layer's dZ (SMX_EPI_ACT_GRAD) instead of dX,
    so the (N x K) gradient never makes a separate elementwise pass (gb_up: optional colsum output; normally None,
    the upstream layer's own wgrad yields its bias gradient).
    dx_drop = (p, seed): x is the output of a dropout of that seed; its backward rides in the dgrad epilogue.
    ln = the `.spec` of a LayerNorm backward closure whose OUTPUT is x (x = LN(ln["x"])): the dgrad epilogue then runs that
    LayerNorm backward (SMX_EPI_LN_BWD; check ln_fusable first): returns the gradient w.r.t. ln["x"] (+ res_grad), and,
    with ln_second = (alpha, mask, drop), the pair (dx, alpha * D(dx) * mask).
    dx_split = [(lo, hi, epilogue kwargs | None), ...]: x is a concatenation whose column ranges go to different consumers;
    the dgrad then runs as one GEMM per range over the column slice W[:, lo:hi], each with the first elementwise step of ITS
    consumer in the epilogue (dropout backward, an activation gradient: ops.epilogue keywords); returns the list of outputs."""
    N, M = dy.shape
    K = x.shape[1]
    if drop is not None and drop[0] <= 0.0:
        drop = None
    plain = act == L.ACT_NONE and mask is None and alpha == 1.0 and drop is None
    # the bias gradient (column sums of dZ) is a by-product of the wgrad GEMM, which stages the dZ tiles anyway
    wb = gb is not None and gW is not None and K % 4 == 0 and _WGRAD_BIAS
    gb_pass = None if wb else gb
    if dz_ready or plain:
        dz = dy
        if gb_pass is not None or (dgroup is not None and not dz_ready):
            ops.act_mask_bwd(dy, None, None, L.ACT_NONE, 1.0, None, gb_pass, None if dz_ready else dgroup, gdiv)
    else:
        dz = torch.empty((N, M), dtype=dy.dtype, device=dy.device)
        ops.act_mask_bwd(dy, z, mask, act, alpha, dz, gb_pass, dgroup, gdiv, drop)
    if gW is not None:
        _wgrad(dz, x, gW, N, M, K, gb if wb else None)
    dx = None
    if need_dx:
        if dx_split is not None:
            assert ln is None and up is None and dx_drop is None and res_grad is None and dx_out is None
            outs = []
            for lo, hi, ekw in dx_split:
                o = torch.empty((N, hi - lo), dtype=dy.dtype, device=dy.device)
                ops.gemm(L.GEMM_NN, dz, W[:, lo:hi], o, N, hi - lo, M, ops.epilogue(**ekw) if ekw else None)
                outs.append(o)
            return outs, dz
        if (slabs_ok and wparam is not None and ln is None and up is None and dx_drop is None and res_grad is None and dx_out is None and
                _vec_ok(dz) and splitk_cfg(N, K, M, dz.dtype) is not None):
            # small batch, the caller's next step is a LayerNorm backward: the dgrad as float32 split-K slabs, summed inside that kernel
            ks, ns = splitk_cfg(N, K, M, dz.dtype)
            slabs = torch.empty((ns, N, K), dtype=torch.float32, device=dy.device)
            ops.gemm_panel_slabs(dz, wpacked(wparam, dz.dtype, True, None, kslice=ks), slabs, N, K, ks, ns)
            return ops.Slabs(slabs), dz
        dx = dx_out if dx_out is not None else torch.empty((N, K), dtype=dy.dtype, device=dy.device)
        if ln is not None:
            assert up is None and dx_drop is None and K % 64 == 0
            gw_ln, gb_ln = gacc(ln["gw_param"]).view(-1), gacc(ln["gb_param"]).view(-1)
            tr = L.lib().smx_gemm_ln_tile_rows_for(N, K)      # rows per LayerNorm-fused tile = per partial row pair (64 or 128)
            ntile = (N + tr - 1) // tr
            ws = deferred_ws(gw_ln.data_ptr(), ntile * 2 * K * 4, dy.device)
            dx2 = torch.empty((N, K), dtype=dy.dtype, device=dy.device) if ln_second is not None else None
            lact = (ln["b"].detach(), ln["act"]) if ln["act"] != L.ACT_NONE else None
            e = ops.epilogue(res=res_grad, ln_bwd=(ln["x"], ln["stats"], ln["w"].detach(), ws, dx2, ln_second, lact,
                                                   ln["x"].dtype != dy.dtype))
            ops.gemm(L.GEMM_NN, dz, W, dx, N, K, M, e)
            defer(ws.data_ptr(), gw_ln, 2 * K, ntile, 1, K)
            defer(ws.data_ptr() + 4 * K, gb_ln, 2 * K, ntile, 1, K)
            return ((dx, dx2) if ln_second is not None else dx), dz
        if up is not None:
            assert res_grad is None
            z_up, act_up, mask_up, alpha_up, drop_up, gb_up = up
            # wparam = the fp32 parameter behind W (M, K): the act-grad dgrad on the panel-resident kernel, W packed transposed
            if (wparam is not None and gb_up is None and dx.dtype == dz.dtype and panel_ok(dz, K, M, act_up) and _vec_ok(dx, z_up) and
                    _span_ok(dx, z_up)):
                ops.gemm_panel(dz, wpacked(wparam, dz.dtype, True), dx, N, K, M,
                               ops.epilogue(act=act_up, act_grad_z=z_up, drop=drop_up, row_mask=mask_up, alpha=alpha_up))
                return dx, dz
            e = ops.epilogue(act=act_up, act_grad_z=z_up, row_mask=mask_up, alpha=alpha_up, drop=drop_up, colsum=gb_up)
        else:
            assert dx_drop is None or res_grad is None
            # plain (or dropout-backward) dgrad with a short reduction and a wide output: the panel-resident kernel, W packed transposed
            if (wparam is not None and res_grad is None and dx.dtype == dz.dtype and panel_ok(dz, K, M, L.ACT_NONE) and _vec_ok(dx) and
                    _span_ok(dx)):
                ops.gemm_panel(dz, wpacked(wparam, dz.dtype, True), dx, N, K, M, ops.epilogue(drop=dx_drop))
                return dx, dz
            e = ops.epilogue(res=res_grad, drop=dx_drop)
        ops.gemm(L.GEMM_NN, dz, W, dx, N, K, M, e)
    return dx, dz


# ----------------------------------------------------------------------------------------------------
# VanillaNN: [Linear | ParallelLinear -> act] x blocks ; mask applied in the last block's epilogue
# ----------------------------------------------------------------------------------------------------
def mlp_fwd(x, layers, act, mask, need_bwd, dtype, last_res=None, last_drop=None, last_out=None, last_drop_cols=0):
    """layers: list of dicts {kind: 'linear'|'parallel', W, b, H}.  x (N, F).  Returns (y, saved).
    last_res / last_drop: the last (Linear) layer's epilogue also applies dropout and adds a residual,
    y = last_res + D(act(z)) (hand the same last_drop to mlp_bwd)."""
    saved = []
    n = len(layers)
    for i, ly in enumerate(layers):
        last = i == n - 1
        mk = mask if last else None
        if ly["kind"] == "linear":
            Wc = wcast(ly["W"], dtype)
            y, z = linear_fwd(x, Wc, ly["b"], act, mk, save_z=need_bwd, res=last_res if last else None,
                              drop=last_drop if last else None, out=last_out if last else None,
                              drop_cols=last_drop_cols if last else 0, wparam=ly["W"])
        else:
            Wc = wcast(ly["W"], dtype)                     # (H, f, h)
            H, f, h = Wc.shape
            N = x.shape[0]
            y = torch.empty((N, H * h), dtype=x.dtype, device=x.device)
            z = torch.empty((N, H * h), dtype=x.dtype, device=x.device) if (need_bwd and act != L.ACT_NONE) else None
            e = ops.epilogue(bias=ly["b"], bias_batch_stride=h, act=act, z=z, row_mask=mk)
            ops.gemm(L.GEMM_NN, x[:, :f], Wc[0], y[:, :h], N, h, f, e, batch=H, sa=f, sb=f * h, sc=h,
                     lda=x.stride(0), ldb=h, ldc=H * h)
        saved.append((x, z, mk))
        x = y
    return x, saved


def mlp_bwd(dy, layers, act, saved, dtype, need_dx=True, res_grad=None, dx_out=None, dz_ready=False, last_drop=None, ln=None,
            ln_second=None, dx_split=None):
    """dz_ready: dy already is the LAST layer's dZ and its bias gradient is done (fused upstream, see linear_bwd `up`).
    last_drop: the dropout mlp_fwd fused into the last layer.
    dx_split: handed to the FIRST layer's linear_bwd (a Linear): the input gradient comes back as a list of column ranges."""
    assert dx_split is None or layers[0]["kind"] == "linear"
    n = len(layers)
    for i in range(n - 1, -1, -1):
        ly = layers[i]
        x, z, mk = saved[i]
        first = i == 0
        want_dx = need_dx or not first
        if ly["kind"] == "linear":
            Wc = wcast(ly["W"], dtype)
            # a Linear below an activated Linear: this layer's dgrad epilogue emits the lower layer's dZ and db
            up = None
            if not first and layers[i - 1]["kind"] == "linear" and act != L.ACT_NONE:
                up = (saved[i - 1][1], act, saved[i - 1][2], 1.0, None, None)
            dy, _ = linear_bwd(dy, x, Wc, z, act, mk, 1.0, gacc(ly["W"]), gacc(ly["b"]), want_dx,
                               res_grad if first else None, dx_out=dx_out if first else None, dz_ready=dz_ready, up=up,
                               drop=last_drop if i == n - 1 else None, ln=ln if first else None,
                               ln_second=ln_second if first else None, dx_split=dx_split if first else None, wparam=ly["W"])
            dz_ready = up is not None
        else:
            Wc = wcast(ly["W"], dtype)
            H, f, h = Wc.shape
            N = dy.shape[0]
            gb = gacc(ly["b"])
            if act == L.ACT_NONE and mk is None:
                dz = dy
                if gb is not None:
                    ops.act_mask_bwd(dy, None, None, L.ACT_NONE, 1.0, None, gb.view(-1))
            else:
                dz = torch.empty_like(dy)
                ops.act_mask_bwd(dy, z, mk, act, 1.0, dz, gb.view(-1) if gb is not None else None)
            gW = gacc(ly["W"])
            if gW is not None:   # dW[h] (f x h) += x_h^T dz_h
                ops.wgrad(x[:, :f], dz[:, :h], gW[0], N, f, h, batch=H, sz=f, sx=h, sw=f * h, lddz=x.stride(0),
                          ldx=dz.stride(0), lddw=h)
            if want_dx:          # dx_h = dz_h W[h]^T : NT with B = W[h] viewed (f rows, h reduce-contiguous)
                if first and dx_out is not None:
                    dx = dx_out
                else:
                    dx = torch.empty((N, H * f), dtype=dy.dtype, device=dy.device)
                r = res_grad if first else None
                ops.gemm(L.GEMM_NT, dz[:, :h], Wc[0], dx[:, :f], N, f, h, ops.epilogue(res=r[:, :f] if r is not None else None),
                         batch=H, sa=h, sb=f * h, sc=f, lda=dz.stride(0), ldb=h, ldc=dx.stride(0))
                dy = dx
            else:
                dy = None
    return dy


# ----------------------------------------------------------------------------------------------------
# summary pooling variants: masked mean (per utterance), DynChunk window mean, dense (T,T) weights
# ----------------------------------------------------------------------------------------------------
class DynChunkMask:
    """The (T,T) DynChunk summary mask in closed form (TransformerASR.py:85-110, masked_false_or_true=False):
    frame t of chunk c sees chunks [c-left, c].  `dense()` materialises the reference's boolean matrix."""

    def __init__(self, T, chunk_size, left_context=None):
        self.T, self.chunk_size, self.left_context = T, chunk_size, left_context

    def dense(self, device="cpu"):
        t = torch.arange(self.T, device=device)
        c = t // self.chunk_size
        hi = (c + 1) * self.chunk_size
        m = t[None, :] < hi[:, None]
        if self.left_context is not None:
            lo = (c - self.left_context) * self.chunk_size
            m = m & (t[None, :] >= lo[:, None])
        return m


def _chunk_mean_seqpar(x, out, B, T, chunk, left, reverse=False):
    """The Dynamic Chunk Training summary (chunk c averages the chunks [c - left, c], all earlier ones with left = None:
    summary_mixing.py:224-235 with the mask of TransformerASR.py:85-110) with the time axis sharded over the sequence group.
    Shards hold whole chunks (T % chunk == 0), so a window reaches into other shards only through chunk SUMS.  Round 6: the
    arithmetic is two launches of the chunk kernels themselves (smx_chunk_mean_sharded: global chunk indices and window lengths,
    an additive carry per chunk) around ONE all-gather of small tensors - left = None: the (B, D) totals of the shards, a finite
    left: the (B, left, D) chunk sums at the shard's edge (needs left <= chunks per shard); nothing of activation size is touched
    outside the kernels.  reverse: the transposed operator M^T (x / rowsum(M)) - the same exchange towards the LATER shards."""
    W, r = SP.world(), SP.rank()
    if T % chunk != 0:
        raise ValueError(f"sequence-parallel Dynamic Chunk Training: the frames per rank ({T}) must be a multiple of the chunk size ({chunk})")
    C = T // chunk
    if left is not None and left > C:
        raise NotImplementedError(f"sequence-parallel Dynamic Chunk Training: left context {left} chunks > {C} chunks per rank")
    D = x.shape[1]
    ws = torch.empty((B, C, D), dtype=torch.float32, device=x.device)
    ops.chunk_mean_sharded(x, None, B, T, D, chunk, left, reverse, r * C, 1, ws)
    carry, c0, cn = None, 0, 0
    if left is None:
        # running sums: the shard's total is the last row (forward) / the first (transposed operator)
        tot = SP.all_gather((ws[:, 0] if reverse else ws[:, C - 1]).contiguous())
        others = range(r + 1, W) if reverse else range(r)
        if len(others) > 0:
            carry = sum((tot[q] for q in others), torch.zeros_like(tot[0])).contiguous()
    elif left > 0:
        edges = SP.all_gather((ws[:, :left] if reverse else ws[:, C - left:]).contiguous())     # (B, left, D): what the neighbour's windows reach
        if reverse and r < W - 1:
            carry, c0, cn = edges[r + 1].cumsum(1).contiguous(), C - left, left    # local chunk C - left + i reaches the next shard's chunks [0, i]
        elif not reverse and r > 0:
            carry, c0, cn = edges[r - 1].flip(1).cumsum(1).flip(1).contiguous(), 0, left   # local chunk i reaches back to chunk C - left + i
    ops.chunk_mean_sharded(None, out, B, T, D, chunk, left, reverse, r * C, 2, ws, carry, c0, cn)
    return out


def _expdecay_seqpar(x, out, B, T, decay, reverse=False):
    """The expdecay summary with the time axis sharded over the sequence group: x / out are this rank's (B*T, D) rows of
    a (B, world*T, D) sequence.  (M x)_t = f_t + g_t - x_t is two recurrences, so the rows of a shard see the other shards
    only through ONE (B, D) state per direction.  Round 6: smx_expdecay_mean_sharded runs the O(T) kernels with the GLOBAL frame
    index in the denominators (rowwise.hip ed_inv_den) and the entering states as the scans' initial values; between its two
    phases ONE all-gather of (2, B, D) per rank carries the leaving states and a (B, D) fold turns them into the entering ones.
    reverse: M (x / rowsum(M)), the transposed operator (M is symmetric)."""
    W, r = SP.world(), SP.rank()
    g = float(decay)
    D = x.shape[1]
    ws = torch.empty((L.lib().smx_expdecay_mean_workspace(B, T, D) + 3) // 4, dtype=torch.float32, device=x.device)
    ends = torch.empty((2, B, D), dtype=torch.float32, device=x.device)
    ops.expdecay_mean_sharded(x, None, B, T, g, reverse, r * T, W * T, 1, ends, ws)
    allends = SP.all_gather(ends)                                               # per rank: (f leaving right, g leaving left)
    gT = g ** T
    f_in = torch.zeros_like(ends[0])
    for q in range(r):                                                          # f entering = f_{q} + decay^T f entering q
        f_in = allends[q][0] + gT * f_in
    g_in = torch.zeros_like(f_in)
    for q in range(W - 1, r, -1):
        g_in = allends[q][1] + gT * g_in
    ops.expdecay_mean_sharded(x, out, B, T, g, reverse, r * T, W * T, 2, torch.stack([f_in, g_in]).contiguous(), ws)
    return out


def _dense_pool_fwd(s, B, T, Wn):
    """sbar[b] = Wn (T,T) @ s[b]  (Wn already row-normalised, compute dtype)."""
    D = s.shape[1]
    out = torch.empty((B * T, D), dtype=s.dtype, device=s.device)
    ops.gemm(L.GEMM_NN, Wn, s, out, T, D, T, None, batch=B, sa=0, sb=T * s.stride(0), sc=T * D)
    return out


def _dense_pool_bwd(dsbar, B, T, Wn, ds_out):
    """ds[b] = Wn^T @ dsbar[b]  (TN: A = Wn stored (K=T, N=T))."""
    D = dsbar.shape[1]
    ops.gemm(L.GEMM_TN, Wn, dsbar, ds_out, T, D, T, ops.epilogue(), batch=B, sa=0, sb=T * dsbar.stride(0),
             sc=T * ds_out.stride(0))
    return ds_out


# ----------------------------------------------------------------------------------------------------
# the SummaryMixing cell
# ----------------------------------------------------------------------------------------------------
def cell_run(P, cfg, B, T, mask, sum_mask, p_drop=0.0):
    """Build run(x, need_bwd) for the cell.  P: parameter dict, cfg: mode/act/l.
    Returns a closure operating on (B,T,d) tensors.  `skip_is_input_res`: optional (N,s) residual added to the
    output (Conformer `x + skip`, Conformer.py:530) -- its gradient is returned by bwd as a second value."""
    mode, act, l = cfg["mode"], cfg["act"], cfg["local_proj_out_dim"]

    def run(x3, need_bwd, res=None, ln_next=None, out=None, out_drop=None):
        """ln_next = (gamma, beta, eps) of the LayerNorm that follows the cell output: run in the merge GEMM's epilogue where
        possible; a third value (LN(y), stats) | None is then returned.
        out / out_drop = (p, seed): the caller applies dropout to the cell output and wants it in a (N, s_out) view of its own
        buffer (the Branchformer's merge input): both ride in the merge GEMM's epilogue, and the backward takes the dropout
        backward into its first pass (not for the lite mode)."""
        dtype = x3.dtype
        x = ops.rows2d(x3)
        N = x.shape[0]
        dev = x.device
        pool_kind = "mean"
        Wn = None
        sm = sum_mask
        decay = None
        def _sdim():
            ly = P["summary_proj"][-1]
            return ly["W"].shape[0] if ly["kind"] == "linear" else ly["W"].shape[0] * ly["W"].shape[2]
        if mode == "SummaryMixing-expdecay" and sm is None and _sdim() % 8 == 0:
            # no sum_mask: (M s)/rowsum(M) with M_ij = decay^|i-j| is a two-sided exponential filter -> O(T) kernels
            # (smx_expdecay_mean_*); the frozen decay constant (summary_mixing.py:154-157) is read once
            decay = cfg["decay"]() if "decay" in cfg else float(P["decay_constant"].detach().float().cpu())
            pool_kind = "expdecay"
        elif mode == "SummaryMixing-expdecay":
            # Laplace weights (summary_mixing.py:316-365): M_ij = decay^|i-j| * binary_mask
            idx = torch.arange(T, device=dev)
            lap = torch.exp((idx[None, :] - idx[:, None]).abs().float() * torch.log(P["decay_constant"].detach().float()))
            if isinstance(sm, DynChunkMask):
                sm = sm.dense(dev)
            if sm is not None:
                lap = lap * sm.to(device=dev, dtype=torch.float32)
            sm = lap
        if mode == "SummaryMixing-lite":
            sm = None                                   # the lite branch ignores sum_mask (:286-310)
        if isinstance(sm, DynChunkMask):
            pool_kind = "chunk"
        elif sm is not None:
            pool_kind = "dense"
            w = sm.to(device=dev, dtype=torch.float32)
            Wn = ops.cast((w / w.sum(dim=1, keepdim=True)).contiguous(), dtype)   # mask prep (T,T): plumbing

        # ---- projections -------------------------------------------------------------------------
        cat = s1 = s2 = None
        cat_has_local = False
        if mode == "SummaryMixing-fast":
            gp = P["global_proj"]
            if p_drop > 0.0 and len(gp) == 1 and gp[0]["kind"] == "linear" and l % 8 == 0:
                # training: the projection writes straight into the merge input cat = [D(local) | s]: the dropout of the
                # local half (summary_mixing.py:282-284) rides in its epilogue (columns < l only, mask indexed (n, l)); the
                # summary half is pooled from there and then overwritten by the dropped broadcast of its mean
                s1 = ops.new_dropout_seed()
                cat = torch.empty((N, 2 * l), dtype=dtype, device=dev)
                g, sv_g = mlp_fwd(x, gp, act, mask, need_bwd, dtype, last_drop=(p_drop, s1), last_out=cat, last_drop_cols=l)
                cat_has_local = True
            else:
                g, sv_g = mlp_fwd(x, gp, act, mask, need_bwd, dtype)               # (N, 2l)
            local, s = g[:, :l], g[:, l:]
            sv_l = sv_s = None
        elif mode == "SummaryMixing-lite":
            s, sv_s = mlp_fwd(x, P["summary_proj"], act, mask, need_bwd, dtype)
            local = None
            sv_l = sv_g = None
        else:
            lp = P["local_proj"]
            lw_ = lp[-1]["W"].shape[0] if lp[-1]["kind"] == "linear" else 0
            if p_drop > 0.0 and lw_ and lw_ % 8 == 0 and _sdim() % 8 == 0:
                # training: the local projection's last Linear writes D(local) straight into the merge input
                # cat = [D(local) | D(repeat(sbar))] (summary_mixing.py:237-239): its dropout rides in the GEMM epilogue
                s1 = ops.new_dropout_seed()
                cat = torch.empty((N, lw_ + _sdim()), dtype=dtype, device=dev)
                local, sv_l = mlp_fwd(x, lp, act, mask, need_bwd, dtype, last_drop=(p_drop, s1), last_out=cat[:, :lw_])
                cat_has_local = True
            else:
                local, sv_l = mlp_fwd(x, lp, act, mask, need_bwd, dtype)
            s, sv_s = mlp_fwd(x, P["summary_proj"], act, mask, need_bwd, dtype)
            sv_g = None
        sdim = s.shape[1]

        # ---- summary ------------------------------------------------------------------------------
        inv = None
        pool_fused = False
        sp = SP.enabled()
        if sp and pool_kind not in ("mean", "expdecay", "chunk"):
            raise NotImplementedError("sequence-parallel mode supports the per-utterance mean, the Dynamic Chunk Training mask and "
                                      "the mask-free expdecay summary (no dense sum_mask)")
        if sp and pool_kind == "expdecay":
            sbar = torch.empty((N, sdim), dtype=dtype, device=dev)
            _expdecay_seqpar(s, sbar, B, T, decay)
        elif sp and pool_kind == "chunk":
            sbar = torch.empty((N, sdim), dtype=dtype, device=dev)
            _chunk_mean_seqpar(s, sbar, B, T, sm.chunk_size, sm.left_context)
        elif sp:
            # time axis sharded over the group: local partial sums + valid-frame counts, ONE all-reduce, then the mean
            ssum, _ = ops.masked_mean(s, mask, B, T, scale=False)
            cnt = (mask.view(B, T).sum(1, dtype=torch.float32) if mask is not None
                   else torch.full((B,), float(T), dtype=torch.float32, device=dev))
            buf = torch.cat([ssum, cnt[:, None]], 1)
            SP.all_reduce_sum(buf)
            inv = (1.0 / buf[:, sdim]).contiguous()
            sbar = (buf[:, :sdim] * inv[:, None]).contiguous()
        elif pool_kind == "mean" and p_drop > 0.0 and cat_has_local and _POOL_FUSE and ops.pool_bcast_ok(B, T, sdim):
            # small batches (round 6): mean over time + repeat + the merge input's dropout in ONE launch, straight into the summary
            # half of cat (fast mode: in place - a workgroup reads all T rows of its columns before it writes any)
            s2 = ops.new_dropout_seed()
            sbar = None
            _, inv = ops.pool_bcast(s, mask, B, T, ds=cat[:, cat.shape[1] - sdim:], scale=True, want_mean=False, want_inv=True,
                                    drop=(p_drop, s2))
            pool_fused = True
        elif pool_kind == "mean":
            sbar, inv = ops.masked_mean(s, mask, B, T, scale=True, want_inv=True)   # (B, sdim) fp32
        elif pool_kind == "chunk":
            sbar = torch.empty((N, sdim), dtype=dtype, device=dev)
            ops.chunk_mean(s, sbar, B, T, sm.chunk_size, sm.left_context)
        elif pool_kind == "expdecay":
            sbar = torch.empty((N, sdim), dtype=dtype, device=dev)
            ops.expdecay_mean(s, sbar, B, T, decay)
        else:
            sbar = _dense_pool_fwd(s, B, T, Wn)

        if mode == "SummaryMixing-lite":
            if res is None:
                y3 = ops.cast(sbar, dtype).unsqueeze(1).expand(B, T, sdim)        # stride-0 view like :308
            else:                                                                 # inside an encoder layer: res + summary
                yb = torch.empty((N, sdim), dtype=res.dtype, device=dev)           # (res.dtype: the stream may be float32)
                ops.bcast_rows(sbar, None, yb, B, T)
                y3 = ops.axpby(1.0, res, 1.0, yb).view(B, T, sdim)

            def bwd_lite(dy3, dz_in=None):
                assert dz_in is None
                dy = ops.rows2d(dy3.contiguous())
                dsbar, _ = ops.masked_mean(dy, None, B, T, scale=False)            # sum over time
                SP.all_reduce_sum(dsbar)                                           # (sequence-parallel: over all shards)
                ds = torch.empty((N, sdim), dtype=dtype, device=dev)
                ops.bcast_rows(dsbar, inv, ds, B, T)
                dx = mlp_bwd(ds, P["summary_proj"], act, sv_s, dtype)
                return dx.view(B, T, -1)
            return (y3, (bwd_lite if need_bwd else None), None) if ln_next is not None else (y3, (bwd_lite if need_bwd else None))

        # ---- merge: y = res + act(local W_l^T + sbar W_s^T + b) -------------------------------------
        mg = P["summary_local_merging"][0]
        Wm = wcast(mg["W"], dtype)                                              # (s_out, l + sdim)
        lw = local.shape[1]
        Wl, Ws = Wm[:, :lw], Wm[:, lw:]
        post = []
        if p_drop > 0.0:
            # training: dropout acts on cat[local, repeat(sbar)] per FRAME (summary_mixing.py:237-239,282-284), which
            # breaks the per-utterance factorisation -> materialise the dropped concatenation once and run K = l + s
            if pool_fused:
                pass                                       # (cat is complete: smx_pool_bcast wrote the dropped broadcast)
            elif cat_has_local:
                s2 = ops.new_dropout_seed()
            else:
                s1, s2 = ops.new_dropout_seed(), ops.new_dropout_seed()
                cat = torch.empty((N, lw + sdim), dtype=dtype, device=dev)
                ops.dropout(local, p_drop, s1, out=cat[:, :lw])
            if pool_fused:
                pass
            elif pool_kind == "mean":
                ops.bcast_rows(sbar, None, cat[:, lw:], B, T, drop=(p_drop, s2))   # repeat + dropout in one pass
            else:
                ops.dropout(sbar, p_drop, s2, out=cat[:, lw:])
            sbar_t = None
            lnn = (ln_next[0], ln_next[1], ln_next[2], L.ACT_NONE, need_bwd) if (ln_next is not None and ln_next_ok(cat, Wm.shape[0], ln_next, Wm, res, need_bwd, wparam=mg["W"])) else None
            y, zm = linear_fwd(cat, Wm, mg["b"], act, None, res=res, save_z=need_bwd, ln_next=lnn, ln_post=post, out=out,
                               drop=out_drop, wparam=mg["W"])
        elif pool_kind == "mean":
            sbar_t = ops.cast(sbar, dtype)                                         # (B, sdim) in compute dtype
            c0, _ = linear_fwd(sbar_t, Ws, None, out_f32=True)                     # (B, s_out) fp32
            lnn = (ln_next[0], ln_next[1], ln_next[2], L.ACT_NONE, need_bwd) if (ln_next is not None and ln_next_ok(local, Wl.shape[0], ln_next, Wl, res, need_bwd)) else None
            y, zm = linear_fwd(local, Wl, mg["b"], act, None, res=res, c0=c0, c0_mode=L.C0_GROUP, c0_div=T,
                               save_z=need_bwd, ln_next=lnn, ln_post=post, out=out, drop=out_drop)
        else:
            sbar_t = sbar
            c0, _ = linear_fwd(sbar, Ws, None, out_f32=True)                       # (N, s_out) fp32
            y, zm = linear_fwd(local, Wl, mg["b"], act, None, res=res, c0=c0, c0_mode=L.C0_ROW, save_z=need_bwd, out=out,
                               drop=out_drop)
        y3 = y.view(B, T, -1)
        post = post[0] if post else None
        if not need_bwd:
            return (y3, None, post) if ln_next is not None else (y3, None)

        def bwd(dy3, ln=None, ln_res=None, ln_second=None, dz_in=None):
            """ln / ln_res / ln_second (only when bwd.can_fuse_ln): the LayerNorm whose output is this cell's input runs its
            backward in the epilogue of the cell's input dgrad (linear_bwd(ln=...)); then 2-D tensors come back: the
            gradient w.r.t. that LayerNorm's input (+ ln_res), or the pair with the second output.
            dz_in: dy * act'(zm) already computed by the producer of dy (see `pre`)."""
            dy = ops.rows2d(dy3) if out_drop is not None else ops.rows2d(dy3 if dy3.is_contiguous() else dy3.contiguous())
            s_out = dy.shape[1]
            gWm, gbm = gacc(mg["W"]), gacc(mg["b"])
            # buffer that receives [dlocal | ds] for the fast mode (one dg for the fused projection)
            if mode == "SummaryMixing-fast":
                dg = torch.empty((N, 2 * l), dtype=dtype, device=dev)
                dlocal_out, ds_out = dg[:, :l], dg[:, l:]
            else:
                dlocal_out = torch.empty((N, lw), dtype=dtype, device=dev)
                ds_out = torch.empty((N, sdim), dtype=dtype, device=dev)
            # the projections whose outputs are `local` (and, fused, `s`): their act/mask backward for the LOCAL columns is
            # fused into the merge dgrad's epilogue (SMX_EPI_ACT_GRAD), which then emits dZ and the bias gradient
            if mode == "SummaryMixing-fast":
                lproj, sv_lp = P["global_proj"], sv_g
            else:
                lproj, sv_lp = P["local_proj"], sv_l
            fuse_local = lproj[-1]["kind"] == "linear"
            up_local = None
            if fuse_local:
                z_lp, mk_lp = sv_lp[-1][1], sv_lp[-1][2]
                up_local = (z_lp[:, :lw] if z_lp is not None else None, act if z_lp is not None else L.ACT_NONE, mk_lp, 1.0,
                            None, None)                   # (the bias gradient comes out of the projection's own wgrad)
                if z_lp is None and mk_lp is None:
                    up_local = None
                    fuse_local = False
            # fast mode + per-utterance mean: the broadcast of the summary gradient applies the act/mask backward of the
            # summary columns of global_proj itself (smx_masked_mean_bwd_act) - no separate act_mask_bwd pass over ds
            sum_done = False

            def bcast_side():
                """(z, mask, act) of the projection that produced the summary columns when its act / mask backward can ride in the
                broadcast of the summary gradient, else None."""
                if mode == "SummaryMixing-fast" and fuse_local:
                    z_g, mk_g = sv_g[-1][1], sv_g[-1][2]
                    if z_g is not None or mk_g is not None:
                        return (z_g[:, l:] if z_g is not None else None, mk_g, act)
                elif mode != "SummaryMixing-fast" and P["summary_proj"][-1]["kind"] == "linear":
                    # full / expdecay modes: the same for the last layer of summary_proj (its mlp_bwd then starts from dZ)
                    z_s, mk_s = sv_s[-1][1], sv_s[-1][2]
                    if z_s is not None or mk_s is not None:
                        return (z_s, mk_s, act if z_s is not None else L.ACT_NONE)
                return None

            def bcast_ds(dsbar_):
                nonlocal sum_done
                SP.all_reduce_sum(dsbar_)                  # sequence-parallel: the mean's gradient sums over every shard
                side = bcast_side()
                if side is not None:
                    ops.bcast_rows_act_bwd(dsbar_, inv, ds_out, B, T, side[0], side[1], side[2])
                    sum_done = True
                    return
                ops.bcast_rows(dsbar_, inv, ds_out, B, T)

            def sum_bcast_ds(dsd_):
                """ds_out = broadcast over t of (sum_t dsd_) * inv [* act'(z) * mask]: one launch for small batches (smx_pool_bcast)."""
                nonlocal sum_done
                if _POOL_FUSE and not SP.enabled() and ops.pool_bcast_ok(B, T, dsd_.shape[1]):
                    side = bcast_side()
                    z_, mk_, a_ = side if side is not None else (None, None, L.ACT_NONE)
                    ops.pool_bcast(dsd_, None, B, T, ds=ds_out, scale=False, want_mean=False, inv_in=inv, z=z_, mask_out=mk_,
                                   act=a_ if z_ is not None else L.ACT_NONE)
                    sum_done = side is not None
                    return
                dsbar_, _ = ops.masked_mean(dsd_, None, B, T, scale=False)               # sum over time
                bcast_ds(dsbar_)
            if p_drop > 0.0:
                # dgrad of the K = l + s merge as two GEMMs over the column halves of W: the dropout backward of each half
                # (and the local half's act/mask backward) rides in the epilogue instead of separate passes
                wb = gbm is not None and gWm is not None and (lw + sdim) % 4 == 0 and _WGRAD_BIAS
                if dz_in is not None:
                    dzm = dz_in
                    if not wb and gbm is not None:
                        ops.act_mask_bwd(dzm, None, None, L.ACT_NONE, 1.0, None, gbm)
                else:
                    dzm = torch.empty((N, s_out), dtype=dtype, device=dev)
                    ops.act_mask_bwd(dy, zm, None, act, 1.0, dzm, None if wb else gbm, drop=out_drop)
                if gWm is not None:
                    _wgrad(dzm, cat, gWm, N, s_out, lw + sdim, gbm if wb else None)
                if fuse_local:
                    e = ops.epilogue(act=up_local[1], act_grad_z=up_local[0], row_mask=up_local[2], drop=(p_drop, s1),
                                     colsum=up_local[5]) if up_local[0] is not None else \
                        ops.epilogue(row_mask=up_local[2], drop=(p_drop, s1), colsum=up_local[5])
                else:
                    e = ops.epilogue(drop=(p_drop, s1))
                ops.gemm(L.GEMM_NN, dzm, Wm[:, :lw], dlocal_out, N, lw, s_out, e)
                dsd = torch.empty((N, sdim), dtype=dtype, device=dev)
                ops.gemm(L.GEMM_NN, dzm, Wm[:, lw:], dsd, N, sdim, s_out, ops.epilogue(drop=(p_drop, s2)))
                if pool_kind == "mean":
                    sum_bcast_ds(dsd)
                elif pool_kind == "chunk":
                    (_chunk_mean_seqpar if SP.enabled() else ops.chunk_mean)(dsd, ds_out, B, T, sm.chunk_size, sm.left_context, reverse=True)
                elif pool_kind == "expdecay":
                    (_expdecay_seqpar if SP.enabled() else ops.expdecay_mean)(dsd, ds_out, B, T, decay, reverse=True)
                else:
                    _dense_pool_bwd(dsd, B, T, Wn, ds_out)
            elif pool_kind == "mean":
                _, dzm = linear_bwd(dy if dz_in is None else dz_in, local, Wl, zm, act, None, 1.0,
                                    gWm[:, :lw] if gWm is not None else None, gbm, True, None, dx_out=dlocal_out, up=up_local,
                                    dz_ready=dz_in is not None, drop=out_drop)
                # per-utterance sums of dZ (the gradient of the C0 side input) through the fixed-order pool kernel: the
                # fused variant (smx_act_mask_bwd dgroup) adds with fp32 atomics, i.e. not bit-reproducibly
                dc0, _ = ops.masked_mean(dzm, None, B, T, scale=False)
                dc0_t = ops.cast(dc0, dtype)
                if gWm is not None:      # dW_s += dc0^T sbar
                    ops.wgrad(dc0_t, sbar_t, gWm[:, lw:], B, s_out, sdim)
                dsbar = torch.empty((B, sdim), dtype=torch.float32, device=dev)
                ops.gemm(L.GEMM_NN, dc0_t, Ws, dsbar, B, sdim, s_out, ops.epilogue(out_mode=L.OUT_F32))
                bcast_ds(dsbar)
            else:
                _, dzm = linear_bwd(dy if dz_in is None else dz_in, local, Wl, zm, act, None, 1.0,
                                    gWm[:, :lw] if gWm is not None else None, gbm, True, None, dx_out=dlocal_out, up=up_local,
                                    dz_ready=dz_in is not None, drop=out_drop)
                if gWm is not None:      # dW_s += dzm^T sbar
                    ops.wgrad(dzm, sbar_t, gWm[:, lw:], N, s_out, sdim)
                dsb = torch.empty((N, sdim), dtype=dtype, device=dev)
                ops.gemm(L.GEMM_NN, dzm, Ws, dsb, N, sdim, s_out, None)
                if pool_kind == "chunk":
                    (_chunk_mean_seqpar if SP.enabled() else ops.chunk_mean)(dsb, ds_out, B, T, sm.chunk_size, sm.left_context, reverse=True)
                elif pool_kind == "expdecay":
                    (_expdecay_seqpar if SP.enabled() else ops.expdecay_mean)(dsb, ds_out, B, T, decay, reverse=True)
                else:
                    _dense_pool_bwd(dsb, B, T, Wn, ds_out)
            if mode == "SummaryMixing-fast":
                if fuse_local:
                    # dg[:, :l] already holds dZ (and db[:l] is done): finish the summary columns in place
                    z_g, mk_g = sv_g[-1][1], sv_g[-1][2]
                    if (z_g is not None or mk_g is not None) and not sum_done:
                        ops.act_mask_bwd(ds_out, z_g[:, l:] if z_g is not None else None, mk_g,
                                         act if z_g is not None else L.ACT_NONE, 1.0, ds_out, None)
                if ln is not None:
                    return mlp_bwd(dg, P["global_proj"], act, sv_g, dtype, dz_ready=fuse_local, res_grad=ln_res, ln=ln,
                                   ln_second=ln_second)
                dx = mlp_bwd(dg, P["global_proj"], act, sv_g, dtype, dz_ready=fuse_local)
            else:
                dx = mlp_bwd(dlocal_out, P["local_proj"], act, sv_l, dtype, dz_ready=fuse_local)
                dx = mlp_bwd(ds_out, P["summary_proj"], act, sv_s, dtype, res_grad=dx, dz_ready=sum_done)
            return dx.view(B, T, -1)
        bwd.can_fuse_ln = (mode == "SummaryMixing-fast" and len(P["global_proj"]) == 1 and P["global_proj"][0]["kind"] == "linear")
        if bwd.can_fuse_ln:      # the dgrad that would carry the LayerNorm backward: dX = dG (N x 2l) W_g - what ln_fusable must check
            bwd.ln_reduce, bwd.ln_W = P["global_proj"][0]["W"].shape[0], wcast(P["global_proj"][0]["W"], dtype)
        # what the cell does first to its incoming gradient: dy * act'(zm) (a producer that can, writes it as a second output)
        bwd.pre = (1.0, None, None, zm, act) if (zm is not None and act != L.ACT_NONE) else None
        return (y3, bwd, post) if ln_next is not None else (y3, bwd)
    return run


# ----------------------------------------------------------------------------------------------------
# LayerNorm block, FFN module, Conformer conv module (2-D row views in, closures out)
# ----------------------------------------------------------------------------------------------------
def ln_fwd(x, w, b, eps, need_bwd, act=L.ACT_NONE, wp=None, bp=None, pre=None, out_dtype=None):
    """y = act(LayerNorm(x)); bwd(dy, res) = res + d/dx.  w/b are flat (D) views; wp/bp name the owning parameters
    when those are multi-dimensional (the (F', C) affine of the conv front-end).
    pre = (y, stats): the GEMM that produced x already ran this LayerNorm in its epilogue (linear_fwd(ln_next=...)).
    out_dtype: dtype of y when x is the float32 residual stream of a bf16 model (gradients then have the dtype of dy)."""
    if pre is not None:
        y, stats = pre
    else:
        y, stats = ops.layernorm_fwd(x, w.detach(), b.detach(), eps, need_bwd, act, out_dtype=out_dtype)

    def bwd(dy, res=None, out=None, second=None, preact=None):
        """second = (alpha, mask, drop): also return alpha * D(dx) * mask, what the NEXT backward block applies first to
        this gradient (its `pre` attribute) - written from the same registers instead of by a separate pass.
        preact = (z, zact): x = zact(z); return the gradient w.r.t. z (check ops.layernorm_bwd_preact_ok first)."""
        gw = gacc(wp).view(-1) if wp is not None else gacc(w)
        gb = gacc(bp).view(-1) if bp is not None else gacc(b)
        if preact is not None:
            assert res is None and second is None
            N, D = x.shape
            if _Deferred.enabled and gw is not None and gb is not None:
                ws = deferred_ws(gw.data_ptr(), L.lib().smx_layernorm_bwd_workspace(N, D), x.device)
                dz = ops.layernorm_bwd_preact(dy, x, w.detach(), b.detach(), stats, preact[0], preact[1], None, None, act, ws=ws, dx_out=out)
                nb = L.lib().smx_layernorm_bwd_blocks(N)
                defer(ws.data_ptr(), gw, 2 * D, nb, 1, D)
                defer(ws.data_ptr() + 4 * D, gb, 2 * D, nb, 1, D)
                return dz
            return ops.layernorm_bwd_preact(dy, x, w.detach(), b.detach(), stats, preact[0], preact[1], gw, gb, act, dx_out=out)
        wide2 = None
        if second is not None and x.shape[1] > 2048:       # the fused second output exists for D <= 2048: do it in separate
            wide2, second = second, None                   # passes, but ALWAYS return the pair the caller unpacks
        if isinstance(dy, ops.Slabs) and not (_Deferred.enabled and gw is not None and gb is not None and wide2 is None):
            dy = dy.sum(x.dtype if x.dtype != torch.float32 else torch.bfloat16)      # (not reached by the encoder layers: plain torch)
        if _Deferred.enabled and gw is not None and gb is not None:
            N, D = x.shape
            ws = deferred_ws(gw.data_ptr(), L.lib().smx_layernorm_bwd_workspace(N, D), x.device)
            dx = ops.layernorm_bwd(dy, x, w.detach(), b.detach(), stats, None, None, res, act, ws=ws, dx_out=out, second=second)
            nb = L.lib().smx_layernorm_bwd_blocks(N)
            defer(ws.data_ptr(), gw, 2 * D, nb, 1, D)
            defer(ws.data_ptr() + 4 * D, gb, 2 * D, nb, 1, D)
        else:
            dx = ops.layernorm_bwd(dy, x, w.detach(), b.detach(), stats, gw, gb, res, act, dx_out=out, second=second)
        if wide2 is not None:
            a2, m2, d2 = wide2[:3]
            dx2 = ops.dropout(dx, d2[0], d2[1]) if (d2 is not None and d2[0] > 0.0) else dx
            dx2 = ops.act_mask_bwd(dx2, None, m2, L.ACT_NONE, a2, torch.empty_like(dx)) if (m2 is not None or a2 != 1.0 or dx2 is dx) else dx2
            return dx, dx2
        return dx
    # what a dgrad GEMM needs to run this backward in its own epilogue (linear_bwd(ln=...))
    bwd.spec = {"x": x, "w": w, "b": b, "stats": stats, "act": act, "gw_param": wp if wp is not None else w,
                "gb_param": bp if bp is not None else b}
    return y, (bwd if need_bwd else None)


def dwconv_bwd_deferred(dy, p_, wd, bd, gwd, gbd, B, T, D, k, glu, pad_mode, chunk, gate=None, dgate_out=None):
    """ops.dwconv_bwd with the tap / bias partial rows folded by the block's one smx_reduce_jobs launch (k = 31 path)."""
    if not (_Deferred.enabled and gwd is not None):
        return ops.dwconv_bwd(dy, p_, wd, bd, gwd, gbd, B, T, D, k, glu, pad_mode, chunk, gate=gate, dgate_out=dgate_out)
    nbytes = L.lib().smx_dwconv1d_glu_bwd_workspace(B, T, D, k)
    ws = deferred_ws(gwd.data_ptr(), nbytes, dy.device)
    dp, dg, deferred = ops.dwconv_bwd(dy, p_, wd, bd, gwd, gbd, B, T, D, k, glu, pad_mode, chunk, gate=gate, dgate_out=dgate_out,
                                      ws=ws)
    if deferred:
        rows = L.lib().smx_dwconv1d_glu_bwd_partial_rows(L.BF16 if dy.dtype == torch.bfloat16 else L.F32, B, T, D, k, 1 if glu else 0, pad_mode, chunk, int(gate is not None))
        _Deferred.jobs.append((ws.data_ptr(), gwd.data_ptr(), D * (k + 1), k, rows, D, k, 1.0, k + 1))
        if gbd is not None:
            _Deferred.jobs.append((ws.data_ptr() + 4 * k, gbd.data_ptr(), D * (k + 1), 1, rows, D, 1, 1.0, k + 1))
    return dp, dg


def ln_pair_ok(x, M, res, affine, wparam=None):
    """Can the epilogue that runs a LayerNorm behind this Linear (ln_next_ok holds) also run a SECOND LayerNorm on its output
    (smx_gemm_ln_pair_ok: the 128 x 512 tile, float32 residual stream)?"""
    if (wparam is not None and _LN_PAIR and res is not None and res.dtype == torch.float32 and x.dtype == torch.bfloat16 and
            splitk_cfg(x.shape[0], M, x.shape[1], x.dtype) is not None and all(v.is_contiguous() and v.data_ptr() % 16 == 0 for v in affine)):
        return True                                        # (the split-K reducer runs both LayerNorms on the row in registers)
    return (_LN_PAIR and res is not None and res.dtype == torch.float32 and x.dtype == torch.bfloat16 and
            L.lib().smx_gemm_ln_pair_ok(L.BF16, x.shape[0], M, x.shape[1]) == 1 and
            all(v.is_contiguous() and v.data_ptr() % 16 == 0 for v in affine))


def ffn_module_fwd(x, P, act, need_bwd, dtype, alpha=0.5, p=0.0, pre_ln=None, ln_next=None, ln_pair=None):
    """y = x + alpha * D2(W2 D1(act(W1 LN(x) + b1)) + b2)   (Conformer.py:458-472,507,536; D = dropout, p = 0 in eval).
    With p == 0 the second Linear's epilogue carries the residual and alpha (no extra pass).
    pre_ln = (LN(x), stats) when the producer of x already ran this module's LayerNorm in its epilogue; ln_next = (gamma,
    beta, eps) of the LayerNorm that follows the module: it runs in the second Linear's epilogue where possible, and a
    third value (LN(y), stats) | None is returned.  ln_pair = (gamma2, beta2, eps2): the LayerNorm that follows THAT one (the next
    layer's first); where the epilogue can run both, a fourth value (LN2(LN(y)), stats2) | None is returned."""
    h, ln_b = ln_fwd(x, P["ln_w"], P["ln_b"], 1e-5, need_bwd, pre=pre_ln, out_dtype=dtype)
    W1, W2 = wcast(P["W1"], dtype), wcast(P["W2"], dtype)
    d1 = (p, ops.new_dropout_seed()) if p > 0.0 else None       # both dropouts are fused into the GEMM epilogues
    d2 = (p, ops.new_dropout_seed()) if p > 0.0 else None
    a, z1 = linear_fwd(h, W1, P["b1"], act, None, save_z=need_bwd, drop=d1, wparam=P["W1"])
    post = []
    lnn = ((ln_next[0], ln_next[1], ln_next[2], L.ACT_NONE, need_bwd) + tuple(ln_next[3:4])) if (ln_next is not None and ln_next_ok(a, W2.shape[0], ln_next, W2, x, need_bwd, wparam=P["W2"])) else None
    if lnn is not None and ln_pair and len(lnn) > 5 and ln_pair_ok(a, W2.shape[0], x, ln_pair[:2], wparam=P["W2"]):
        lnn = lnn + (ln_pair,)
    y, _ = linear_fwd(a, W2, P["b2"], L.ACT_NONE, None, res=x, alpha=alpha, drop=d2, ln_next=lnn, ln_post=post, wparam=P["W2"])
    post, post2 = (post[0] if post else None), (post[1] if len(post) > 1 else None)
    ret = lambda b: ((y, b, post, post2) if ln_pair is not None else (y, b, post)) if ln_next is not None else (y, b)
    if not need_bwd:
        return ret(None)

    def bwd(dy, dz_in=None, second=None):
        """dz_in: alpha * D2(dy) already computed by the producer of dy (see `pre`); second: forwarded to the module's own
        LayerNorm backward, the producer of the gradient this function returns (then a pair comes back)."""
        # the second Linear's dgrad epilogue applies D1 and act'(z1): it emits dZ1 directly; db1 comes out of W1's wgrad
        if dz_in is not None:
            dz1, _ = linear_bwd(dz_in, a, W2, None, L.ACT_NONE, None, 1.0, gacc(P["W2"]), gacc(P["b2"]), dz_ready=True,
                                up=(z1, act, None, 1.0, d1, None), wparam=P["W2"])
        else:
            dz1, _ = linear_bwd(dy, a, W2, None, L.ACT_NONE, None, alpha, gacc(P["W2"]), gacc(P["b2"]), drop=d2,
                                up=(z1, act, None, 1.0, d1, None), wparam=P["W2"])
        if ln_fusable(ln_b.spec, h.shape[0], h.shape[1], dtype, W1.shape[0], W1):     # the LayerNorm backward rides in the dgrad epilogue
            out, _ = linear_bwd(dz1, h, W1, z1, act, None, 1.0, gacc(P["W1"]), gacc(P["b1"]), dz_ready=True, res_grad=dy,
                                ln=ln_b.spec, ln_second=second)
            return out
        dh, _ = linear_bwd(dz1, h, W1, z1, act, None, 1.0, gacc(P["W1"]), gacc(P["b1"]), dz_ready=True, wparam=P["W1"], slabs_ok=True)
        return ln_b(dh, res=dy, second=second)
    bwd.pre = (alpha, None, d2)          # what this block does first to its incoming gradient: alpha * D2(dy)
    return ret(bwd)

def conv_module_fwd(x, P, act, mask, B, T, need_bwd, dtype, chunk=0, residual=True, p=0.0, pre_ln=None, ln_next=None):
    """y = [x +] mask * Linear(act(LN(dwconv(GLU(pw(LN(x)))))))   (Conformer.py:314-331,532-534).
    pre_ln / ln_next: as in ffn_module_fwd (the module's first LayerNorm done by the producer of x; the LayerNorm that
    follows the module done in the out-projection's epilogue)."""
    d = x.shape[1]
    if SP.enabled():
        r = _conv_module_fwd_sp(x, P, act, mask, B, T, need_bwd, dtype, chunk, residual, p)
        return r + (None,) if ln_next is not None else r
    h, ln1_b = ln_fwd(x, P["ln1_w"], P["ln1_b"], 1e-5, need_bwd, pre=pre_ln, out_dtype=dtype)
    Wp = wcast(P["Wp"], dtype).view(2 * d, d)                    # Conv1d(d,2d,1) weight viewed as a Linear
    p_, _ = linear_fwd(h, Wp, P["bp"], wparam=P["Wp"])
    k = P["wd"].shape[-1]
    wd = P["wd"].detach().reshape(d, k)
    c = ops.dwconv_fwd(p_, wd, P["bd"].detach() if P["bd"] is not None else None, B, T, d, k, True, L.PAD_ZERO, chunk)
    a, ln2_b = ln_fwd(c, P["ln2_w"], P["ln2_b"], 1e-5, need_bwd, act)      # LN + activation fused
    Wo = wcast(P["Wo"], dtype)
    dr = (p, ops.new_dropout_seed()) if p > 0.0 else None   # Linear -> Dropout -> * mask (+ x): one epilogue
    post = []
    lnn = (ln_next[0], ln_next[1], ln_next[2], L.ACT_NONE, need_bwd) if (ln_next is not None and ln_next_ok(a, Wo.shape[0], ln_next, Wo, x if residual else None, need_bwd, wparam=P["Wo"])) else None
    y, _ = linear_fwd(a, Wo, P["bo"], L.ACT_NONE, mask, res=x if residual else None, drop=dr, ln_next=lnn, ln_post=post, wparam=P["Wo"])
    post = post[0] if post else None
    if not need_bwd:
        return (y, None, post) if ln_next is not None else (y, None)

    def bwd(dy, dz_in=None, second=None):
        fuse2 = ln_fusable(ln2_b.spec, a.shape[0], d, dtype, Wo.shape[0], Wo)     # LN2 (+ activation) backward in the out-projection's dgrad
        kw = dict(ln=ln2_b.spec) if fuse2 else {}
        if dz_in is not None:
            da, _ = linear_bwd(dz_in, a, Wo, None, L.ACT_NONE, None, 1.0, gacc(P["Wo"]), gacc(P["bo"]), dz_ready=True, **kw)
        else:
            da, _ = linear_bwd(dy, a, Wo, None, L.ACT_NONE, mask, 1.0, gacc(P["Wo"]), gacc(P["bo"]), drop=dr, **kw)
        dc = da if fuse2 else ln2_b(da)
        gwd = gacc(P["wd"])
        dp, _ = dwconv_bwd_deferred(dc, p_, wd, P["bd"].detach() if P["bd"] is not None else None, gwd.view(d, k),
                                    gacc(P["bd"]), B, T, d, k, True, L.PAD_ZERO, chunk)
        if ln_fusable(ln1_b.spec, h.shape[0], d, dtype, 2 * d, Wp):
            out, _ = linear_bwd(dp, h, Wp, None, L.ACT_NONE, None, 1.0, gacc(P["Wp"]).view(2 * d, d), gacc(P["bp"]),
                                res_grad=dy if residual else None, ln=ln1_b.spec, ln_second=second)
            return out
        dh, _ = linear_bwd(dp, h, Wp, None, L.ACT_NONE, None, 1.0, gacc(P["Wp"]).view(2 * d, d), gacc(P["bp"]), wparam=P["Wp"], slabs_ok=True)
        return ln1_b(dh, res=dy if residual else None, second=second)
    # what this block does first to its incoming gradient: D(dy) * mask (nothing to precompute without mask and dropout)
    bwd.pre = (1.0, mask, dr) if (mask is not None or dr is not None) else None
    bwd.ln1_fused = ln_fusable(ln1_b.spec, h.shape[0], d, dtype, 2 * d, Wp)   # then `second` may carry (.., z, act) of the consumer
    return (y, bwd, post) if ln_next is not None else (y, bwd)


def _conv_module_fwd_sp(x, P, act, mask, B, T, need_bwd, dtype, chunk, residual, p):
    """conv_module_fwd with the time axis sharded (sequence_parallel.py): the (k-1)/2 frames either side come from the
    neighbour ranks as halos of the module INPUT; LN / pointwise conv / GLU are recomputed on them, the depthwise conv
    runs on the extended sequence and only the centre T frames go on.  At the two ends of the whole sequence the halo
    must act as the conv's zero padding, i.e. zero AFTER the GLU: the pointwise output rows there are cleared.
    Dynamic Chunk Convolution (chunk > 0, Conformer.py:190-313): a frame never reads beyond its own chunk, and shards hold whole
    chunks, so there is no right halo; the left halo is rounded up to whole chunks so that the chunk grid of the extended
    sequence is the grid of the whole one."""
    d = x.shape[1]
    k = P["wd"].shape[-1]
    H = (k - 1) // 2
    if chunk:
        if T % chunk != 0:
            raise ValueError(f"sequence-parallel Dynamic Chunk Convolution: the frames per rank ({T}) must be a multiple of the chunk size ({chunk})")
        Hl, Hr = (H + chunk - 1) // chunk * chunk, 0
    else:
        Hl = Hr = H
    if T < Hl:
        raise ValueError(f"sequence-parallel shards need at least {Hl} frames per rank (got {T})")
    Te = T + Hl + Hr
    first_rank, last_rank = SP.rank() == 0, SP.rank() == SP.world() - 1
    x3 = x.view(B, T, d)
    lh, rh = SP.exchange_halos(x3[:, :Hl], x3[:, T - Hl:])
    xe = torch.cat([lh, x3, rh] if Hr else [lh, x3], 1).view(B * Te, d)
    h, ln1_b = ln_fwd(xe, P["ln1_w"], P["ln1_b"], 1e-5, need_bwd)
    Wp = wcast(P["Wp"], dtype).view(2 * d, d)
    p_, _ = linear_fwd(h, Wp, P["bp"])

    def clear_ends(t2, width):
        t3 = t2.view(B, Te, width)
        if first_rank:
            t3[:, :Hl].zero_()
        if last_rank and Hr:
            t3[:, Te - Hr:].zero_()
    clear_ends(p_, 2 * d)
    wd = P["wd"].detach().reshape(d, k)
    ce = ops.dwconv_fwd(p_, wd, P["bd"].detach() if P["bd"] is not None else None, B, Te, d, k, True, L.PAD_ZERO, chunk or 0)
    c = ce.view(B, Te, d)[:, Hl:Hl + T].contiguous().view(B * T, d)
    a, ln2_b = ln_fwd(c, P["ln2_w"], P["ln2_b"], 1e-5, need_bwd, act)
    Wo = wcast(P["Wo"], dtype)
    dr = (p, ops.new_dropout_seed()) if p > 0.0 else None
    y, _ = linear_fwd(a, Wo, P["bo"], L.ACT_NONE, mask, res=x if residual else None, drop=dr)
    if not need_bwd:
        return y, None

    def bwd(dy, dz_in=None, second=None):
        assert dz_in is None and second is None       # (no `pre` attribute: the caller keeps its elementwise passes)
        da, _ = linear_bwd(dy, a, Wo, None, L.ACT_NONE, mask, 1.0, gacc(P["Wo"]), gacc(P["bo"]), drop=dr)
        dc = ln2_b(da)
        dce = torch.zeros((B, Te, d), dtype=dtype, device=x.device)       # the halo outputs were dropped: zero gradient
        dce[:, Hl:Hl + T] = dc.view(B, T, d)
        gwd = gacc(P["wd"])
        dp, _ = ops.dwconv_bwd(dce.view(B * Te, d), p_, wd, P["bd"].detach() if P["bd"] is not None else None,
                               gwd.view(d, k), gacc(P["bd"]), B, Te, d, k, True, L.PAD_ZERO, chunk or 0)
        clear_ends(dp, 2 * d)                                             # nothing flows into the zero padding
        dh, _ = linear_bwd(dp, h, Wp, None, L.ACT_NONE, None, 1.0, gacc(P["Wp"]).view(2 * d, d), gacc(P["bp"]))
        dxe = ln1_b(dh).view(B, Te, d)
        g_lh = dxe[:, :Hl]
        g_first, g_last = SP.return_halo_grads(g_lh, dxe[:, Te - Hr:] if Hr else torch.zeros_like(g_lh))
        dx = dxe[:, Hl:Hl + T].contiguous()
        if Hr:
            dx[:, :Hr] += g_first
        dx[:, T - Hl:] += g_last
        dx = dx.view(B * T, d)
        if residual:
            dx = ops.axpby(1.0, dx, 1.0, dy)
        return dx
    return y, bwd


def encoder_stack(src, layers, make_layer_run, norm, params, compute_dtype=None, pair_next=False):
    """A whole encoder stack (layers + final LayerNorm) as ONE autograd block.  make_layer_run(layer, compute) -> run(x3, need).
    pair_next=True (the Conformer stack): make_layer_run(layer, compute, next_layer) -> run(x3, need, pre_ln=, with_post=True)
    returning (y, bwd, post) - a layer's norm2 and the next layer's first LayerNorm run in one launch and `post` = that second
    LayerNorm's (output, statistics) is handed to the next layer as `pre_ln`.
    compute: dtype of the GEMM operands (default: the input's); the residual stream between the layers is
    stream_dtype(compute) - float32 for a bf16 model by default.  The gradient between two layers never visits autograd (it
    would cast the bf16 gradient of a float32 stream tensor with one aten kernel per layer); every layer's parameter-gradient
    reductions and its bucket hook (`_on_bwd_done`, trainer.FlatAdamW) still run right behind that layer's backward."""
    B, T, d = src.shape
    compute = compute_dtype or src.dtype
    stream = stream_dtype(compute)

    def run(xin, need):
        x = xin
        if x.dtype != stream:
            x = ops.cast(ops.rows2d(x), stream).view(B, T, d)
        bwds = []
        pair = _LN_PAIR and pair_next
        pre = None
        for i, layer in enumerate(layers):
            if pair:
                x, b, pre = make_layer_run(layer, compute, layers[i + 1] if i + 1 < len(layers) else None)(x, need, pre_ln=pre, with_post=True)
            else:
                x, b = make_layer_run(layer, compute)(x, need)
            bwds.append((b, getattr(layer, "_on_bwd_done", None)))
        y, bn = ln_fwd(ops.rows2d(x), norm.weight, norm.bias, norm.eps, need, out_dtype=compute)
        if not need:
            return y.view(B, T, d), None

        def bwd(dy3):
            dy = ops.rows2d(dy3 if dy3.is_contiguous() else dy3.contiguous())
            if dy.dtype != compute:
                dy = ops.cast(dy, compute)
            g = bn(dy).view(B, T, d)
            flush_deferred()
            for b, done in reversed(bwds):
                g = b(g)
                flush_deferred()                       # this layer's parameter gradients are final ...
                if done is not None:
                    done()                             # ... before its bucket is all-reduced
            return g if g.dtype == xin.dtype else ops.cast(ops.rows2d(g), xin.dtype).view(B, T, d)
        return y.view(B, T, d), bwd
    return block(src, run, params)


def final_norm(x3, ln):
    """Encoder-final LayerNorm (eps 1e-6; Conformer.py:738,784 / Branchformer.py:444,489) as one block."""
    B, T, d = x3.shape

    def run(xin, need):
        y, b = ln_fwd(ops.rows2d(xin), ln.weight, ln.bias, ln.eps, need)
        return y.view(B, T, d), ((lambda dy: b(ops.rows2d(dy.contiguous())).view(B, T, d)) if need else None)
    return block(x3, run, [ln.weight, ln.bias])




def input_proj_pe(src3, W, b, pe, T, p=0.0):
    """x = D(src W^T + b) + PE[t]  (TransformerASR.py:349-354,542,547-549).  Without dropout the abs-sine table enters
    the GEMM epilogue as a per-frame side input indexed n % T, so the add costs no extra pass.  The output opens the residual
    stream: float32 for a bf16 model unless SMX_RESIDUAL=bf16 (stream_dtype)."""
    B = src3.shape[0]

    def run(xin, need):
        dtype = xin.dtype
        x = ops.rows2d(xin)
        Wc = wcast(W, dtype)
        dr = (p, ops.new_dropout_seed()) if p > 0.0 else None
        y, _ = linear_fwd(x, Wc, b, c0=pe, c0_mode=L.C0_MOD, c0_div=T, drop=dr, c0_post=True,
                          out_f32=stream_dtype(dtype) == torch.float32 and dtype != torch.float32)
        if not need:
            return y.view(B, T, -1), None

        def bwd(dy3):
            dy = ops.rows2d(dy3 if dy3.is_contiguous() else dy3.contiguous())
            if dy.dtype != dtype:                            # (autograd hands a float32 gradient for the float32 stream tensor)
                dy = ops.cast(dy, dtype)
            dx, _ = linear_bwd(dy, x, Wc, None, L.ACT_NONE, None, 1.0, gacc(W), gacc(b), need_dx=xin.requires_grad, drop=dr)
            return dx.view(xin.shape) if dx is not None else None
        return y.view(B, T, -1), bwd
    return block(src3, run, [W, b])
