"""Utterance-sharded data-parallel training harness for the encoder hot path.

One process per GPU.  All parameters live in ONE flat fp32 buffer (master weights) with parallel flat buffers
for the gradients, the two AdamW moments and the bf16 shadow weights the GEMMs read; ``param.data`` /
``param.grad`` are views, so
  * the HIP kernels accumulate parameter gradients straight into the flat gradient buffer,
  * the gradient all-reduce (RCCL over xGMI, ``torch.distributed`` backend "nccl") runs on a few large
    contiguous buckets - one per encoder layer, launched asynchronously as soon as that layer's backward has
    finished so that it overlaps the rest of the backward pass,
  * global-norm clipping + AdamW + the bf16 shadow refresh are three kernel launches per step
    (smx_sumsq, smx_clip_factor, smx_adamw_step), the clip factor never visits the host.
Forward/backward need no communication: utterances never interact.

Gradient exchange options (xGMI is point-to-point, 7 links x ~153 GB/s per GPU, so a ring collective is per-link bound and
its cost is the bytes on the wire):
  reduce="allreduce"  one all-reduce per bucket (default);
  reduce="rs_ag"      reduce-scatter per bucket, clip + AdamW on this rank's 1/world shard of every bucket only, then an
                      all-gather of the updated weights - the same bytes on the wire as the ring all-reduce, but the
                      weights (not the gradients) travel in the second half, so with grad_dtype=bfloat16 only the
                      gradient half is rounded and the optimizer work per rank drops by 1/world;
  grad_dtype=torch.bfloat16   gradients cross the wire in bf16 (half the bytes; the accumulation inside the kernels, the
                      clip and AdamW stay fp32).
comm_exposed_ms() reports how long the compute stream stood waiting for the collectives (bench.py prints it).
"""
import contextlib
import os

import torch
import torch.distributed as dist

from . import functional as F
from . import ops

ALIGN = 64   # elements; keeps every parameter view 256-byte aligned in fp32 and 128-byte in bf16


def fuse_microbatches(batches, pad_to_longest=False):
    """Gradient accumulation without the loop: the G micro-batches of one optimizer step as ONE batch.

    The reference accumulates `grad_accumulation_factor` forward + backward passes per update (recipes/LibriSpeech/ASR/
    transducer/hparams/conformer_summarymixing_transducer.yaml:65-66, 113-126: 4 x max_batch_len 150 s, "works well for 3090
    24GB GPU, adapt it to your needs").  No operator of the encoder path couples utterances (LayerNorm per frame, the
    padding-masked mean and the depthwise convolution per utterance), so for micro-batches of the SAME padded length the summed
    gradients are the gradients of their concatenation along the batch axis (tests/test_accum_gpu.py: 1e-6 in fp32, the order of
    the fp32 sums over frames is all that differs) - and 4 x 3750 frames fill an MI355X where 3750 leave it launch- and
    latency-bound (bench.py --grad-accum 4 --accum fused: 0.53 -> 0.92 M frames/s).

    pad_to_longest=True also fuses micro-batches of DIFFERENT padded lengths by zero-padding the shorter ones.  That is not the
    same function: the reference's convolution module masks its OUTPUT, not the depthwise convolution's input (Conformer.py:
    327-331), so the last (k-1)/2 frames of an utterance see whatever follows it - zeros at the end of the tensor, LayerNorm /
    bias values of padded frames otherwise.  The reference has exactly this dependence on what a dynamic batch holds; fusing
    changes which utterances end at the tensor's end, nothing else.

    batches: [(src (B_i, T_i, F), wav_len (B_i,) relative lengths), ...].  Returns (src (sum B_i, max T_i, F) zero padded,
    wav_len relative to max T_i: round(wav_len * T) gives every utterance the frame count it had)."""
    if not batches:
        raise ValueError("fuse_microbatches: no micro-batch")
    T = max(int(s.shape[1]) for s, _ in batches)
    srcs, lens = [], []
    for s, wl in batches:
        if s.dim() != 3 or wl.shape[0] != s.shape[0]:
            raise ValueError("fuse_microbatches: expected (B, T, F) features with (B,) relative lengths")
        Ti = int(s.shape[1])
        if Ti < T:
            if not pad_to_longest:
                raise ValueError(f"fuse_microbatches: padded lengths differ ({Ti} vs {T} frames); pad_to_longest=True fuses them "
                                 "anyway (see the docstring: the convolution's edge frames then see padding instead of the tensor's end)")
            s = torch.nn.functional.pad(s, (0, 0, 0, T - Ti))
        srcs.append(s)
        lens.append(torch.round(wl.double() * Ti) / T)
    return torch.cat(srcs, 0), torch.cat(lens).to(batches[0][1].dtype)


def plan_buckets(total, layer_ranges):
    """The gradient buckets of one step in the order their collectives are launched: the encoder layers' ranges as their
    backward passes finish (last layer first), then what lies in front of the first layer (input projection) and behind the
    last one (final LayerNorm).  Pure function of the flat layout: bench.py --dry-run-ranks prints it without a GPU."""
    ranges = sorted(layer_ranges)
    out = [(a, b, f"layer {len(ranges) - 1 - i}") for i, (a, b) in enumerate(reversed(ranges))]
    if ranges and ranges[0][0] > 0:
        out.append((0, ranges[0][0], "front (parameters registered before the layers)"))
    if ranges and ranges[-1][1] < total:
        out.append((ranges[-1][1], total, "back (parameters registered behind the layers: final LayerNorm, input projection)"))
    if not ranges:
        out.append((0, total, "all"))
    return out


def shard_map(buckets, world):
    """reduce='rs_ag': rank r owns elements [a + r n, a + (r + 1) n), n = (b - a) / world, of every bucket [a, b) - its share of
    the reduce-scattered gradients, of the AdamW moments and of the update.  -> {rank: [(start, end), ...]}"""
    out = {r: [] for r in range(world)}
    for a, b, *_ in buckets:
        if (b - a) % world:
            raise ValueError(f"bucket [{a}, {b}) does not divide over {world} ranks (ALIGN = {ALIGN} elements per parameter)")
        n = (b - a) // world
        for r in range(world):
            out[r].append((a + r * n, a + (r + 1) * n))
    return out


class FlatAdamW:
    def __init__(self, module, lr=8e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01, max_grad_norm=5.0,
                 compute_dtype=torch.bfloat16, process_group=None, buckets=None, reduce="allreduce", grad_dtype=torch.float32):
        if reduce not in ("allreduce", "rs_ag"):
            raise ValueError("FlatAdamW: reduce must be 'allreduce' or 'rs_ag'")
        if grad_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("FlatAdamW: grad_dtype must be torch.float32 or torch.bfloat16")
        self.reduce, self.grad_dtype = reduce, grad_dtype
        self.params = [p for p in module.parameters() if p.requires_grad]
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total, self.offs = total, offs
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(total, dtype=torch.bfloat16, device=dev) if compute_dtype == torch.bfloat16 else None
        for p, o in zip(self.params, offs):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[o:o + n].view(p.shape)
            p.grad = self.flat_g[o:o + n].view(p.shape)
        if self.shadow is not None and dev.type == "cuda":
            ops.L.check(ops.L.lib().smx_cast_from_f32(ops.L.BF16, ops._p(self.flat_p), ops._p(self.shadow), total,
                                                      ops._stream()), "smx_cast_from_f32")
            for p, o in zip(self.params, offs):
                F.register_shadow(p, self.shadow[o:o + p.numel()].view(p.shape))
        self.lr, self.betas, self.eps, self.wd, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.step_count = 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # SMX_FORCE_ALLREDUCE=1 exercises the collective path on a single rank (identity all-reduce) for smoke tests
        self._collective = self.world > 1 or (os.environ.get("SMX_FORCE_ALLREDUCE") == "1" and dist.is_initialized())
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._clip = torch.tensor([1.0, 0.0], dtype=torch.float32, device=dev)   # [clip factor (0 = skip), skipped steps]
        self._pending = []
        self._dev_step = None
        # communication staging: a bf16 image of the gradients (grad_dtype=bfloat16) and, for rs_ag, this rank's shard of the
        # reduced gradients / of the updated weights (bucket [a, b) -> shard elements [a/world, b/world) of these buffers)
        rank = dist.get_rank(process_group) if (dist.is_available() and dist.is_initialized()) else 0
        self.rank = rank
        self._comm = torch.zeros(total, dtype=torch.bfloat16, device=dev) if (self._collective and grad_dtype == torch.bfloat16) else None
        if self._collective and reduce == "rs_ag":
            assert total % self.world == 0, "flat size must divide by the world size (ALIGN = 64 elements)"
            self._shard_g = torch.zeros(total // self.world, dtype=torch.float32, device=dev)
            self._shard_p = torch.zeros(total // self.world, dtype=torch.float32, device=dev)
            self._shard_c = torch.zeros(total // self.world, dtype=torch.bfloat16, device=dev) if self._comm is not None else None
        self._reduced = []            # bucket ranges whose collective was launched in this step (rs_ag: the shards to update)
        self._no_sync = False         # inside no_sync(): bucket hooks launch nothing (gradient accumulation)
        self._shard_layout = None     # rs_ag: the bucket ranges of the first sharded update (= who owns which moment elements)
        self._measure, self._exposed = False, []
        # buckets: list of (start, end) element ranges of the flat buffers, in backward-completion order
        self.buckets = buckets or [(0, total)]
        # in-place weight changes behind the optimizer's back (module.load_state_dict of a checkpoint) must reach the bf16
        # shadows the GEMMs read: they are only refreshed inside smx_adamw_step otherwise
        self._hook = module.register_load_state_dict_post_hook(lambda mod, incompatible: self.refresh_shadows())

    # ---- state ------------------------------------------------------------------------------------
    def refresh_shadows(self):
        """Re-cast the fp32 master weights into the bf16 shadows (call after ANY in-place change of param.data that did
        not go through step(): load_state_dict does it by itself, an EMA / averaging copy_ must call this)."""
        if self.shadow is not None and self.flat_p.is_cuda:
            ops.L.check(ops.L.lib().smx_cast_from_f32(ops.L.BF16, ops._p(self.flat_p), ops._p(self.shadow), self.total,
                                                      ops._stream()), "smx_cast_from_f32")
        F.weights_changed()           # (packed weight images of the panel GEMM are stale now)

    def _full_moments(self):
        """Complete copies of the two AdamW moment buffers.  reduce="rs_ag": every rank updates only its 1/world shard of each
        bucket, so the local buffers are current there and stale elsewhere - the shards are all-gathered bucket by bucket
        (mirroring the weight all-gather of _apply_update) into fresh tensors; a COLLECTIVE in that mode."""
        m, v = self.exp_avg.clone(), self.exp_avg_sq.clone()
        if self._collective and self.reduce == "rs_ag" and self._shard_layout:
            for a, b in self._shard_layout:
                sa, sb = self._shard(a, b)
                for full, local in ((m, self.exp_avg), (v, self.exp_avg_sq)):
                    dist.all_gather_into_tensor(full[a:b], local[sa:sb].contiguous(), group=self.pg)
        return m, v

    def state_dict(self, gather=True):
        """Optimizer state for checkpoint / resume (the weights themselves are in module.state_dict()).

        gather=True (default): the moments are COMPLETE, independent of the gradient exchange mode, and the checkpoint resumes
        under any mode / world size.  With reduce="rs_ag" that makes this call a COLLECTIVE (an all-gather per bucket): EVERY
        rank must make it - the usual `if rank == 0: save(opt.state_dict())` deadlocks; call it on all ranks and let any one
        of them write the result.

        gather=False: no communication.  Under "rs_ag" the result holds only what this rank owns ("owned": its element ranges;
        the moments elsewhere are stale), is marked "complete": False, and load_state_dict accepts it only on the same rank of
        the same world size and bucket layout - the per-rank checkpoint file idiom.  In every other mode it is the same as
        gather=True."""
        step = int(self._dev_step.item()) if self._dev_step is not None else self.step_count
        sd = {"step": step, "skipped_steps": self.skipped_steps(), "total": self.total, "complete": True}
        sharded = bool(self._collective and self.reduce == "rs_ag" and self._shard_layout)
        if gather or not sharded:
            sd["exp_avg"], sd["exp_avg_sq"] = self._full_moments()
        else:
            sd["exp_avg"], sd["exp_avg_sq"] = self.exp_avg.clone(), self.exp_avg_sq.clone()
            sd.update(complete=False, rank=self.rank, world=self.world,
                      owned=[self._shard(a, b) for a, b in self._shard_layout], layout=list(self._shard_layout))
        return sd

    def load_state_dict(self, sd):
        if sd["total"] != self.total:
            raise ValueError(f"FlatAdamW.load_state_dict: flat size {sd['total']} != {self.total} (different model?)")
        if not sd.get("complete", True):
            layout = tuple(tuple(r) for r in sd.get("layout", ()))
            mine = (self.reduce == "rs_ag" and sd.get("rank") == self.rank and sd.get("world") == self.world
                    and self._shard_layout in (None, layout))
            if not mine:
                raise ValueError("FlatAdamW.load_state_dict: a state_dict(gather=False) shard of rank "
                                 f"{sd.get('rank')} / world {sd.get('world')} only resumes on that rank of the same world "
                                 "size and bucket layout under reduce='rs_ag'; save with gather=True (a collective) for a "
                                 "portable checkpoint")
            self._shard_layout = layout        # the next step checks its buckets against the checkpoint's ownership map
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        if self._dev_step is not None:
            self._dev_step.fill_(self.step_count)
        self._clip[1] = float(sd.get("skipped_steps", 0))
        self.refresh_shadows()

    def skipped_steps(self):
        """Steps whose gradient norm was NaN / Inf and whose update was therefore skipped (host read)."""
        return int(self._clip[1].item())

    def _rebind_grads(self):
        """param.grad must be a view of flat_g.  torch's module.zero_grad() (set_to_none=True) detaches it and the kernels
        then accumulate into freshly allocated tensors: fold those back (single rank), or fail loudly when buckets may
        already have been all-reduced without them."""
        base = self.flat_g.data_ptr()
        for p, o in zip(self.params, self.offs):
            g = p.grad
            if g is not None and g.data_ptr() == base + 4 * o:
                continue
            if self._collective:
                raise RuntimeError("FlatAdamW: a parameter's .grad no longer points into the flat gradient buffer "
                                   "(module.zero_grad(set_to_none=True)?) - use optimizer.zero_grad(); the gradient buckets "
                                   "were all-reduced without it")
            view = self.flat_g[o:o + p.numel()].view(p.shape)
            if g is None:
                view.zero_()
            else:
                view.copy_(g)
            p.grad = view

    # ---- bucket plumbing ---------------------------------------------------------------------------
    def param_range(self, params):
        """Element range of the flat buffers covering `params` (must be contiguous in registration order)."""
        ids = {id(p) for p in params}
        idx = [i for i, p in enumerate(self.params) if id(p) in ids]
        lo, hi = min(idx), max(idx)
        assert hi - lo + 1 == len(idx), "bucket parameters must be contiguous in the flat buffer"
        end = self.offs[hi + 1] if hi + 1 < len(self.offs) else self.total
        return self.offs[lo], end

    def _shard(self, start, end):
        """This rank's 1/world piece of bucket [start, end): (first element, end element) in the flat buffers."""
        n = (end - start) // self.world
        return start + self.rank * n, start + (self.rank + 1) * n

    def reduce_bucket_async(self, start, end):
        """Launch the collective of one gradient bucket (called right after its producer's backward): an all-reduce, or -
        reduce="rs_ag" - a reduce-scatter that leaves this rank's 1/world shard of the summed bucket in the shard buffer."""
        if not self._collective or self._no_sync:
            return
        W = self.world
        assert (end - start) % W == 0, "bucket length must divide by the world size"
        src = self.flat_g[start:end]
        if self._comm is not None:                        # bf16 on the wire
            src = self._comm[start:end]
            self._k_cast(self.flat_g[start:end], src)
        if self.reduce == "allreduce":
            work = dist.all_reduce(src, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        else:
            dst = (self._shard_c if self._comm is not None else self._shard_g)[start // W:end // W]
            work = dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._pending.append(work)
        self._reduced.append((start, end))

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation under data parallelism (the role of torch DDP's no_sync; the reference's fit_batch wraps the
        first grad_accumulation_factor - 1 micro-batches in it, core.py): inside, the bucket hooks launch NO collective - the
        kernels keep adding into the flat gradient buffer - and the micro-batch that runs outside reduces the accumulated sums.

            opt.zero_grad()
            with opt.no_sync():
                for x in micro[:-1]: model(x).backward(...)
            model(micro[-1]).backward(...)      # bucket hooks fire here
            opt.step()"""
        prev, self._no_sync = self._no_sync, True
        try:
            yield
        finally:
            self._no_sync = prev

    def _finish_reduce(self):
        """Wait for the launched collectives (the compute stream waits, not the host) and bring bf16 results back to fp32."""
        ev = None
        if self._measure and self.flat_p.is_cuda and not torch.cuda.is_current_stream_capturing():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._pending:
            w.wait()
        self._pending = []
        if self._comm is not None:
            W = self.world
            for a, b in self._reduced:
                if self.reduce == "allreduce":
                    self._k_cast(self._comm[a:b], self.flat_g[a:b])
                else:
                    self._k_cast(self._shard_c[a // W:b // W], self._shard_g[a // W:b // W])
        if ev is not None:
            ev[1].record()
            self._exposed.append(ev)
            if len(self._exposed) > 256:                   # (a long run keeps the most recent steps only)
                del self._exposed[:128]

    def measure_comm(self, enable=True):
        """Record, from now on, how long the compute stream waits for the gradient collectives in every step."""
        self._measure, self._exposed = enable, []

    def comm_exposed_ms(self):
        """Mean time per step the compute stream stood waiting for the collectives (+ the bf16 casts back) since
        measure_comm(True): communication NOT hidden behind the backward pass.  Host read (synchronises)."""
        if not self._exposed:
            return 0.0
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)

    def zero_grad(self):
        self.flat_g.zero_()

    def use_device_step_counter(self, enable=True):
        """Keep the step count (AdamW bias correction, dropout epoch) in device memory so that a whole training step
        can be captured once in a hipGraph (``torch.cuda.graph``) and replayed: no kernel argument changes between
        steps, the counter does (include/smx.h: smx_step_counter_add)."""
        if enable:
            if self._dev_step is None:
                self._dev_step = torch.full((1,), self.step_count, dtype=torch.int64, device=self.flat_p.device)
            ops.set_step_counter(self._dev_step)
        else:
            ops.set_step_counter(None)
            self._dev_step = None

    def step(self, reduce_all=False):
        F.flush_deferred()          # (no-op unless a backward ran outside functional.block)
        if not (self.flat_p.is_cuda and torch.cuda.is_current_stream_capturing()):
            self._rebind_grads()
        if self._collective:
            if reduce_all:
                self.reduce_bucket_async(0, self.total)
            self._finish_reduce()
        self.step_count += 1
        if self._dev_step is not None:
            ops.step_counter_add(self._dev_step, 1)
        self._apply_update(1.0 / self.world)
        self._reduced = []

    # ---- hipGraph under data parallelism: the collective stays OUTSIDE the graphs -------------------------------------
    def all_reduce_all(self):
        """ONE all-reduce of the whole flat gradient buffer, waited for on the current stream (used between the two
        captured halves of a step: forward + backward | update)."""
        if self._collective:
            self._reduced = []                            # (everything is reduced here: earlier entries are stale)
            self.reduce_bucket_async(0, self.total)
            self._finish_reduce()

    def replay(self, graph):
        """Replay a captured step that contains this optimizer's update.  The replay rewrites the bf16 shadows without running any
        Python: the packed weight images of the panel GEMM (functional.wpacked) must be re-packed by the next EAGER forward."""
        graph.replay()
        F.weights_changed()

    def update_only(self):
        """The part of step() behind the collective (capturable: device step counter, clip, AdamW, shadows)."""
        assert self._dev_step is not None, "call use_device_step_counter(True) before capturing"
        assert not (self._collective and self.reduce == "rs_ag"), "update_only (graph capture) needs reduce='allreduce'"
        ops.step_counter_add(self._dev_step, 1)
        self._apply_update(1.0 / self.world)

    # ---- the update: device kernels behind three primitives (the gloo tests substitute CPU restatements) -------------
    def _k_cast(self, src, dst):
        """dst <- src between float32 and bfloat16 (smx_cast_from_f32 / smx_cast_to_f32)."""
        lib, n = ops.L.lib(), src.numel()
        if src.dtype == torch.float32:
            ops.L.check(lib.smx_cast_from_f32(ops.L.BF16, ops._p(src), ops._p(dst), n, ops._stream()), "smx_cast_from_f32")
        else:
            ops.L.check(lib.smx_cast_to_f32(ops.L.BF16, ops._p(src), ops._p(dst), n, ops._stream()), "smx_cast_to_f32")

    def _k_sumsq(self, g):
        ops.sumsq(g, self._sumsq)                         # _sumsq[0] += sum(g^2), fixed summation order

    def _k_clip(self, gscale):
        ops.clip_factor(self._sumsq, self.max_grad_norm, gscale, self._clip)

    def _k_adamw(self, a, b, g, gscale, clip):
        """AdamW on flat elements [a, b) with the gradient tensor g (b - a elements)."""
        F.weights_changed()           # (the bf16 shadows / fp32 biases change in place: packed weight images are stale)
        ops.adamw_step(self.flat_p[a:b], g, self.exp_avg[a:b], self.exp_avg_sq[a:b],
                       self.shadow[a:b] if self.shadow is not None else None, self.lr, self.betas[0], self.betas[1], self.eps,
                       self.wd, 0 if self._dev_step is not None else self.step_count, gscale, clip)

    def _apply_update(self, gscale):
        """Global-norm clip + AdamW + bf16 shadow refresh: three HIP kernel launches per owned range, nothing visits the
        host.  reduce="rs_ag": the ranges are this rank's shards of the reduce-scattered buckets, the squared norm is
        completed by one scalar all-reduce and the updated weights are all-gathered bucket by bucket."""
        sharded = self._collective and self.reduce == "rs_ag"
        W = self.world
        if sharded:
            covered = sum(b - a for a, b in self._reduced)
            assert covered == self.total, f"rs_ag: the reduced buckets cover {covered} of {self.total} elements"
            # the bucket ranges decide which rank owns which element's moments: they must not change between steps
            layout = tuple(sorted(self._reduced))
            if self._shard_layout is None:
                self._shard_layout = layout
            elif layout != self._shard_layout:
                raise RuntimeError("FlatAdamW(reduce='rs_ag'): the gradient buckets changed between steps - the AdamW moments are "
                                   "sharded by bucket and would be mixed up (keep ONE bucket plan, or reduce='allreduce')")
            work = [(self._shard(a, b), self._shard_g[a // W:b // W]) for a, b in self._reduced]
        else:
            work = [((0, self.total), self.flat_g)]
        clip = None
        if self.max_grad_norm is not None and self.max_grad_norm > 0:
            self._sumsq.zero_()
            for _, g in work:
                self._k_sumsq(g)
            if sharded:
                dist.all_reduce(self._sumsq, op=dist.ReduceOp.SUM, group=self.pg)
            self._k_clip(gscale)
            clip = self._clip
        for (a, b), g in work:
            self._k_adamw(a, b, g, gscale, clip)
        if sharded:
            for a, b in self._reduced:
                sa, sb = self._shard(a, b)
                mine = self._shard_p[a // W:b // W]
                mine.copy_(self.flat_p[sa:sb])
                dist.all_gather_into_tensor(self.flat_p[a:b], mine, group=self.pg)
            self.refresh_shadows()

    def grad_norm(self):
        """Host read of the last global gradient norm (diagnostics only)."""
        return float(self._sumsq.sqrt().item()) / self.world
