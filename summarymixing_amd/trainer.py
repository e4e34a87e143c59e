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
"""
import os

import torch
import torch.distributed as dist

from . import functional as F
from . import ops

ALIGN = 64   # elements; keeps every parameter view 256-byte aligned in fp32 and 128-byte in bf16


class FlatAdamW:
    def __init__(self, module, lr=8e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01, max_grad_norm=5.0,
                 compute_dtype=torch.bfloat16, process_group=None, buckets=None):
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

    def state_dict(self):
        """Optimizer state for checkpoint / resume (the weights themselves are in module.state_dict())."""
        step = int(self._dev_step.item()) if self._dev_step is not None else self.step_count
        return {"step": step, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "skipped_steps": self.skipped_steps(), "total": self.total}

    def load_state_dict(self, sd):
        if sd["total"] != self.total:
            raise ValueError(f"FlatAdamW.load_state_dict: flat size {sd['total']} != {self.total} (different model?)")
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

    def reduce_bucket_async(self, start, end):
        """Launch the all-reduce of one gradient bucket (called right after its producer's backward)."""
        if self._collective:
            self._pending.append(dist.all_reduce(self.flat_g[start:end], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def zero_grad(self):
        self.flat_g.zero_()

    def use_device_step_counter(self, enable=True):
        """Keep the step count (AdamW bias correction, dropout epoch) in device memory so that a whole training step
        can be captured once in a hipGraph (``torch.cuda.graph``) and replayed: no kernel argument changes between
        steps, the counter does (include/smx.h: smx_set_step_counter)."""
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
            for w in self._pending:
                w.wait()
            self._pending = []
        self.step_count += 1
        if self._dev_step is not None:
            ops.step_counter_add(self._dev_step, 1)
        self._apply_update(1.0 / self.world)

    # ---- hipGraph under data parallelism: the collective stays OUTSIDE the graphs -------------------------------------
    def all_reduce_all(self):
        """ONE all-reduce of the whole flat gradient buffer, waited for on the current stream (used between the two
        captured halves of a step: forward + backward | update)."""
        if self._collective:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.pg)

    def update_only(self):
        """The part of step() behind the collective (capturable: device step counter, clip, AdamW, shadows)."""
        assert self._dev_step is not None, "call use_device_step_counter(True) before capturing"
        ops.step_counter_add(self._dev_step, 1)
        self._apply_update(1.0 / self.world)

    def _apply_update(self, gscale):
        """Global-norm clip + AdamW + bf16 shadow refresh: three HIP kernel launches, nothing visits the host."""
        clip = None
        if self.max_grad_norm is not None and self.max_grad_norm > 0:
            self._sumsq.zero_()
            ops.sumsq(self.flat_g, self._sumsq)
            ops.clip_factor(self._sumsq, self.max_grad_norm, gscale, self._clip)
            clip = self._clip
        ops.adamw_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.shadow, self.lr, self.betas[0],
                       self.betas[1], self.eps, self.wd, 0 if self._dev_step is not None else self.step_count, gscale, clip)

    def grad_norm(self):
        """Host read of the last global gradient norm (diagnostics only)."""
        return float(self._sumsq.sqrt().item()) / self.world
