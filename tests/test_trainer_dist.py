"""World-size-2 gloo tests (CPU) of the data-parallel plumbing in summarymixing_amd.trainer.FlatAdamW:
flat-buffer views, per-layer gradient buckets reduced asynchronously, 1/world scaling, and gradient equivalence
(all-reduced shard gradients == single-process gradients on the concatenated batch).  The HIP kernels cannot run
here, so the per-rank gradients come from the CPU oracle and the final parameter update is a CPU restatement of
smx_adamw_step; what is under test is the N > 1 path: bucketing + RCCL/gloo all-reduce + scaling."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import smx_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(d=16, layers=3):
    torch.manual_seed(0)
    from summarymixing_amd.lobes.models.transformer.Conformer import ConformerEncoder
    enc = ConformerEncoder(layers, d, 32, 2, kernel_size=5, activation="swish", dropout=0.0, attention_type="SummaryMixing",
                           local_proj_hid_dim=[d], local_proj_out_dim=d, summary_hid_dim=[d], mode="SummaryMixing-fast")
    with torch.no_grad():
        for p in enc.parameters():
            p.normal_(0, 0.2)
    return enc


def _oracle_grads_into(enc, x, pad, r):
    """Accumulate d(sum(y*r))/dparams of the oracle encoder into enc's .grad views."""
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    y = O.conformer_encoder(x, sd, "", "swish", "SummaryMixing-fast", x.shape[-1], None, pad)
    (y * r).sum().backward()
    for k, p in enc.named_parameters():
        p.grad.add_(sd[k].grad)


class _CpuUpdate:
    """CPU restatement of smx_adamw_step (decoupled weight decay, bias correction) for the gloo test."""

    def _apply_update(self, gscale):
        g = self.flat_g * gscale
        if self.max_grad_norm:
            nrm = g.norm()
            g = g * min(1.0, self.max_grad_norm / (float(nrm) + 1e-6))
        b1, b2 = self.betas
        self.flat_p.mul_(1 - self.lr * self.wd)
        self.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** self.step_count, 1 - b2 ** self.step_count
        self.flat_p.addcdiv_(self.exp_avg, self.exp_avg_sq.sqrt() / bc2 ** 0.5 + self.eps, value=-self.lr / bc1)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from summarymixing_amd.trainer import FlatAdamW

        class Opt(_CpuUpdate, FlatAdamW):
            pass
        enc = _model()
        opt = Opt(enc, lr=1e-2, max_grad_norm=5.0, compute_dtype=torch.float32)
        # per-layer buckets, launched in backward order like bench.py does through the block hooks
        ranges = [opt.param_range(list(l.parameters())) for l in enc.layers]
        assert ranges[0][0] == 0 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        g = torch.Generator().manual_seed(5)
        X = torch.randn(4, 9, 16, generator=g)
        R = torch.randn(4, 9, 16, generator=g)
        lens = torch.tensor([9, 5, 7, 9])
        pad = torch.arange(9)[None] < lens[:, None]
        sl = slice(rank * 2, rank * 2 + 2)                       # utterance shard of this rank
        opt.zero_grad()
        _oracle_grads_into(enc, X[sl], pad[sl], R[sl])
        for a, b in reversed(ranges):
            opt.reduce_bucket_async(a, b)
        opt.reduce_bucket_async(ranges[-1][1], opt.total)        # tail: final LayerNorm
        for w in opt._pending:
            w.wait()
        summed = opt.flat_g.clone()
        opt.step()
        if rank == 0:
            torch.save({"summed": summed, "params": opt.flat_p.clone(), "ranges": ranges, "total": opt.total}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp2_gradient_equivalence_and_update(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    # single process on the concatenated batch
    from summarymixing_amd.trainer import FlatAdamW

    class Opt(_CpuUpdate, FlatAdamW):
        pass
    enc = _model()
    opt = Opt(enc, lr=1e-2, max_grad_norm=5.0, compute_dtype=torch.float32)
    g = torch.Generator().manual_seed(5)
    X = torch.randn(4, 9, 16, generator=g)
    R = torch.randn(4, 9, 16, generator=g)
    lens = torch.tensor([9, 5, 7, 9])
    pad = torch.arange(9)[None] < lens[:, None]
    opt.zero_grad()
    _oracle_grads_into(enc, X, pad, R)
    torch.testing.assert_close(got["summed"], opt.flat_g, rtol=1e-5, atol=1e-6)      # sum over ranks == full batch
    # the DP step divides by world (mean over ranks); emulate by scaling the single-process gradient
    opt.flat_g.mul_(0.5)
    opt.step()
    # Adam divides by sqrt(v): elements with g ~ 0 amplify fp32 summation-order noise of the all-reduce
    torch.testing.assert_close(got["params"], opt.flat_p, rtol=1e-4, atol=5e-5)


def _worker_accum(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from summarymixing_amd.trainer import FlatAdamW

        class Opt(_CpuUpdate, FlatAdamW):
            pass
        enc = _model()
        opt = Opt(enc, lr=1e-2, max_grad_norm=5.0, compute_dtype=torch.float32)
        ranges = [opt.param_range(list(l.parameters())) for l in enc.layers]
        g = torch.Generator().manual_seed(5)
        X = torch.randn(4, 9, 16, generator=g)
        R = torch.randn(4, 9, 16, generator=g)
        lens = torch.tensor([9, 5, 7, 9])
        pad = torch.arange(9)[None] < lens[:, None]

        def hooks():                                              # what the block hooks + bench.py's tail do per backward pass
            for a, b in reversed(ranges):
                opt.reduce_bucket_async(a, b)
            opt.reduce_bucket_async(ranges[-1][1], opt.total)
        opt.zero_grad()
        with opt.no_sync():                                       # micro-batch 1 of this rank's shard: nothing on the wire
            i = rank * 2
            _oracle_grads_into(enc, X[i:i + 1], pad[i:i + 1], R[i:i + 1])
            hooks()
            assert not opt._pending and not opt._reduced
        _oracle_grads_into(enc, X[i + 1:i + 2], pad[i + 1:i + 2], R[i + 1:i + 2])
        hooks()                                                   # micro-batch 2: the accumulated sums are reduced
        assert len(opt._pending) == len(ranges) + 1
        for w in opt._pending:
            w.wait()
        summed = opt.flat_g.clone()
        opt.step()
        if rank == 0:
            torch.save({"summed": summed, "params": opt.flat_p.clone()}, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp2_gradient_accumulation_under_no_sync(tmp_path):
    """FlatAdamW.no_sync(): two micro-batches per rank, collectives only behind the second; the reduced gradients are the
    full-batch gradients (the role of DDP's no_sync in the reference's fit_batch with grad_accumulation_factor > 1)."""
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker_accum, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    from summarymixing_amd.trainer import FlatAdamW

    class Opt(_CpuUpdate, FlatAdamW):
        pass
    enc = _model()
    opt = Opt(enc, lr=1e-2, max_grad_norm=5.0, compute_dtype=torch.float32)
    g = torch.Generator().manual_seed(5)
    X = torch.randn(4, 9, 16, generator=g)
    R = torch.randn(4, 9, 16, generator=g)
    lens = torch.tensor([9, 5, 7, 9])
    pad = torch.arange(9)[None] < lens[:, None]
    opt.zero_grad()
    for i in range(4):           # (one utterance at a time: the conv module's edge frames see the padding of their own micro-batch)
        _oracle_grads_into(enc, X[i:i + 1], pad[i:i + 1], R[i:i + 1])
    torch.testing.assert_close(got["summed"], opt.flat_g, rtol=1e-5, atol=1e-6)
    opt.flat_g.mul_(0.5)
    opt.step()
    torch.testing.assert_close(got["params"], opt.flat_p, rtol=1e-4, atol=5e-5)


def test_flat_views_alias_parameters():
    from summarymixing_amd.trainer import FlatAdamW

    class Opt(_CpuUpdate, FlatAdamW):
        pass
    enc = _model(layers=1)
    before = {k: v.clone() for k, v in enc.state_dict().items()}
    opt = Opt(enc, compute_dtype=torch.float32)
    for k, v in enc.state_dict().items():
        torch.testing.assert_close(v, before[k])
    p = next(enc.parameters())
    p.grad.fill_(1.0)
    assert opt.flat_g[: p.numel()].eq(1).all() and p.data_ptr() == opt.flat_p.data_ptr()
    assert opt.total % 64 == 0 and all(o % 64 == 0 for o in opt.offs)


# ---- reduce="rs_ag" and grad_dtype=bfloat16 (world 2, gloo) ------------------------------------------------------------
class _CpuKernels:
    """CPU restatements of the three device primitives of FlatAdamW._apply_update (smx_cast_*, smx_sumsq, smx_clip_factor,
    smx_adamw_step), so that the sharded update runs through the product's own control flow here."""

    def _k_cast(self, src, dst):
        dst.copy_(src)

    def _k_sumsq(self, g):
        self._sumsq += (g.double() ** 2).sum().float()

    def _k_clip(self, gscale):
        nrm = float(self._sumsq.sqrt()) * gscale
        self._clip[0] = min(1.0, self.max_grad_norm / (nrm + 1e-6))

    def _k_adamw(self, a, b, g, gscale, clip):
        g = g * gscale * (float(clip[0]) if clip is not None else 1.0)
        b1, b2 = self.betas
        p, m, v = self.flat_p[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b]
        p.mul_(1 - self.lr * self.wd)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** self.step_count, 1 - b2 ** self.step_count
        p.addcdiv_(m, v.sqrt() / bc2 ** 0.5 + self.eps, value=-self.lr / bc1)


def _worker_modes(rank, world, port, out, reduce, grad_dtype):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from summarymixing_amd.trainer import FlatAdamW

        class Opt(_CpuKernels, FlatAdamW):
            pass
        enc = _model()
        opt = Opt(enc, lr=1e-2, max_grad_norm=0.5, compute_dtype=torch.float32, reduce=reduce, grad_dtype=grad_dtype)
        ranges = [opt.param_range(list(l.parameters())) for l in enc.layers]
        g = torch.Generator().manual_seed(5)
        X = torch.randn(4, 9, 16, generator=g)
        R = torch.randn(4, 9, 16, generator=g)
        lens = torch.tensor([9, 5, 7, 9])
        pad = torch.arange(9)[None] < lens[:, None]
        sl = slice(rank * 2, rank * 2 + 2)
        for _ in range(2):                                       # two steps: the moments and the gathered weights carry over
            opt.zero_grad()
            _oracle_grads_into(enc, X[sl], pad[sl], R[sl])
            for a, b in reversed(ranges):
                opt.reduce_bucket_async(a, b)
            opt.reduce_bucket_async(ranges[-1][1], opt.total)
            opt.step()
        torch.save({"params": opt.flat_p.clone(), "clip": opt._clip.clone(), "flat_g": opt.flat_g.clone(),
                    "shard_g": opt._shard_g.clone() if reduce == "rs_ag" else None,
                    "ranges": list(reversed(ranges)) + [(ranges[-1][1], opt.total)]}, out + f".{rank}")
    finally:
        dist.destroy_process_group()


def _single_process_reference(steps=2):
    from summarymixing_amd.trainer import FlatAdamW

    class Opt(_CpuKernels, FlatAdamW):
        pass
    enc = _model()
    opt = Opt(enc, lr=1e-2, max_grad_norm=0.5, compute_dtype=torch.float32)
    g = torch.Generator().manual_seed(5)
    X = torch.randn(4, 9, 16, generator=g)
    R = torch.randn(4, 9, 16, generator=g)
    lens = torch.tensor([9, 5, 7, 9])
    pad = torch.arange(9)[None] < lens[:, None]
    for _ in range(steps):
        opt.zero_grad()
        _oracle_grads_into(enc, X, pad, R)
        opt.flat_g.mul_(0.5)                                     # the DP step averages over the 2 ranks
        opt.step()
    return opt


@pytest.mark.timeout(300)
@pytest.mark.parametrize("reduce,grad_dtype,tol", [("rs_ag", torch.float32, 1e-3), ("allreduce", torch.bfloat16, 5e-2),
                                                   ("rs_ag", torch.bfloat16, 5e-2)])
def test_dp2_reduce_modes_match_single_process(tmp_path, reduce, grad_dtype, tol):
    """reduce-scatter + sharded AdamW + all-gather (and gradients in bf16 on the wire) give the weights of the
    single-process step on the concatenated batch; both ranks end with IDENTICAL weights."""
    out = str(tmp_path / "r")
    mp.spawn(_worker_modes, args=(2, _free_port(), out, reduce, grad_dtype), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["params"], r1["params"])               # ranks never drift apart
    assert float(r0["clip"][0]) < 1.0                            # the clip was active (norm completed across the shards)
    ref = _single_process_reference()
    # (1) the gradients the update saw: sum over the ranks (= 2 x the single-process mean gradient), exact in fp32,
    #     within three bf16 roundings (each rank's cast, the sum) when they cross the wire in bf16
    if reduce == "rs_ag":
        got_g = torch.zeros_like(ref.flat_g)
        for a, b in r0["ranges"]:
            n = (b - a) // 2
            got_g[a:a + n] = r0["shard_g"][a // 2:b // 2]
            got_g[a + n:b] = r1["shard_g"][a // 2:b // 2]
    else:
        got_g = r0["flat_g"]
    want_g = ref.flat_g * 2
    gerr = (got_g - want_g).abs().max().item() / want_g.abs().max().item()
    assert gerr <= (1e-5 if grad_dtype == torch.float32 else 2 ** -7), gerr
    # (2) the UPDATE (two AdamW steps of size ~lr).  Adam normalises every element by its own sqrt(v), so elements whose
    #     gradient is ~0 turn bf16 rounding into sign flips: RMS over the whole vector for bf16, max-abs for fp32
    base = FlatAdamWBase()
    du, dr = r0["params"] - base, ref.flat_p - base
    if grad_dtype == torch.float32:
        err = (du - dr).abs().max().item() / dr.abs().max().item()
    else:
        err = ((du - dr).pow(2).mean().sqrt() / dr.pow(2).mean().sqrt()).item()
    assert err <= tol, err


def FlatAdamWBase():
    """The initial flat weights (same seed as every worker)."""
    from summarymixing_amd.trainer import FlatAdamW

    class Opt(_CpuKernels, FlatAdamW):
        pass
    return Opt(_model(), compute_dtype=torch.float32).flat_p.clone()


# ---- checkpoint / resume under reduce="rs_ag": the saved moments are complete (round-3 ADVICE) ---------------------------------
def _worker_resume(rank, world, port, out, phase):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from summarymixing_amd.trainer import FlatAdamW

        class Opt(_CpuKernels, FlatAdamW):
            pass
        enc = _model()
        opt = Opt(enc, lr=1e-2, max_grad_norm=0.5, compute_dtype=torch.float32, reduce="rs_ag")
        ranges = [opt.param_range(list(l.parameters())) for l in enc.layers]
        g = torch.Generator().manual_seed(5)
        X = torch.randn(4, 9, 16, generator=g)
        R = torch.randn(4, 9, 16, generator=g)
        lens = torch.tensor([9, 5, 7, 9])
        pad = torch.arange(9)[None] < lens[:, None]
        sl = slice(rank * 2, rank * 2 + 2)

        def one_step():
            opt.zero_grad()
            _oracle_grads_into(enc, X[sl], pad[sl], R[sl])
            for a, b in reversed(ranges):
                opt.reduce_bucket_async(a, b)
            opt.reduce_bucket_async(ranges[-1][1], opt.total)
            opt.step()

        if phase == "straight":                                  # three steps in one go
            for _ in range(3):
                one_step()
        elif phase == "save":                                    # two steps, then a checkpoint written by RANK 0 ONLY
            for _ in range(2):
                one_step()
            sd = opt.state_dict()                                # (collective: both ranks call it)
            if rank == 0:
                torch.save({"opt": sd, "model": enc.state_dict()}, out + ".ckpt")
        elif phase == "save_local":                              # two steps, then one file PER RANK, no communication
            for _ in range(2):
                one_step()
            sd = opt.state_dict(gather=False)
            assert not sd["complete"] and sd["rank"] == rank and sum(b - a for a, b in sd["owned"]) == opt.total // world
            torch.save({"opt": sd, "model": enc.state_dict()}, out + f".ckpt{rank}")
        elif phase == "resume_local":
            ck = torch.load(out + f".ckpt{rank}")
            other = torch.load(out + f".ckpt{1 - rank}")
            with pytest.raises(ValueError, match="only resumes on that rank"):
                opt.load_state_dict(other["opt"])
            enc.load_state_dict(ck["model"])
            opt.load_state_dict(ck["opt"])
            one_step()
        else:                                                    # fresh processes resume from rank 0's file, one more step
            ck = torch.load(out + ".ckpt")
            enc.load_state_dict(ck["model"])
            opt.load_state_dict(ck["opt"])
            one_step()
        full = opt.state_dict()
        torch.save({"params": opt.flat_p.clone(), "exp_avg": full["exp_avg"], "exp_avg_sq": full["exp_avg_sq"]},
                   out + f".{phase}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_dp2_rs_ag_checkpoint_resume(tmp_path):
    """A checkpoint saved from rank 0 under reduce='rs_ag' holds the COMPLETE AdamW moments (each rank only updates its
    shard): resuming from it and taking one more step gives bit-identical weights and moments to three uninterrupted steps."""
    out = str(tmp_path / "r")
    for phase in ("straight", "save", "resume"):
        mp.spawn(_worker_resume, args=(2, _free_port(), out, phase), nprocs=2, join=True)
    a0, b0, b1 = torch.load(out + ".straight.0"), torch.load(out + ".resume.0"), torch.load(out + ".resume.1")
    assert torch.equal(b0["params"], b1["params"])
    for k in ("params", "exp_avg", "exp_avg_sq"):
        assert torch.equal(a0[k], b0[k]), k
    # the saved moments are complete: no half of any bucket is still zero
    ck = torch.load(out + ".ckpt")["opt"]
    nz = (ck["exp_avg_sq"] != 0).float().mean().item()
    full = (a0["exp_avg_sq"] != 0).float().mean().item()
    assert nz > 0.9 * full, (nz, full)
    # the non-collective per-rank idiom (state_dict(gather=False)): same trajectory, and a foreign shard is refused
    for phase in ("save_local", "resume_local"):
        mp.spawn(_worker_resume, args=(2, _free_port(), out, phase), nprocs=2, join=True)
    c0, c1 = torch.load(out + ".resume_local.0"), torch.load(out + ".resume_local.1")
    assert torch.equal(c0["params"], c1["params"])
    for k in ("params", "exp_avg", "exp_avg_sq"):
        assert torch.equal(a0[k], c0[k]), k
