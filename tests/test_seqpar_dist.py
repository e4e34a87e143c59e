"""World-size-2 gloo tests (CPU) of the sequence-parallel exchange steps (summarymixing_amd/sequence_parallel.py):
the (sums | counts) all-reduce behind the cell's per-utterance mean, the conv halo exchange and its transposed exchange in
the backward, shard() and reduce_gradients().  The HIP kernels cannot run here: the per-frame arithmetic around the
exchanges is plain torch on CPU; what is under test is the N > 1 plumbing (tests/test_seqpar_gpu.py runs the real encoder
sharded over two ranks on the GPU box)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as Fn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from summarymixing_amd import sequence_parallel as SP
        assert not SP.enabled()
        g = torch.Generator().manual_seed(11)
        B, T, d, k = 3, 40, 6, 31
        H = (k - 1) // 2
        x = torch.randn(B, T, d, generator=g)
        r = torch.randn(B, T, d, generator=g)
        w = torch.randn(d, 1, k, generator=g)
        lens = torch.tensor([T, 12, 33])
        m = (torch.arange(T)[None] < lens[:, None])
        with SP.sequence_parallel():
            assert SP.enabled() and SP.world() == world and SP.rank() == rank
            Tl = T // world
            xl, ml = SP.shard(x), SP.shard(m)
            assert torch.equal(xl, x[:, rank * Tl:(rank + 1) * Tl])
            # --- the cell's mean: local sums + counts, one all-reduce ----------------------------------------------
            buf = torch.cat([(xl * ml[..., None]).sum(1), ml.sum(1, dtype=torch.float32)[:, None]], 1)
            SP.all_reduce_sum(buf)
            mean = buf[:, :d] / buf[:, d:]
            ref = (x * m[..., None]).sum(1) / m.sum(1, keepdim=True)
            assert torch.allclose(mean, ref, atol=1e-6)
            # --- depthwise conv with halos, forward and backward ------------------------------------------------------
            xf = x.clone().requires_grad_(True)
            yf = Fn.conv1d(xf.transpose(1, 2), w, padding=H, groups=d).transpose(1, 2)
            (yf * r).sum().backward()
            xl = xl.clone().requires_grad_(True)
            lh, rh = SP.exchange_halos(xl.detach()[:, :H], xl.detach()[:, Tl - H:])
            if rank == 0:
                assert lh.abs().max() == 0
            if rank == world - 1:
                assert rh.abs().max() == 0
            xe = torch.cat([lh, xl.detach(), rh], 1).requires_grad_(True)
            ye = Fn.conv1d(xe.transpose(1, 2), w, padding=H, groups=d).transpose(1, 2)[:, H:H + Tl]
            assert torch.allclose(ye, SP.shard(yf.detach()), atol=1e-5)
            (ye * SP.shard(r)).sum().backward()
            g_first, g_last = SP.return_halo_grads(xe.grad[:, :H], xe.grad[:, H + Tl:])
            dx = xe.grad[:, H:H + Tl].clone()
            dx[:, :H] += g_first
            dx[:, Tl - H:] += g_last
            assert torch.allclose(dx, SP.shard(xf.grad), atol=1e-5)
            # --- partial parameter gradients are SUMMED over the group -----------------------------------------------
            p = torch.nn.Parameter(torch.zeros(4, 3))
            q = torch.nn.Parameter(torch.zeros(5))
            p.grad = torch.full((4, 3), float(rank + 1))
            q.grad = torch.arange(5.0) * (rank + 1)
            SP.reduce_gradients([p, q])
            tot = sum(range(1, world + 1))
            assert torch.equal(p.grad, torch.full((4, 3), float(tot))) and torch.equal(q.grad, torch.arange(5.0) * tot)
        assert not SP.enabled()
        out.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        out.put((rank, "FAIL " + repr(e) + "\n" + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_sequence_parallel_exchanges_world2():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_shard_requires_divisible_length():
    import pytest
    from summarymixing_amd import sequence_parallel as SP
    SP._State.world, SP._State.rank = 3, 0
    try:
        with pytest.raises(ValueError):
            SP.shard(torch.zeros(2, 10, 4))
    finally:
        SP._State.world, SP._State.rank = 1, 0
