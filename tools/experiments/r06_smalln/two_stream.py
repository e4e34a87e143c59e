"""Probe: does running two half-batches CONCURRENTLY (two streams, two captured graphs) beat one launch chain over the whole batch?
Two independent encoders + optimizers (B/2 each) replayed side by side against one encoder at B.  The library's workspaces are shared
between the two graphs, so the VALUES of the concurrent run are garbage - this measures time only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import bench
from summarymixing_amd.trainer import FlatAdamW

dev = torch.device("cuda:0")
dtype = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfgname = sys.argv[2] if len(sys.argv) > 2 else "c2b"


def make(b, seed):
    cfg = dict(bench.CONFIGS[cfgname]); cfg["B"] = b
    enc = bench.build_encoder(cfg, dev, 0.15)
    opt = FlatAdamW(enc, lr=8e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.01, max_grad_norm=5.0, compute_dtype=dtype)
    src, wl, r, _ = bench.synthetic_batch(cfg, seed, dev, dtype)
    def step():
        opt.zero_grad()
        enc(src, wl).backward(r)
        opt.step()
    return step


def capture(step, stream):
    with torch.cuda.stream(stream):
        for _ in range(3):
            step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        step()
    torch.cuda.synchronize()
    return g


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
full = make(B, 0)
gF = capture(full, sA)
def run_full():
    with torch.cuda.stream(sA):
        gF.replay()
print(f"one chain, B = {B}: graph {timeit(run_full):7.3f} ms   eager {timeit(lambda: full()):7.3f} ms")
hA, hB = make(B // 2, 1), make(B // 2, 2)
gA, gB = capture(hA, sA), capture(hB, sB)
def seq():
    with torch.cuda.stream(sA):
        gA.replay(); gB.replay()
def conc():
    with torch.cuda.stream(sA):
        gA.replay()
    with torch.cuda.stream(sB):
        gB.replay()
print(f"two halves of {B // 2}: one after the other {timeit(seq):7.3f} ms   side by side on two streams {timeit(conc):7.3f} ms")
