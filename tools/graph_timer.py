"""Launch timing below the host's eager launch floor (~10 us per ctypes call): capture `reps` launches of fn into ONE hipGraph and time
replays with HIP events - what a kernel costs inside bench.py's replayed step (the small-batch regime), back to back with itself."""
import torch


def graph_us(fn, reps=40, replays=6):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(reps):
                fn()
        g.replay()
        s.synchronize()
        ts = []
        for _ in range(replays):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            g.replay()
            e1.record(s)
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    torch.cuda.current_stream().wait_stream(s)
    return sorted(ts)[len(ts) // 2]
