import torch, time
def T(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it*1e3
x=torch.empty(64000,2048,dtype=torch.bfloat16,device='cuda'); y=torch.empty_like(x); z=torch.empty(2,64000,2048,dtype=torch.bfloat16,device='cuda')
t=T(lambda: x.fill_(1.0)); print(f"fill 262MB: {t:.1f} us  {x.numel()*2/t*1e-6:.2f} TB/s")
t=T(lambda: z.fill_(1.0)); print(f"fill 524MB: {t:.1f} us  {z.numel()*2/t*1e-6:.2f} TB/s")
t=T(lambda: y.copy_(x)); print(f"copy 262MB->262MB: {t:.1f} us  {2*x.numel()*2/t*1e-6:.2f} TB/s (r+w)")
xs=x[:, :1024]
t=T(lambda: xs.fill_(1.0)); print(f"fill strided (2KB of every 4KB row) 131MB: {t:.1f} us  {xs.numel()*2/t*1e-6:.2f} TB/s")
