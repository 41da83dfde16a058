import torch, time
dev="cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e-3
for mb in (42, 126, 168, 512, 2048):
    n=mb*1024*1024//2
    a=torch.empty(n, dtype=torch.bfloat16, device=dev); b=torch.randn(n, device=dev).to(torch.bfloat16)
    tz=t(lambda: a.zero_()); tc=t(lambda: a.copy_(b)); tr=t(lambda: b.sum())
    print(f"{mb:5d} MB: write-only {mb/1024/tz/1024*1.048576:6.2f} TB/s  copy(r+w) {2*mb/1024/tc/1024*1.048576:6.2f} TB/s  read-only {mb/1024/tr/1024*1.048576:6.2f} TB/s")
