import torch
for mb in (64, 256, 1024):
    n = mb * 1024 * 1024 // 2
    a = torch.randn(n, device='cuda').bfloat16(); b = torch.randn(n, device='cuda').bfloat16(); c = torch.empty_like(a)
    for name, f, passes in (("copy", lambda: c.copy_(a), 2), ("add", lambda: torch.add(a, b, out=c), 3), ("sum", lambda: a.sum(), 1)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{mb} MiB {name}: {ms*1e3:.1f} us  {passes*n*2/ms/1e9:.2f} TB/s")
