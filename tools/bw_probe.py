import torch, time
for mb in (105, 210, 419, 838):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device="cuda").to(torch.bfloat16); y = torch.empty_like(x); z = torch.empty_like(x)
    for name, fn, traffic in (("copy (r+w)", lambda: y.copy_(x), 2), ("read-only sum", lambda: x.float().sum() if False else torch.sum(x, dtype=torch.float32), 1),
                              ("fill (w)", lambda: y.fill_(1.0), 1), ("add out-of-place (2r+w)", lambda: torch.add(x, y, out=z), 3)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{mb:4d} MB {name:24s} {us:8.1f} us  {traffic * mb * 1.048576 / us * 1e3:7.0f} GB/s", flush=True)
