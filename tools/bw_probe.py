import torch
dev = torch.device("cuda:0")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for mb in (115.6, 462.4, 1849.6):
    n = int(mb * 1e6 / 4)
    x = torch.empty(n, dtype=torch.float32, device=dev)
    y = torch.empty(n, dtype=torch.float32, device=dev)
    u = torch.empty(n, dtype=torch.uint8, device=dev)
    t_fill = timeit(lambda: x.fill_(1.0))
    t_copy = timeit(lambda: y.copy_(x))
    t_cvt = timeit(lambda: y.copy_(u))
    print(f"{mb:.0f} MB: fill {t_fill:.1f}us = {mb/t_fill*1e3/1e3:.2f} TB/s write | copy {t_copy:.1f}us = {2*mb/t_copy:.2f} TB/s r+w | u8->f32 {t_cvt:.1f}us = {1.25*mb/t_cvt:.2f} TB/s")
