import torch, time
N = 65536*120*252
a = torch.empty(N, dtype=torch.float64, device="cuda")
b = torch.empty(N//2, dtype=torch.float64, device="cuda")
def t(fn, k=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/k
ms = t(lambda: a.zero_()); print(f"zero_  {a.numel()*8/ms/1e6:.1f} GB/s write  ({ms:.3f} ms)")
ms = t(lambda: a.fill_(1.5)); print(f"fill_  {a.numel()*8/ms/1e6:.1f} GB/s write  ({ms:.3f} ms)")
ms = t(lambda: a[:N//2].copy_(b)); print(f"copy_  {b.numel()*8*2/ms/1e6:.1f} GB/s read+write ({ms:.3f} ms)")
ms = t(lambda: torch.cuda.memset if False else a.view(torch.uint8).zero_()); print(f"zero_u8 {a.numel()*8/ms/1e6:.1f} GB/s")
