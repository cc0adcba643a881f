"""Dev tool: what torch's own fill reaches on this box, by buffer size (the ceiling the dc/dx writers are compared with)."""
import sys, torch
def t(fn, k=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/k
for gb in (4, 16, 48, 96):
    N = int(gb * 1e9 / 8)
    a = torch.empty(N, dtype=torch.float64, device="cuda")
    ms = t(lambda: a.zero_()); print(f"{gb:3d} GB zero_  {N*8/ms/1e6:7.1f} GB/s ({ms:.3f} ms)")
    ms = t(lambda: a.fill_(1.5)); print(f"{gb:3d} GB fill_  {N*8/ms/1e6:7.1f} GB/s ({ms:.3f} ms)", flush=True)
    del a
