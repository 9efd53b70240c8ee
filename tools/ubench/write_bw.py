import sys, os
sys.path.insert(0, '/root/repo')
import torch, time
x = torch.empty(6, 4**12, dtype=torch.float32, device='cuda')
y = torch.empty(4**12, dtype=torch.float32, device='cuda')
for name, fn in (("fill 402MB", lambda: x.fill_(1.0)), ("copy 67MB->6x (broadcast add)", lambda: torch.add(y, 1.0, out=x[0])), ("x.copy_(x2) 402MB r+w", None)):
    if fn is None:
        x2 = torch.empty_like(x); fn = lambda: x2.copy_(x)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    ts.sort(); print(name, "median %.4f ms" % ts[5], "%.0f GB/s written" % (x.numel()*4/ts[5]/1e6 if "fill" in name or "r+w" in name else y.numel()*4/ts[5]/1e6))
