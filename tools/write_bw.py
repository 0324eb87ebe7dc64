"""Pure-write / copy bandwidth probe (torch kernels): how fast can a write-only pass be on this part?"""
import time
import torch
dev = torch.device("cuda", 0)
n = 1 << 31
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)
for name, fn, nbytes in (("fill_ (write only)", lambda: a.fill_(1.0), 4 * n), ("zero_ (memset)", lambda: a.zero_(), 4 * n),
                         ("copy_ (read + write)", lambda: b.copy_(a), 8 * n), ("mul_ in place (read + write)", lambda: a.mul_(1.0001), 8 * n),
                         ("sum (read only)", lambda: a.sum(), 4 * n)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:32s} {dt*1e3:7.2f} ms  {nbytes/dt/1e12:5.2f} TB/s")
