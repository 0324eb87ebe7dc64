"""Does the 256-MB memory-side cache serve data a previous kernel wrote?  In-place x.mul_() over buffers of growing size,
back to back (each launch re-reads what the previous one wrote): bytes / time per size."""
import time
import torch
dev = torch.device("cuda", 0)
for mb in (32, 64, 96, 128, 160, 192, 224, 256, 320, 512, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.rand(n, device=dev)
    reps = max(20, int(40 * 256 / mb))
    for _ in range(5):
        x.mul_(1.0000001)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        x.mul_(1.0000001)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{mb:5d} MB: {dt*1e6:8.1f} us per pass  {2*n*4/dt/1e12:6.2f} TB/s (read + write)")
# producer / consumer on different buffers: write A (fill), read A into B
for mb in (64, 128, 512):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    for _ in range(3):
        a.fill_(1.5); b.copy_(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        a.fill_(1.5); b.copy_(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"fill {mb} MB then copy it: {dt*1e6:8.1f} us per pair  {3*n*4/dt/1e12:6.2f} TB/s")
