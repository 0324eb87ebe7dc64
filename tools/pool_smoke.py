import sys, time
sys.path.insert(0, '.')
import torch
from biahub_amd.device import volume_pool, empty
dev = torch.device('cuda', 0)
t0 = time.perf_counter()
a = empty((512, 2048, 2048), torch.float32, dev); torch.cuda.synchronize()
print('pooled 8.6 GB alloc %.1f ms' % ((time.perf_counter() - t0) * 1e3), hex(a.data_ptr()))
a.fill_(1.0); print(float(a[3, 5, 7]), float(a.sum(dtype=torch.float64)))
t0 = time.perf_counter(); b = torch.empty((512, 2048, 2048), dtype=torch.float32, device=dev); torch.cuda.synchronize()
print('plain 8.6 GB alloc %.1f ms' % ((time.perf_counter() - t0) * 1e3))
for name, t in (('pooled', a), ('plain', b)):
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); t.mul_(1.0001); torch.cuda.synchronize()
    print(name, 'in-place mul_ %.3f ms' % ((time.perf_counter() - t0) * 1e3))
    c = torch.empty_like(t) if name == 'plain' else empty(t.shape, t.dtype, dev)
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); c.copy_(t); torch.cuda.synchronize()
    print(name, 'copy_ %.3f ms' % ((time.perf_counter() - t0) * 1e3))
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); c.fill_(2.0); torch.cuda.synchronize()
    print(name, 'fill_ %.3f ms' % ((time.perf_counter() - t0) * 1e3))
    del c
del a, b
torch.cuda.empty_cache()
print('reserved after empty_cache %.2f GB' % (torch.cuda.memory_reserved() / 1e9))
from biahub_amd.device import release_volume_pool
release_volume_pool()
print('reserved after release_volume_pool %.2f GB' % (torch.cuda.memory_reserved() / 1e9))
free, total = torch.cuda.mem_get_info()
print('driver free %.1f of %.1f GB' % (free / 1e9, total / 1e9))
