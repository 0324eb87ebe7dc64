#!/usr/bin/env python3
"""How does a large device -> pinned-host copy travel (SDMA engine or a blit kernel on the CUs), and how fast?
Run under `rocprofv3 --memory-copy-trace --kernel-trace --stats` with different ROCclr / ROCr environment settings."""
import os, time, torch
dev = torch.device("cuda", 0)
n = 1 << 30  # 4 GiB of float32
d = torch.rand(n, device=dev)
h = torch.empty(n, dtype=torch.float32, pin_memory=True)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.copy_(d, non_blocking=True); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print({k: os.environ.get(k) for k in ("HSA_ENABLE_SDMA", "GPU_BLIT_ENGINE_TYPE", "DEBUG_CLR_LIMIT_BLIT_WG", "GPU_FORCE_BLIT_COPY_SIZE")},
      f"D2H {4 * n / dt / 1e9:.1f} GB/s", flush=True)
