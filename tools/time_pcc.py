"""Phase cross-correlation of two resident 512 x 2048 x 2048 volumes: per-call time with and without the correlation volume."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from biahub_amd.estimate_stabilization import phase_cross_corr_device  # noqa: E402

dev = torch.device("cuda", 0)
shape = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 2048, 2048)
vol = torch.rand(shape, device=dev)
mov = torch.roll(vol, (3, -7, 11), (0, 1, 2))
for norm in ("magnitude", None, "classic"):
    for want in (False, True):
        sh, _ = phase_cross_corr_device(vol, mov, norm, want_corr=want)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            sh, c = phase_cross_corr_device(vol, mov, norm, want_corr=want)
            del c
        torch.cuda.synchronize()
        print(f"norm={norm} want_corr={want}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms  shift {sh}", flush=True)

from biahub_amd.estimate_stabilization import PreparedPhaseCrossCorr  # noqa: E402

for roll in (False, True):
    with PreparedPhaseCrossCorr(mov, fixed_is_second=True, device=dev) as h:
        sh, _ = h(vol, "magnitude", roll=roll)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            sh, _ = h(vol, "magnitude", roll=roll)
        torch.cuda.synchronize()
        print(f"prepared (stored image = second factor) roll={roll}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms  shift {sh}", flush=True)
