import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from biahub_amd import _lib
from biahub_amd.device import ptr
from bench import PSF_SHAPE, PSF_SIGMA, gaussian_psf, synthetic_position
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
shape = (512, 2048, 2048)
d = synthetic_position(shape, 1, dev)
psf = gaussian_psf(PSF_SHAPE, PSF_SIGMA, dev)
lib = _lib.load()
h = C.c_void_p(); assert lib.bh_ctx_create(0, None, C.byref(h)) == 0
lib.bh_ctx_set_stream(h, C.c_void_p(torch.cuda.current_stream(0).cuda_stream)); lib.bh_ctx_set_timing(h, 1)
ms = C.c_float()
outs = [torch.empty_like(d) for _ in range(6)]
ds = [d] + [d.clone() for _ in range(2)]
for j, dd in enumerate(ds):
    for i, out in enumerate(outs):
        for _ in range(2):
            assert lib.bh_richardson_lucy(h, ptr(dd), ptr(psf), 33, 17, 17, *shape, 3, 1e-6, ptr(out)) == 0
            lib.bh_last_elapsed_ms(h, _lib.T_RL_ITER, C.byref(ms))
        print(f"d{j} out{i}: {ms.value:.3f}", end="  ", flush=True)
    print()
