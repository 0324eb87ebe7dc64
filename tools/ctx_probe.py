"""Companion of codeobj_probe.py: ONE library (one code object), several contexts in one process — each builds its own plan
tables and scratch.  python tools/ctx_probe.py [ncontexts]"""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from biahub_amd import _lib
from biahub_amd.device import ptr
from bench import PSF_SHAPE, PSF_SIGMA, gaussian_psf, synthetic_position
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
shape = (512, 2048, 2048)
d = synthetic_position(shape, 1, dev)
out = torch.empty_like(d)
psf = gaussian_psf(PSF_SHAPE, PSF_SIGMA, dev)
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
if "--spec-only" in sys.argv:  # one context; BH_FC_SPEC_REALLOC=1 in the environment re-allocates only the spectrum per call
    h = C.c_void_p()
    assert lib.bh_ctx_create(0, None, C.byref(h)) == 0
    lib.bh_ctx_set_stream(h, C.c_void_p(torch.cuda.current_stream(0).cuda_stream))
    lib.bh_ctx_set_timing(h, 1)
    ms = C.c_float()
    for i in range(n):
        assert lib.bh_richardson_lucy(h, ptr(d), ptr(psf), 33, 17, 17, *shape, 4, 1e-6, ptr(out)) == 0
        lib.bh_last_elapsed_ms(h, _lib.T_RL_ITER, C.byref(ms))
        print(f"call {i}: {ms.value:.3f} ms/iter", flush=True)
    sys.exit(0)
for i in range(n):
    h = C.c_void_p()
    assert lib.bh_ctx_create(0, None, C.byref(h)) == 0
    lib.bh_ctx_set_stream(h, C.c_void_p(torch.cuda.current_stream(0).cuda_stream))
    lib.bh_ctx_set_timing(h, 1)
    ms = C.c_float()
    for _ in range(2):
        assert lib.bh_richardson_lucy(h, ptr(d), ptr(psf), 33, 17, 17, *shape, 4, 1e-6, ptr(out)) == 0
        lib.bh_last_elapsed_ms(h, _lib.T_RL_ITER, C.byref(ms))
    print(f"context {i}: {ms.value:.3f} ms/iter", flush=True)
    lib.bh_ctx_release_workspace(h)
