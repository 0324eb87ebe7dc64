"""Is the per-process state of the fused update kernel a property of where its code object sits?  Load several copies of
libbhcore.so into ONE process (each brings its own code object, context and scratch) and time the R-L iteration through each:
python tools/codeobj_probe.py [ncopies]"""
import ctypes as C, shutil, sys, tempfile
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tmp = Path(tempfile.mkdtemp())
for i in range(n):
    path = tmp / f"libbhcore_copy{i}.so"
    shutil.copy(_lib.LIB_PATH, path)
    lib = C.CDLL(str(path))
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    h = C.c_void_p()
    assert lib.bh_ctx_create(0, None, C.byref(h)) == 0
    lib.bh_ctx_set_stream(h, C.c_void_p(torch.cuda.current_stream(0).cuda_stream))
    lib.bh_ctx_set_timing(h, 1)
    ms = C.c_float()
    for _ in range(2):
        assert lib.bh_richardson_lucy(h, ptr(d), ptr(psf), 33, 17, 17, *shape, 4, 1e-6, ptr(out)) == 0
        lib.bh_last_elapsed_ms(h, _lib.T_RL_ITER, C.byref(ms))
    print(f"copy {i}: {ms.value:.3f} ms/iter", flush=True)
    lib.bh_ctx_release_workspace(h)
