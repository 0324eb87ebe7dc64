#!/usr/bin/env python3
"""Build a tuning variant of libbhcore.so: `tools/build_variant.py NAME -DBH_FC_XNT=512 ...` compiles csrc/fftconv.hip (or `--src=FILE.hip` as the first flag) with
the extra flags and links it with the stock objects into biahub_amd/build/variants/libbhcore_NAME.so (A/B runs on one GPU
box: BHCORE_LIB=<that file> python bench.py ...)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from biahub_amd import build as B  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
src = "fftconv.hip"
if flags and flags[0].startswith("--src="):  # another translation unit, e.g. --src=affine.hip
    src, flags = flags[0][6:], flags[1:]
B.build(verbose=False)
out = B.PKG / "build" / "variants"
out.mkdir(parents=True, exist_ok=True)
obj = out / f"{src.split('.')[0]}_{name}.o"
base = ["-O3", f"--offload-arch={B.ARCH}", "-fPIC", "-std=c++17", f"-I{B.INCLUDE}", f"-I{B.CSRC}", "-Wall", "-Wno-unused-function"]
subprocess.run([B._hipcc(), *base, *flags, "-c", str(B.CSRC / src), "-o", str(obj)], check=True)
objs = [str(obj) if s == src else str(B.PKG / "build" / (s + ".o")) for s in B.SOURCES]
lib = out / f"libbhcore_{name}.so"
subprocess.run([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", str(lib), *objs, "-L/opt/rocm/lib", "-lhipfft",
                "-Wl,-rpath,/opt/rocm/lib"], check=True)
print(lib)
