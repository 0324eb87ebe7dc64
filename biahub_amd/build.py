"""Build libbhcore.so (HIP, gfx950) in-tree with hipcc.  `python -m biahub_amd.build`."""

from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
LIB = PKG / "libbhcore.so"
SOURCES = ["context.hip", "deskew.hip", "fill.hip", "deconv.hip", "fftconv.hip", "affine.hip", "spline.hip", "copy.hip", "regmetric.hip", "flatfield.hip", "psf.hip", "binning.hip", "codec.hip", "lz4.hip", "mask.hip", "invtf.hip", "host_deskew.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


# textual includes of a translation unit (rebuild triggers)
INCLUDES = {"fftconv.hip": ("fftconv_xpass.inc", "fftconv_xw.inc", "fftconv_x3.inc", "fftconv_colw.inc", "fftconv_colz.inc", "fftconv_colz3.inc"), "affine.hip": ("affine_zwalk.inc", "affine_zoblique.inc")}


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [CSRC / "common.hpp", INCLUDE / "bhcore.h"] + [CSRC / i for v in INCLUDES.values() for i in v]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return LIB
    objdir = PKG / "build"
    objdir.mkdir(exist_ok=True)
    flags = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", f"-I{INCLUDE}", f"-I{CSRC}",
             "-Wall", "-Wno-unused-function"]

    def compile_one(src: str) -> Path:
        obj = objdir / (src + ".o")
        srcp = CSRC / src
        hdr_t = max([(CSRC / "common.hpp").stat().st_mtime, (INCLUDE / "bhcore.h").stat().st_mtime]
                    + [(CSRC / i).stat().st_mtime for i in INCLUDES.get(src, ())])
        if not force and obj.exists() and obj.stat().st_mtime > max(srcp.stat().st_mtime, hdr_t):
            return obj
        cmd = [_hipcc(), *flags, "-c", str(srcp), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs),
           "-L/opt/rocm/lib", "-lhipfft", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
