#!/usr/bin/env python3
"""Relink every tuning variant under biahub_amd/build/variants against the CURRENT stock objects (after the other translation
units changed): `tools/relink_variants.py`.  A variant object NAME is `<unit>_<name>.o` (tools/build_variant.py)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from biahub_amd import build as B  # noqa: E402

B.build(verbose=False)
out = B.PKG / "build" / "variants"
for obj in sorted(out.glob("*.o")):
    unit, name = obj.stem.split("_", 1)
    src = unit + ".hip"
    objs = [str(obj) if s == src else str(B.PKG / "build" / (s + ".o")) for s in B.SOURCES]
    lib = out / f"libbhcore_{name}.so"
    subprocess.run([B._hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", str(lib), *objs, "-L/opt/rocm/lib", "-lhipfft",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    print(lib.name)
