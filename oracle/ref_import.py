"""Loader for the *reference* (czbiohub-sf/biahub) — test infrastructure only.

Used ONLY by ``tests/golden/make_golden.py`` inside the build container, where the
reference checkout is mounted read-only at ``/root/reference``.  It never runs on the GPU
box (the reference does not travel) and nothing in the product path imports it.

The reference's orchestration dependencies (iohub, submitit, monai, waveorder, ants, ...)
are absent from this image, so they are replaced by permissive stub modules: only the
pure torch / numpy / scipy arithmetic of the hot path is exercised.
"""

from __future__ import annotations

import importlib
import importlib.metadata
import sys
import types

REFERENCE_ROOT = "/root/reference"

_STUBS = [
    "submitit",
    "iohub",
    "iohub.ngff",
    "iohub.ngff.utils",
    "iohub.ngff.models",
    "monai",
    "monai.transforms",
    "monai.transforms.spatial",
    "monai.transforms.spatial.array",
    "natsort",
    "tqdm",
    "humanize",
    "waveorder",
    "waveorder.models",
    "waveorder.models.isotropic_fluorescent_thick_3d",
    "waveorder.focus",
    "waveorder.cli",
    "ants",
    "largestinteriorrectangle",
]


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy


def load_reference():
    """Make ``import biahub.<hot-path module>`` work; returns nothing."""
    sys.dont_write_bytecode = True
    for name in _STUBS:
        if name not in sys.modules:
            m = _StubModule(name)
            m.__path__ = []  # behave as a package
            sys.modules[name] = m
    sys.modules["natsort"].natsorted = sorted
    sys.modules["tqdm"].tqdm = lambda x, *a, **k: x

    real_version = importlib.metadata.version

    def _version(name):
        if name == "biahub":
            return "0.0.0+reference"
        return real_version(name)

    importlib.metadata.version = _version
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def reference_available() -> bool:
    import os

    return os.path.isdir(os.path.join(REFERENCE_ROOT, "biahub"))
