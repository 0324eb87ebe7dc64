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
import importlib.util
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
    # only needed by biahub.estimate_stabilization (phase cross-correlation)
    "dask",
    "dask.array",
    "pystackreg",
    "matplotlib",
    "matplotlib.pyplot",
    "skimage",
    "skimage.transform",
    "skimage.feature",
    "skimage.filters",
    "skimage.measure",
    "skimage.registration",
    "sklearn",
    "sklearn.neighbors",
    # only needed by biahub.characterize_psf (detect_peaks)
    "markdown",
]


class _DummyMeta(type):
    def __getattr__(cls, name):  # `da.Array` in an annotation is a CLASS attribute lookup
        if name.startswith("__"):
            raise AttributeError(name)
        return cls


class _Dummy(metaclass=_DummyMeta):
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
            try:  # never shadow a package that is really installed
                if importlib.util.find_spec(name.split(".")[0]) is not None:
                    continue
            except (ImportError, ValueError):
                pass
            m = _StubModule(name)
            m.__path__ = []  # behave as a package
            sys.modules[name] = m
    if isinstance(sys.modules.get("natsort"), _StubModule):
        sys.modules["natsort"].natsorted = sorted
    if isinstance(sys.modules.get("tqdm"), _StubModule):
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
