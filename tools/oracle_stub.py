"""Dev-container-only helper: make the upstream `credit` package importable on CPU.

The upstream tree at /root/reference imports many optional packages at module
import time (xarray, netCDF4, timm, ...). None of them is needed for the
CrossFormer forward path, so this installs a meta-path finder that serves empty
stand-in modules for exactly those top-level names. It is used ONLY by
tools/make_goldens.py and by the reference-pinning tests, which are skipped
when /root/reference does not exist (e.g. on the GPU box).

Import this module BEFORE importing anything from `credit`.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

REFERENCE_ROOT = os.environ.get("WX_REFERENCE_ROOT", "/root/reference")

_ABSENT = {
    "xarray", "netCDF4", "timm", "torch_harmonics", "bridgescaler", "echo",
    "segmentation_models_pytorch", "zarr", "numba", "overrides", "torchvision",
    "cftime", "h5py", "metpy", "pvlib", "haversine", "torch_geometric",
    "torchmetrics", "obstore", "pygrib", "cartopy", "dask", "gcsfs", "s3fs",
    "numcodecs", "h5netcdf", "tensorboard",
}


class _Hollow(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        stand_in = MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, stand_in)
        return stand_in


class _HollowFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        top = fullname.split(".")[0]
        if top in _ABSENT:
            try:  # prefer the real thing when it exists
                for f in sys.meta_path:
                    if f is self:
                        continue
                    spec = f.find_spec(fullname, path, target) if hasattr(f, "find_spec") else None
                    if spec is not None:
                        return spec
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Hollow(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "credit"))


def install():
    """Idempotently install the finder and put the reference on sys.path."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    if not any(isinstance(f, _HollowFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _HollowFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
