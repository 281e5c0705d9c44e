"""genvarloader_amd -- MI355X (gfx950) haplotype reconstruction + one-hot for GenVarLoader.

Only the hot path lives here: the HIP kernels + C-ABI (``csrc/``, ``include/gvl_hip.h``),
the device-resident dataset arrays (:class:`HapsDevice`) and a drop-in mirror of the
reference's FFI entry points (:mod:`genvarloader_amd.ffi`).  Importing the package does
not need a GPU; calling a compute entry point without the HIP library or without a HIP
device raises (there is no CPU fallback).
"""

from . import synth  # noqa: F401
from ._lib import GvlError, lib_path, load  # noqa: F401

__all__ = ["HapsDevice", "ffi", "synth", "GvlError", "load", "lib_path"]


def __getattr__(name):
    # torch is imported lazily so that `import genvarloader_amd` stays cheap
    if name == "HapsDevice":
        from .device import HapsDevice

        return HapsDevice
    if name in ("ffi", "device", "loader", "sharding"):
        import importlib

        return importlib.import_module(f".{name}", __name__)
    raise AttributeError(name)
