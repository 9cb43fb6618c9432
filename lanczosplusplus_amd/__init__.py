"""lanczosplusplus_amd -- MI355X-native Lanczos inner engine behind LanczosPlusPlus's
Model / Basis / InternalProductStored plug-in surface (one hot path).

The compute path is liblpp_engine.so (hand-written HIP for gfx950 + a C ABI, csrc/);
this package is the thin host layer: ctypes binding, geometry/input helpers, the
torch.distributed communicator used by the multi-GPU path.
"""
from ._capi import LIB_PATH, LppError  # noqa: F401
from .engine import LanczosEngine, partition_rows, split_csr, tridiag_lowest  # noqa: F401
from . import geometry  # noqa: F401

__all__ = ["LanczosEngine", "LppError", "partition_rows", "split_csr", "tridiag_lowest", "geometry", "LIB_PATH"]
