"""MI355X-native QCNet / TempEnsLearnAgg graph-attention hot path (package directory
``tempenslearnagg-qcnet_amd``, importable as ``tempenslearnagg_qcnet_amd``).

Layout mirrors the reference's top-level packages: ``layers`` / ``modules`` / ``predictors`` / ``utils``;
``csrc`` holds the HIP kernels and the C ABI (``include/qcnet_hip.h``), ``ops`` the tensor-level wrappers.
"""
from . import _lib, data, ops, synthetic  # noqa: F401
from .data import Batch, HeteroData  # noqa: F401

__version__ = '0.1.0'
