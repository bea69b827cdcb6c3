"""aim_amd -- MI355X-native AIM ViT-CLIP + Adapter hot path.

Python host code over ``libaim_hip.so`` (hand-written gfx950 kernels behind the C ABI declared in
``include/aim_kernels.h``).  The module surface mirrors the reference's MMAction2 one for this path:
``ViT_CLIP`` (backbone registry entry), ``Recognizer3D``, ``I3DHead``, ``CrossEntropyLoss``,
``build_model`` and a minimal mmcv-compatible ``Config``.

There is no CPU or eager-PyTorch fallback for the backbone: without the HIP library (or without a
GPU) constructing the compute path raises.
"""
from .lib import load_library, library_path, LibraryNotBuilt  # noqa: F401

__all__ = ["load_library", "library_path", "LibraryNotBuilt"]
