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

from .registry import (BACKBONES, HEADS, LOSSES, MODELS, RECOGNIZERS, Config, Registry,  # noqa: E402,F401
                       build_backbone, build_from_cfg, build_head, build_loss, build_model, build_recognizer,
                       register_into_mmaction)
from .backbone import ViT_CLIP  # noqa: E402,F401
from .aim_variant import AIM  # noqa: E402,F401
from .recognizer import (CrossEntropyLoss, GPUNormalize, I3DHead, Recognizer3D,  # noqa: E402,F401
                         register_module_hooks, top_k_accuracy)

from .dist import DistOptimizerHook, FlatAdamW, build_optimizer  # noqa: E402,F401

__all__ += ["DistOptimizerHook", "FlatAdamW", "build_optimizer", "BACKBONES", "HEADS", "LOSSES", "MODELS", "RECOGNIZERS", "Config", "Registry", "build_backbone",
            "build_from_cfg", "build_head", "build_loss", "build_model", "build_recognizer", "register_into_mmaction",
            "ViT_CLIP", "AIM", "Recognizer3D", "I3DHead", "CrossEntropyLoss", "GPUNormalize", "register_module_hooks",
            "top_k_accuracy"]
