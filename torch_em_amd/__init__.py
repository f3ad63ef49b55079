"""torch_em_amd -- MI355X-native 3D U-Net training path behind torch-em's interfaces.

Public names mirror `torch_em/__init__.py:5-10` of the reference for the hot path.
Importing the package does not need a GPU; running any op does (there is no CPU fallback).
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    # lazy: keep `import torch_em_amd` light for the CPU-only checks
    if name in ("default_segmentation_trainer", "default_segmentation_loader", "default_segmentation_dataset"):
        from . import segmentation
        return getattr(segmentation, name)
    if name in ("model", "loss", "transform", "trainer", "multi_gpu_training", "segmentation", "ops", "data"):
        import importlib
        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
