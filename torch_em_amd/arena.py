"""Flat fp32 arenas for parameters, gradients and optimizer state.

MI355X-first memory layout for the training step: the 46 parameter tensors of the U-Net live
in ONE contiguous HBM buffer (each tensor a 16-byte-aligned view), and so do their gradients and
AdamW moments.  The optimizer step is then a single launch over 85 MB and the data-parallel
gradient exchange is an RCCL all-reduce over contiguous ranges of the gradient arena -- no
per-tensor launches, no bucket copies (the reference relies on torch's for-each AdamW over 46
tensors and on DDP's bucket flatten/unflatten copies, multi_gpu_training.py:79).
"""
from typing import Dict, List, Tuple

import torch


def arena_layout(params: List[torch.Tensor]) -> Tuple[Dict[int, Tuple[int, int]], int]:
    """id(param) -> (offset, numel) with every offset a multiple of 4 floats; total length."""
    total, offsets = 0, {}
    for p in params:
        offsets[id(p)] = (total, p.numel())
        total += (p.numel() + 3) // 4 * 4
    return offsets, total


class ParamArena:
    """Re-homes the parameters of a module into one flat buffer (p.data becomes a view)."""

    def __init__(self, module: torch.nn.Module):
        self.params = [p for p in module.parameters()]
        self.offsets, self.total = arena_layout(self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        for p in self.params:
            o, n = self.offsets[id(p)]
            view = self.flat[o:o + n].view(p.shape)
            view.copy_(p.data)
            p.data = view

    def is_current(self) -> bool:
        """False once something (e.g. module.to(device)) replaced the parameter storages."""
        base = self.flat.data_ptr()
        return all(p.data_ptr() == base + 4 * self.offsets[id(p)][0] for p in self.params)

    def grads_flat(self):
        """The flat gradient arena if every p.grad is the engine's view of one buffer in arena order, else None.
        The engine leaves a reference to its arena on the first parameter (`_tem_grad_flat`); autograd may have
        detached the views it was handed, so membership is verified by address."""
        flat = getattr(self.params[0], "_tem_grad_flat", None)
        if flat is None or flat.numel() < self.total or self.params[0].grad is None:
            return None
        base = flat.data_ptr()
        for p in self.params:
            g = p.grad
            if g is None or g.data_ptr() != base + 4 * self.offsets[id(p)][0] or not g.is_contiguous():
                return None
        return flat[:self.total]
