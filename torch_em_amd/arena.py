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
        """The flat gradient arena if every p.grad is a view of one buffer in arena order, else None."""
        g0 = self.params[0].grad
        if g0 is None:
            return None
        base = g0.data_ptr() - 4 * self.offsets[id(self.params[0])][0]
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != base + 4 * self.offsets[id(p)][0] or not p.grad.is_contiguous():
                return None
        root = g0._base if g0._base is not None else g0
        while root._base is not None:
            root = root._base
        if root.data_ptr() != base or root.numel() < self.total or root.dim() != 1:
            return None
        return root[:self.total]
