"""The input side of the training step on the device.

The reference prepares a sample in the CPU loader workers -- raw transform `standardize` (transform/raw.py:40-65), label
transform (boundaries / affinities, transform/label.py), augmentations (transform/augmentation.py), see
`SegmentationDataset.__getitem__` (data/segmentation_dataset.py:226-249) -- and the trainer copies the finished float
tensors to the GPU (`trainer/default_trainer.py:812`).  Here the loader ships the RAW volume and the INTEGER labels and
`DevicePrefetcher` does the rest on the device, one batch ahead of the training step:

    loader thread      pinned (x_raw, labels) of batch k+1
    side HIP stream    H2D copy -> standardize -> flips / warps (replayed on the labels) -> target kernel
    compute stream     training step of batch k      (waits on the side stream's event before it touches batch k+1)

so neither the PCIe copy (16 MB + 16 MB for a cfg-2 batch instead of 16 MB + 64 MB of float targets) nor the pre-pass
(one pass over each tensor, HBM-bound) sits on the step's critical path.  Every pre-pass kernel runs on torch's current
stream, which inside `torch.cuda.stream(side)` is the side stream; workspaces are per stream (ops._workspace).
"""
from typing import Callable, Iterable, Optional

import torch


class DevicePrefetcher:
    """Iterates a loader of (x, y) CPU batches; yields device batches whose copy and `prepass` were enqueued on a side
    stream while the previous batch was being trained on.  `prepass(x, y) -> (x, y)` runs on the device."""

    def __init__(self, loader: Iterable, device, prepass: Optional[Callable] = None, enabled: bool = True):
        self.loader, self.device, self.prepass, self.enabled = loader, torch.device(device), prepass, enabled
        self._stream = None

    def __len__(self):
        return len(self.loader)

    def _fetch(self, it):
        try:
            x, y = next(it)
        except StopIteration:
            return None
        if not self.enabled:
            x, y = x.to(self.device, non_blocking=True), y.to(self.device, non_blocking=True)
            if self.prepass is not None:
                x, y = self.prepass(x, y)
            return x, y, None
        if self._stream is None:
            self._stream = torch.cuda.Stream(self.device)
        with torch.cuda.stream(self._stream):
            xs = [t if (not torch.is_tensor(t) or t.is_cuda or t.is_pinned()) else t.pin_memory() for t in (x, y)]
            x, y = (t.to(self.device, non_blocking=True) for t in xs)
            if self.prepass is not None:
                x, y = self.prepass(x, y)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return x, y, ev

    def __iter__(self):
        it = iter(self.loader)
        nxt = self._fetch(it)
        while nxt is not None:
            x, y, ev = nxt
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for t in (x, y):   # the caching allocator must not hand these blocks back to the side stream early
                    if torch.is_tensor(t):
                        t.record_stream(cur)
            # enqueue batch k+1 (copy + pre-pass, side stream) BEFORE the caller enqueues the step of batch k
            nxt = self._fetch(it)
            yield x, y
