"""In-memory image/label dataset (the `TensorDataset` branch of default_segmentation_dataset,
reference segmentation.py:143-148, data/tensor_dataset.py, sample pipeline
data/image_collection_dataset.py:241-264): per sample
    random crop to patch_shape -> raw_transform -> label_transform -> transform(raw, labels)
    -> label_transform2 -> float tensors with a channel axis.
File-backed datasets (HDF5/zarr/tif readers, ~250 download wrappers) are data plumbing outside
the hot path and are not re-implemented; feed arrays.
"""
from typing import Callable, List, Optional, Tuple, Union

import numpy as np
import torch


class TensorDataset(torch.utils.data.Dataset):
    def __init__(self, images: List[Union[np.ndarray, torch.Tensor]], labels: List[Union[np.ndarray, torch.Tensor]],
                 patch_shape: Tuple[int, ...], raw_transform: Optional[Callable] = None,
                 label_transform: Optional[Callable] = None, label_transform2: Optional[Callable] = None,
                 transform: Optional[Callable] = None, dtype: torch.dtype = torch.float32,
                 label_dtype: torch.dtype = torch.float32, n_samples: Optional[int] = None,
                 sampler: Optional[Callable] = None, with_padding: bool = True, with_channels: bool = False):
        if len(images) != len(labels):
            raise ValueError(f"Number of images and labels does not match: {len(images)}, {len(labels)}")
        ndim = len(patch_shape)
        for im, lab in zip(images, labels):
            if len(im.shape) != ndim + (1 if with_channels else 0):
                raise ValueError("Image shape does not match the patch shape")
            if tuple(im.shape[1:] if with_channels else im.shape) != tuple(lab.shape):
                raise ValueError("Image and label shape does not match")
        self.raw_images, self.label_images = images, labels
        self.patch_shape, self.with_channels, self._ndim = tuple(patch_shape), with_channels, ndim
        self.raw_transform, self.label_transform = raw_transform, label_transform
        self.label_transform2, self.transform = label_transform2, transform
        self.sampler, self.with_padding = sampler, with_padding
        self.dtype, self.label_dtype = dtype, label_dtype
        self._len = len(images) if n_samples is None else n_samples
        self.sample_random_index = n_samples is not None

    def __len__(self):
        return self._len

    @property
    def ndim(self):
        return self._ndim

    def _crop(self, raw, labels):
        shape = labels.shape
        start = [np.random.randint(0, sh - psh) if sh - psh > 0 else 0 for sh, psh in zip(shape, self.patch_shape)]
        bb = tuple(slice(s, s + p) for s, p in zip(start, self.patch_shape))
        raw = raw[(slice(None),) + bb] if self.with_channels else raw[bb]
        labels = labels[bb]
        if self.with_padding and tuple(labels.shape) != self.patch_shape:
            pad = [(0, p - s) for p, s in zip(self.patch_shape, labels.shape)]
            labels = np.pad(labels, pad, mode="reflect")
            raw = np.pad(raw, ([(0, 0)] if self.with_channels else []) + pad, mode="reflect")
        return raw, labels

    def __getitem__(self, index):
        if self.sample_random_index:
            index = np.random.randint(0, len(self.raw_images))
        raw, labels = self.raw_images[index], self.label_images[index]
        raw = raw.numpy() if torch.is_tensor(raw) else np.asarray(raw)
        labels = labels.numpy() if torch.is_tensor(labels) else np.asarray(labels)
        raw, labels = self._crop(raw, labels)
        if self.raw_transform is not None:
            raw = self.raw_transform(raw)
        if self.label_transform is not None:
            labels = self.label_transform(labels)
        if self.transform is not None:
            raw, labels = self.transform(raw, labels)
        if self.label_transform2 is not None:
            labels = self.label_transform2(labels)
        raw = torch.as_tensor(np.ascontiguousarray(raw)).to(self.dtype)
        labels = torch.as_tensor(np.ascontiguousarray(labels)).to(self.label_dtype)
        if raw.dim() == self._ndim:
            raw = raw[None]
        if labels.dim() == self._ndim:
            labels = labels[None]
        return raw, labels
