"""The small helpers through which the rest of torch-em consumes what the trainers write (reference
torch_em/util/util.py): `get_trainer` (:366-384), `load_model` (:408-460), `model_is_equal` (:463-469),
`get_constructor_arguments` (:299-363), `ensure_tensor[_with_channels]` / `ensure_[spatial_]array` (:77-229).
Host-side plumbing only -- nothing here touches the GPU kernels."""
import os
import warnings
from collections import OrderedDict
from typing import Optional

import numpy as np
import torch

# numpy dtypes torch cannot hold -> the next wider signed type (what the reference's DTYPE_MAP does, util.py:17-22)
_WIDEN = {np.dtype("uint16"): np.int32, np.dtype("uint32"): np.int64, np.dtype("uint64"): np.int64}
_COMPILED_PREFIX = "_orig_mod."  # state_dict keys of torch.compile'd reference models


def ensure_tensor(tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    if isinstance(tensor, np.ndarray):
        if not tensor.dtype.isnative:  # wrong byte order (tif readers): torch.from_numpy refuses it
            tensor = tensor.astype(tensor.dtype.newbyteorder("="))
        if tensor.dtype in _WIDEN:
            tensor = tensor.astype(_WIDEN[tensor.dtype])
        tensor = torch.from_numpy(tensor if tensor.flags.writeable else tensor.copy())
    assert torch.is_tensor(tensor), f"Cannot convert {type(tensor)} to torch"
    return tensor if dtype is None else tensor.to(dtype=dtype)


def _to_rank(x, want: int, what: str):
    """Add one leading channel axis or strip leading singleton axes until `x` has `want` dimensions."""
    assert want <= x.ndim + 1 and x.ndim <= 5, f"{what}: cannot bring {x.ndim} dimensions to {want}"
    if x.ndim == want - 1:
        return x[None]
    while x.ndim > want:
        assert x.shape[0] == 1, f"{what}: leading axes of {tuple(x.shape)} must be singletons"
        x = x[0]
    return x


def ensure_tensor_with_channels(tensor, ndim: int, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Tensor with a channel axis in front of `ndim` spatial axes (ndim = 4: 3 spatial axes + time/channel, as the
    reference); batch axes of size one are dropped."""
    assert ndim in (2, 3, 4), f"{ndim}"
    t = ensure_tensor(tensor, dtype)
    if ndim == 4:
        assert t.ndim in (4, 5), f"{t.ndim}"
        return _to_rank(t, 4, "ensure_tensor_with_channels")
    assert t.ndim >= ndim, f"{t.ndim}"
    return _to_rank(t, ndim + 1, "ensure_tensor_with_channels")


def ensure_array(array, dtype=None) -> np.ndarray:
    if torch.is_tensor(array):
        array = array.detach().cpu().numpy()
    assert isinstance(array, np.ndarray), f"Cannot convert {type(array)} to numpy"
    return array if dtype is None else np.require(array, dtype=dtype)


def ensure_spatial_array(array, ndim: int, dtype=None) -> np.ndarray:
    """numpy array with exactly `ndim` (2 or 3) spatial axes; batch / channel axes must be singletons."""
    assert ndim in (2, 3)
    a = ensure_array(array, dtype)
    assert ndim <= a.ndim <= 5, str(a.ndim)
    return _to_rank(a, ndim, "ensure_spatial_array")


def get_constructor_arguments(obj) -> dict:
    """What has to be stored to re-create `obj` from a checkpoint: torch-em style objects carry `init_kwargs`;
    optimizers / schedulers are rebuilt from their state; DataLoaders from their public settings."""
    if hasattr(obj, "init_kwargs"):
        return obj.init_kwargs
    if isinstance(obj, (torch.optim.Optimizer, torch.optim.lr_scheduler.LRScheduler,
                        torch.optim.lr_scheduler.ReduceLROnPlateau)):
        return {}
    if isinstance(obj, torch.utils.data.DataLoader):
        sampler = getattr(obj, "sampler", None)
        plain = (torch.utils.data.RandomSampler, torch.utils.data.SequentialSampler, torch.utils.data.SubsetRandomSampler)
        if sampler is not None and not isinstance(sampler, plain):
            warnings.warn(f"DataLoader uses sampler {type(sampler).__name__}; only its `shuffle` setting is serialized, "
                          "a trainer rebuilt from the checkpoint will sample differently.")
        shuffle = getattr(sampler, "shuffle", None)
        if shuffle is None:
            shuffle = isinstance(sampler, (torch.utils.data.RandomSampler, torch.utils.data.SubsetRandomSampler))
        keys = ("batch_size", "num_workers", "pin_memory", "drop_last", "persistent_workers", "prefetch_factor", "timeout")
        return {**{k: getattr(obj, k) for k in keys}, "shuffle": bool(shuffle)}
    warnings.warn(f"Constructor arguments for {type(obj)} cannot be deduced; empty arguments are stored and "
                  "DefaultTrainer.from_checkpoint will probably not be able to rebuild it.")
    return {}


def get_trainer(checkpoint, name: str = "best", device=None):
    """Trainer from a checkpoint FOLDER (or the trainer itself, passed through)."""
    from ..trainer import DefaultTrainer
    if isinstance(checkpoint, str):
        assert os.path.exists(checkpoint), checkpoint
        checkpoint = DefaultTrainer.from_checkpoint(checkpoint, name=name, device=device)
    assert isinstance(checkpoint, DefaultTrainer)
    return checkpoint


def load_model(checkpoint: str, model: Optional[torch.nn.Module] = None, name: str = "best",
               state_key: Optional[str] = "model_state", device=None) -> torch.nn.Module:
    """Model from a trainer checkpoint folder or a serialized model / state file.  Without `model` the class and its
    arguments come from the checkpoint's `init` record (folder) or the file is a pickled module; with `model` only the
    state is loaded (`state_key=None`: the file is the bare state dict).  Checkpoints written by the reference load
    here and vice versa -- same keys, `_orig_mod.` prefixes of compiled models are stripped."""
    is_folder = os.path.isdir(checkpoint)
    if model is None:
        if is_folder:
            return get_trainer(checkpoint, name=name, device=device).model
        return torch.load(checkpoint, map_location=device, weights_only=False)
    path = os.path.join(checkpoint, f"{name}.pt") if is_folder else checkpoint
    state = torch.load(path, map_location=device, weights_only=False)
    if state_key is not None:
        state = state[state_key]
    strip = len(_COMPILED_PREFIX)
    model.load_state_dict(OrderedDict((k[strip:] if k.startswith(_COMPILED_PREFIX) else k, v) for k, v in state.items()))
    return model if device is None else model.to(device)


def model_is_equal(model1, model2) -> bool:
    return all(torch.equal(a.data, b.data) for a, b in zip(model1.parameters(), model2.parameters()))
