"""Data-parallel training over the GPUs of one node (reference torch_em/multi_gpu_training.py).

Same entry points and signatures as the reference -- `setup` (:13-18), `cleanup` (:21-24),
`DDP` (:43-52), `train_multi_gpu` (:107-190) -- one OS process per GPU, rank == device index,
`torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm).

What differs is how gradients are exchanged.  The reference wraps the model in torch's
DistributedDataParallel, whose reducer copies gradients into 25 MB buckets and all-reduces
those (85.43 MB per step for the benchmark U-Net).  Here the engine already writes all gradients
into ONE flat arena (torch_em_amd/arena.py), in the order backward produces them, so `DDP`:
  * all-reduces contiguous ranges of that arena IN PLACE (no bucket copies), and
  * launches each range as soon as backward has produced it: ProcessGroupNCCL runs the collective
    on its own HIP stream (it waits on an event of the compute stream), so the exchange overlaps
    the remaining backward kernels; `finish()` joins the streams before the optimizer step.
xGMI is point-to-point (7 links/GPU), so ranges are coalesced to >= `bucket_mb` to stay
bandwidth- rather than latency-bound per link.  ReduceOp.AVG reproduces DDP's averaging.
"""
import os
from typing import Any, Callable, Dict, Optional

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def setup(rank: int, world_size: int, backend: Optional[str] = None, port: Optional[int] = None):
    """Initialise the process group (reference :13-18; 127.0.0.1 instead of 'localhost')."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(port or 12355))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(rank)
    dist.init_process_group(backend, rank=rank, world_size=world_size)


def cleanup():
    """Destroy the process group (reference :21-24)."""
    if dist.is_initialized():
        dist.destroy_process_group()


class GradSync:
    """Overlapped, in-place all-reduce of ranges of the flat gradient arena."""

    def __init__(self, process_group=None, bucket_mb: float = 8.0):
        self.pg = process_group
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self.world = dist.get_world_size(process_group)
        self._avg = dist.get_backend(process_group) == "nccl"
        # observability (bench.py `ddp` keys): bytes and collectives of the last exchange, and -- when `measure` is set --
        # HIP events around the join in finish(): the time the compute stream had to WAIT for collectives (what is left
        # exposed after the overlap with backward)
        self.measure = False
        self.stats = {"bytes": 0, "n_collectives": 0}
        self._events = []
        self.reset()

    def reset(self):
        self._pending = []   # (lo, hi) element ranges produced but not yet launched
        self._works = []
        self._launched = []
        self._bytes = self._ncoll = 0

    @staticmethod
    def _coalesce(ranges):
        ranges = sorted(ranges)
        out = [list(ranges[0])]
        for lo, hi in ranges[1:]:
            if lo <= out[-1][1]:
                out[-1][1] = max(out[-1][1], hi)
            else:
                out.append([lo, hi])
        return [tuple(r) for r in out]

    def _launch(self, flat, ranges):
        for lo, hi in self._coalesce(ranges):
            view = flat[lo:hi]
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            self._works.append((dist.all_reduce(view, op=op, group=self.pg, async_op=True), view))
            self._launched.append((lo, hi))
            self._bytes += (hi - lo) * 4
            self._ncoll += 1

    def ready(self, flat: torch.Tensor, lo: int, hi: int):
        """Backward has finished writing flat[lo:hi] (enqueued on the current stream)."""
        self._pending.append((lo, hi))
        if sum(h - l for l, h in self._pending) >= self.bucket_elems:
            self._launch(flat, self._pending)
            self._pending = []

    def finish(self, flat: torch.Tensor):
        """Launch what is left, then make the current stream wait for every collective."""
        done = sum(h - l for l, h in self._launched) + sum(h - l for l, h in self._pending)
        if done < flat.numel():  # ranges nobody announced (parameters without gradient): exchange everything else
            covered = self._coalesce(self._launched + self._pending) if (self._launched or self._pending) else []
            pos, rest = 0, []
            for lo, hi in covered:
                if lo > pos:
                    rest.append((pos, lo))
                pos = max(pos, hi)
            if pos < flat.numel():
                rest.append((pos, flat.numel()))
            self._pending += rest
        if self._pending:
            self._launch(flat, self._pending)
        ev = None
        if self.measure and flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for work, view in self._works:
            work.wait()
            if not self._avg:
                view.div_(self.world)
        if ev is not None:
            ev[1].record()
            self._events.append(ev)
        self.stats = {"bytes": self._bytes, "n_collectives": self._ncoll}
        self.reset()

    def exposed_ms(self):
        """Mean time (ms) the compute stream waited in finish() over the measured steps so far (synchronises)."""
        if not self._events:
            return None
        torch.cuda.synchronize()
        t = [a.elapsed_time(b) for a, b in self._events]
        self._events = []
        return sum(t) / len(t)


class DDP(torch.nn.Module):
    """Drop-in for the reference's DDP subclass (:43-52): wraps `module`, forwards unknown attributes
    to it (`ddp_model.init_kwargs`, `.out_channels`, ...), keeps the `module.` state_dict prefix."""

    def __init__(self, module: torch.nn.Module, device_ids=None, find_unused_parameters: bool = True,
                 process_group=None, bucket_mb: float = 8.0, broadcast_parameters: bool = True,
                 broadcast_buffers: bool = True):
        super().__init__()
        self.module = module
        self.device_ids = device_ids
        self.find_unused_parameters = find_unused_parameters  # accepted for signature parity; nothing to search:
        # the engine computes every parameter gradient in one autograd node
        self.process_group = process_group
        self.broadcast_buffers = broadcast_buffers
        self.sync = GradSync(process_group, bucket_mb)
        if broadcast_parameters:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=process_group)
        object.__setattr__(module, "_tem_grad_sync", self.sync)

    def _sync_buffers(self):
        """torch DDP's `broadcast_buffers=True` (its default, which the reference keeps, multi_gpu_training.py:79): rank
        0's module buffers -- the running statistics of norm="BatchNorm" / "InstanceNormTrackStats" -- overwrite every
        other rank's at the start of each forward pass.  One flat broadcast per dtype."""
        by_dtype = {}
        for b in self.module.buffers():
            by_dtype.setdefault(b.dtype, []).append(b)
        for bufs in by_dtype.values():
            flat = torch.cat([b.detach().reshape(-1) for b in bufs])
            dist.broadcast(flat, src=0, group=self.process_group)
            pos = 0
            for b in bufs:
                n = b.numel()
                b.data.copy_(flat[pos:pos + n].view_as(b))
                pos += n

    def forward(self, *args, **kwargs):
        if self.broadcast_buffers and self.sync.world > 1 and next(self.module.buffers(), None) is not None:
            self._sync_buffers()
        return self.module(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)


def _create_data_loader(ds_callable, ds_kwargs, loader_kwargs, world_size, rank):
    """Dataset + DistributedSampler + DataLoader per rank (reference :27-40)."""
    ds = ds_callable(**ds_kwargs)
    loader_kwargs = dict(loader_kwargs)
    shuffle = loader_kwargs.pop("shuffle", False)
    sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world_size, rank=rank, shuffle=shuffle)
    loader = torch.utils.data.DataLoader(ds, sampler=sampler, **loader_kwargs)
    loader.shuffle = shuffle
    return loader


def _train_impl(rank, world_size, model_callable, model_kwargs, train_dataset_callable, train_dataset_kwargs,
                val_dataset_callable, val_dataset_kwargs, loader_kwargs, iterations, find_unused_parameters=True,
                optimizer_callable=None, optimizer_kwargs=None, lr_scheduler_callable=None, lr_scheduler_kwargs=None,
                trainer_callable=None, **kwargs):
    """Per-rank body (reference :55-104)."""
    assert "device" not in kwargs
    from .segmentation import default_segmentation_trainer
    print(f"Running DDP on rank {rank}.")
    setup(rank, world_size)
    model = model_callable(**model_kwargs).to(rank)
    ddp_model = DDP(model, device_ids=[rank], find_unused_parameters=find_unused_parameters)
    if optimizer_callable is not None:
        kwargs["optimizer"] = optimizer_callable(model.parameters(), **(optimizer_kwargs or {}))
        if lr_scheduler_callable is not None:
            kwargs["lr_scheduler"] = lr_scheduler_callable(kwargs["optimizer"], **(lr_scheduler_kwargs or {}))
    train_loader = _create_data_loader(train_dataset_callable, train_dataset_kwargs, loader_kwargs, world_size, rank)
    val_loader = _create_data_loader(val_dataset_callable, val_dataset_kwargs, loader_kwargs, world_size, rank)
    trainer_callable = trainer_callable or default_segmentation_trainer
    trainer = trainer_callable(model=ddp_model, train_loader=train_loader, val_loader=val_loader, device=rank,
                               rank=rank, **kwargs)
    trainer.fit(iterations=iterations)
    cleanup()


def train_multi_gpu(model_callable: Callable, model_kwargs: Dict[str, Any], train_dataset_callable: Callable,
                    train_dataset_kwargs: Dict[str, Any], val_dataset_callable: Callable,
                    val_dataset_kwargs: Dict[str, Any], loader_kwargs: Dict[str, Any], iterations: int,
                    find_unused_parameters: bool = True, optimizer_callable: Optional[Callable] = None,
                    optimizer_kwargs: Optional[Dict[str, Any]] = None, lr_scheduler_callable: Optional[Callable] = None,
                    lr_scheduler_kwargs: Optional[Dict[str, Any]] = None, trainer_callable: Optional[Callable] = None,
                    **kwargs) -> None:
    """Run data-parallel training on all GPUs of this node (reference :107-190; same arguments)."""
    world_size = torch.cuda.device_count()
    mp.spawn(_spawn_with_kwargs,
             args=(world_size, model_callable, model_kwargs, train_dataset_callable, train_dataset_kwargs,
                   val_dataset_callable, val_dataset_kwargs, loader_kwargs, iterations, find_unused_parameters,
                   optimizer_callable, optimizer_kwargs, lr_scheduler_callable, lr_scheduler_kwargs,
                   trainer_callable, kwargs),
             nprocs=world_size, join=True)


def _spawn_with_kwargs(rank, world_size, *args):
    *pos, kwargs = args
    _train_impl(rank, world_size, *pos, **kwargs)
