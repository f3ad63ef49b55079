"""cfg 4 rehearsed end to end at the world size this box has: `train_multi_gpu()` ITSELF (reference
torch_em/multi_gpu_training.py:107-190, scripts/run_multi_gpu_train.py:30-37) -- mp.spawn -> setup (RCCL) -> DDP ->
_create_data_loader + DistributedSampler -> default_segmentation_trainer(rank=...) -> fit -> rank-0 checkpoint.

On a one-GPU box world_size = device_count() = 1, so the assertions that need a second rank (bit-identical parameters on
every rank, disjoint samples) are guarded by the number of rank records found and run by themselves on the first
multi-GPU box; everything else (the entry point returns, rank 0 alone writes the checkpoint, the `module.` key prefix, the
checkpoint loads into a bare UNet3d, the iteration count) is asserted at any world size."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

N_SAMPLES, SHAPE, ITERATIONS = 16, (16, 16, 16), 4


class _IndexedVolumes(torch.utils.data.Dataset):
    """Synthetic (raw, 2-channel target) volumes, a function of the sample index alone; remembers what it was asked for."""

    def __init__(self, n, shape, seed):
        self.n, self.shape, self.seed, self.seen = n, tuple(shape), seed, []

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        self.seen.append(int(i))
        g = torch.Generator().manual_seed(self.seed + int(i))
        x = torch.randn((1,) + self.shape, generator=g)
        y = (torch.rand((2,) + self.shape, generator=g) > 0.5).float()
        return x, y


def _dataset(n, shape, seed):
    return _IndexedVolumes(n, shape, seed)


def _recording_trainer(**kwargs):
    """trainer_callable: the default factory with a trainer that leaves one record per RANK next to the checkpoints."""
    from torch_em_amd.segmentation import default_segmentation_trainer
    from torch_em_amd.trainer import DefaultTrainer

    class Recording(DefaultTrainer):
        saves = 0

        def save_checkpoint(self, name, *a, **k):
            if name == "latest":
                Recording.saves += 1
            return super().save_checkpoint(name, *a, **k)

        def fit(self, *a, **k):
            super().fit(*a, **k)
            import torch.distributed as dist
            flat = torch.cat([p.detach().flatten() for p in self.model.parameters()]).cpu()
            torch.save({"rank": self.rank, "world": dist.get_world_size(), "backend": dist.get_backend(),
                        "device": str(next(self.model.parameters()).device), "params": flat,
                        "seen": list(self.train_loader.dataset.seen), "iteration": self._iteration,
                        "graphed": self._graphed is not None, "graph_why": self._graph_why, "saves": Recording.saves,
                        "collectives": dict(self.model.sync.stats)},
                       os.path.join(self.save_root, f"rank{self.rank}.pt"))

    return default_segmentation_trainer(trainer_class=Recording, **kwargs)


@pytest.mark.parametrize("hip_graph", [False, True])
def test_train_multi_gpu_end_to_end(tmp_path, hip_graph):
    from torch_em_amd.model import UNet3d
    from torch_em_amd.multi_gpu_training import train_multi_gpu
    world = torch.cuda.device_count()
    assert world >= 1
    os.environ["MASTER_PORT"] = str(_free_port())     # inherited by the spawned ranks (setup() only sets a default)
    save_root = str(tmp_path)
    model_kwargs = dict(in_channels=1, out_channels=2, depth=2, initial_features=32)
    train_multi_gpu(
        UNet3d, model_kwargs, _dataset, dict(n=N_SAMPLES, shape=SHAPE, seed=1000), _dataset,
        dict(n=2 * world, shape=SHAPE, seed=5000), loader_kwargs={"batch_size": 2, "shuffle": True},
        iterations=ITERATIONS, trainer_callable=_recording_trainer, name="ddp-rehearsal", save_root=save_root,
        mixed_precision=False, hip_graph=hip_graph)
    os.environ.pop("MASTER_PORT", None)

    ckpt_dir = os.path.join(save_root, "checkpoints", "ddp-rehearsal")
    latest = os.path.join(ckpt_dir, "latest.pt")
    assert os.path.exists(latest)
    ck = torch.load(latest, weights_only=False)
    assert ck["iteration"] == ITERATIONS and ck["init"]["rank"] == 0          # written by rank 0 ...
    keys = list(ck["model_state"])
    assert keys and all(k.startswith("module.") for k in keys)               # ... from the DDP wrapper, as the reference's
    bare = UNet3d(**model_kwargs)
    assert keys == ["module." + k for k in bare.state_dict()]
    bare.load_state_dict({k[len("module."):]: v for k, v in ck["model_state"].items()})   # strict

    recs = sorted((torch.load(os.path.join(save_root, f), weights_only=False)
                   for f in os.listdir(save_root) if f.startswith("rank")), key=lambda r: r["rank"])
    assert [r["rank"] for r in recs] == list(range(world))
    flat0 = torch.cat([bare.state_dict()[k].flatten() for k, _ in bare.named_parameters()])
    assert torch.equal(flat0, recs[0]["params"])                               # the checkpoint IS rank 0's final model
    for r in recs:
        assert r["world"] == world and r["backend"] == "nccl" and r["device"] == f"cuda:{r['rank']}"
        assert r["iteration"] == ITERATIONS and r["saves"] >= 1               # every rank CALLS save_checkpoint; one writes
        assert r["graphed"] == hip_graph, r["graph_why"]
        nbytes = recs[0]["params"].numel() * 4                                 # the whole gradient arena was exchanged
        assert nbytes <= r["collectives"]["bytes"] <= nbytes + 4096          # (+ the arena's alignment padding)
        assert torch.equal(r["params"], recs[0]["params"])                     # bit-identical parameters on every rank
    # DistributedSampler: per epoch the ranks see disjoint samples that together cover the dataset (the loader of every
    # rank has N / world / batch_size batches, so 4 iterations span ITERATIONS * 2 * world / N epochs)
    per_epoch = N_SAMPLES // world
    n_epochs = -(-ITERATIONS * 2 // per_epoch)
    for e in range(n_epochs):
        parts = [set(r["seen"][e * per_epoch:(e + 1) * per_epoch]) for r in recs]
        full = e < n_epochs - 1 or (ITERATIONS * 2) % per_epoch == 0
        if full:
            assert set().union(*parts) == set(range(N_SAMPLES))
        for a in range(world):
            for b in range(a + 1, world):
                assert not (parts[a] & parts[b])
    if world == 1:
        assert len(recs[0]["seen"]) >= ITERATIONS * 2


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port
