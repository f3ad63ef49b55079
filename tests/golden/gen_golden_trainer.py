"""Golden vectors made by RUNNING THE REFERENCE's trainer and its own affinity test definitions (build container only).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/gen_golden_trainer.py

G7  (SURVEY.md 8c): `torch_em.default_segmentation_trainer(...).fit(iterations=8)` -- the reference's DefaultTrainer,
    AdamW, ReduceLROnPlateau, DiceLoss -- on a fixed synthetic batch set with UNet2d(1, 2, depth=2, initial_features=4):
    loss of every iteration, learning rate, validation metric / loss of every epoch, counters, the parameters before
    and after, and the key lists of the checkpoint it wrote.
G8  the affinity targets of the reference's OWN brute-force definitions (`affs_brute_force`,
    `affs_brute_force_with_mask`, test/transform/test_label_transforms.py:5-55; that module imports numpy only), called
    here on seeded label images, plus the channel order / inversion / mask convention of
    `AffinityTransform.__call__` (transform/label.py:299-325) applied to them.
G9  a tiny checkpoint written by THIS repo's trainer format is read by the reference's `load_model`
    (util/util.py:408-460) and `DefaultTrainer.from_checkpoint` (trainer/default_trainer.py:288-330): see
    gen_checkpoint_compat() -- it needs a checkpoint produced on a GPU box and is therefore run separately
    (tests/golden/README_checkpoint.md records the result).

`import torch_em` needs packages this image lacks (imageio, skimage, kornia, ...): a meta-path finder serves empty
stand-in modules for exactly those imports (SURVEY.md 8c route B).  Nothing of the trainer path touches them.
The fixtures are data (inputs + what the reference computed); no reference source is copied.
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
MISSING = ("imageio", "elf", "skimage", "torchvision", "kornia", "bioimage_cpp", "tifffile", "h5py", "mrcfile", "natsort",
           "tensorboard", "torch_scatter", "zarr", "z5py", "nifty", "vigra", "pandas_stub_never")


class _DummyMeta(type):
    """Class attributes of a dummy class are dummy classes again (kornia.augmentation.AugmentationBase3D as a base)."""

    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _dummy(name)


def _dummy(name):
    return _DummyMeta(name, (), {"__init__": lambda s, *a, **k: None, "__call__": lambda s, *a, **k: None})


class _Anything(types.ModuleType):
    """A module whose every attribute is a dummy class (whose attributes are dummy classes ...)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = f"{self.__name__}.{name}"
        if sub in sys.modules:
            return sys.modules[sub]
        return _dummy(name)


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in MISSING:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        mod = _Anything(spec.name)
        mod.__path__ = []
        return mod

    def exec_module(self, module):
        pass


def import_reference():
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF_ROOT)
    import torch_em  # noqa: F401
    return sys.modules["torch_em"]


def make_batches(seed, n, shape=(1, 32, 32)):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, *shape, generator=g)
    y = (torch.rand(n, 2, *shape[1:], generator=g) > 0.5).float()
    return x, y


class Recorder:
    """Stands in for the TensorboardLogger: same constructor / method signatures, keeps the scalars."""
    log = None

    def __init__(self, trainer, save_root, **unused):
        Recorder.log = {"train_loss": [], "lr": [], "val_metric": [], "val_loss": [], "val_iter": []}

    def log_train(self, step, loss, lr, x, y, prediction, log_gradients=False):
        Recorder.log["train_loss"].append(float(loss.item()))
        Recorder.log["lr"].append(float(lr))

    def log_validation(self, step, metric, loss, x, y, prediction):
        Recorder.log["val_metric"].append(float(metric))
        Recorder.log["val_loss"].append(float(loss))
        Recorder.log["val_iter"].append(int(step))


def gen_trainer(torch_em, tmp):
    from torch_em.model import UNet2d
    torch.manual_seed(0)
    model = UNet2d(1, 2, depth=2, initial_features=4)
    sd0 = {f"sd0.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()}
    xt, yt = make_batches(11, 8)   # 4 iterations per epoch at batch size 2
    xv, yv = make_batches(12, 4)
    train = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(xt, yt), batch_size=2, shuffle=False)
    val = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(xv, yv), batch_size=2, shuffle=False)
    trainer = torch_em.default_segmentation_trainer(
        name="g7", model=model, train_loader=train, val_loader=val, learning_rate=1e-2, device="cpu",
        mixed_precision=False, logger=Recorder, save_root=tmp, compile_model=False)
    trainer.fit(iterations=8)
    ckpt = torch.load(os.path.join(tmp, "checkpoints", "g7", "latest.pt"), weights_only=False)
    sd1 = {f"sd1.{k}": v.detach().cpu().numpy().copy() for k, v in trainer.model.state_dict().items()}
    log = Recorder.log
    out = dict(xt=xt.numpy(), yt=yt.numpy(), xv=xv.numpy(), yv=yv.numpy(), learning_rate=np.float64(1e-2),
               train_loss=np.array(log["train_loss"], dtype=np.float64), lr=np.array(log["lr"], dtype=np.float64),
               val_metric=np.array(log["val_metric"], dtype=np.float64), val_loss=np.array(log["val_loss"], dtype=np.float64),
               val_iter=np.array(log["val_iter"]), iteration=np.int64(ckpt["iteration"]), epoch=np.int64(ckpt["epoch"]),
               best_epoch=np.int64(ckpt["best_epoch"]), best_metric=np.float64(ckpt["best_metric"]),
               current_metric=np.float64(ckpt["current_metric"]),
               ckpt_keys=np.array(sorted(ckpt.keys())), init_keys=np.array(sorted(ckpt["init"].keys())),
               optimizer_state_keys=np.array(sorted(ckpt["optimizer_state"].keys())),
               model_class=np.array(ckpt["init"]["model_class"]), loss_class=np.array(ckpt["init"]["loss_class"]),
               optimizer_class=np.array(ckpt["init"]["optimizer_class"]),
               lr_scheduler_class=np.array(str(ckpt["init"]["lr_scheduler_class"])), **sd0, **sd1)
    np.savez_compressed(os.path.join(OUT, "g7_trainer_unet2d.npz"), **out)
    print("G7:", [f"{v:.6f}" for v in log["train_loss"]], log["val_metric"], "iteration", ckpt["iteration"],
          "epoch", ckpt["epoch"])


def gen_trainer_amp(torch_em, tmp):
    """G7b: the reference's MIXED-PRECISION loop -- torch.autocast(float16) + torch.GradScaler
    (trainer/default_trainer.py:134-142, 789-803) -- and its fp32 loop on the same data at MFMA-eligible widths
    (UNet2d(1, 2, depth=2, initial_features=32)), 8 iterations each on the CPU: loss per iteration, validation metric, the
    scaler's final scale (autocast's fp16 gradients overflow in step 0 of this run: the step is skipped, 65536 -> 32768)."""
    from torch_em.model import UNet2d
    xt, yt = make_batches(11, 8)
    xv, yv = make_batches(12, 4)
    train = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(xt, yt), batch_size=2, shuffle=False)
    val = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(xv, yv), batch_size=2, shuffle=False)
    out = dict(xt=xt.numpy(), yt=yt.numpy(), xv=xv.numpy(), yv=yv.numpy(), learning_rate=np.float64(1e-3))
    for tag, dt in (("fp32", None), ("amp", "float16")):
        torch.manual_seed(0)
        model = UNet2d(1, 2, depth=2, initial_features=32)
        if tag == "fp32":
            out.update({f"sd0.{k}": v.detach().numpy().copy() for k, v in model.state_dict().items()})
        trainer = torch_em.default_segmentation_trainer(
            name="g7b" + tag, model=model, train_loader=train, val_loader=val, learning_rate=1e-3, device="cpu",
            mixed_precision=dt is not None, mixed_precision_dtype=dt, logger=Recorder, save_root=tmp, compile_model=False)
        trainer.fit(iterations=8)
        log = Recorder.log
        out[f"{tag}_train_loss"] = np.array(log["train_loss"])
        out[f"{tag}_val_metric"] = np.array(log["val_metric"])
        if dt is not None:
            out["amp_final_scale"] = np.float64(trainer.scaler.get_scale())
        print("G7b", tag, [f"{v:.5f}" for v in log["train_loss"]], log["val_metric"])
    np.savez_compressed(os.path.join(OUT, "g7b_trainer_amp_unet2d.npz"), **out)


def gen_trainer_bf16(torch_em, tmp):
    """G7c: the reference's loop with mixed_precision_dtype="bfloat16" -- torch.autocast(bfloat16), NO GradScaler
    (trainer/default_trainer.py:134-142) -- on the data / initial weights of G7b (same seeds), 8 iterations on the CPU."""
    from torch_em.model import UNet2d
    xt, yt = make_batches(11, 8)
    xv, yv = make_batches(12, 4)
    train = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(xt, yt), batch_size=2, shuffle=False)
    val = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(xv, yv), batch_size=2, shuffle=False)
    torch.manual_seed(0)
    model = UNet2d(1, 2, depth=2, initial_features=32)
    trainer = torch_em.default_segmentation_trainer(
        name="g7c", model=model, train_loader=train, val_loader=val, learning_rate=1e-3, device="cpu",
        mixed_precision=True, mixed_precision_dtype="bfloat16", logger=Recorder, save_root=tmp, compile_model=False)
    assert trainer.scaler is not None and not trainer.scaler.is_enabled()   # created disabled for bfloat16 (:138-140)
    trainer.fit(iterations=8)
    log = Recorder.log
    np.savez_compressed(os.path.join(OUT, "g7c_trainer_bf16_unet2d.npz"), bf16_train_loss=np.array(log["train_loss"]),
                        bf16_val_metric=np.array(log["val_metric"]))
    print("G7c bf16", [f"{v:.5f}" for v in log["train_loss"]], log["val_metric"])


def gen_affinities():
    spec = importlib.util.spec_from_file_location("ref_test_label_transforms",
                                                  os.path.join(REF_ROOT, "test/transform/test_label_transforms.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)   # numpy + unittest only
    rng = np.random.RandomState(3)
    out = {}
    cases = [("a", (16, 20), [[-1, 0], [0, -1], [-3, 0], [0, -3]], 5),
             ("b", (24, 17), [[-1, 0], [0, -1], [-2, 2], [4, -1], [0, 5]], 4),
             ("c", (9, 31), [[1, 0], [0, 1], [2, 3]], 3)]
    for tag, shape, offsets, nlab in cases:
        seg = rng.randint(0, nlab, size=shape).astype("uint32")
        out[f"{tag}_seg"] = seg
        out[f"{tag}_offsets"] = np.array(offsets, dtype=np.int64)
        # what the reference's tests require AffinityTransform(offsets)(seg) to equal (test_affinities, :68-79) ...
        out[f"{tag}_affs"] = mod.affs_brute_force(seg, offsets)
        # ... AffinityTransform(offsets, ignore_label=0, add_mask=True) (test_affinities_with_mask, :81-97) ...
        out[f"{tag}_affs_ignore0"], out[f"{tag}_mask_ignore0"] = mod.affs_brute_force_with_mask(seg, offsets)
        # ... and with include_ignore_transitions=True (test_affinities_with_ignore_transition, :99-116)
        out[f"{tag}_affs_trans"], out[f"{tag}_mask_trans"] = mod.affs_brute_force_with_mask(seg, offsets,
                                                                                           mask_bg_transition=False)
    np.savez_compressed(os.path.join(OUT, "g8_affinities_bruteforce.npz"), **out)
    print("G8:", {k: v.shape for k, v in out.items() if k.endswith("_affs")})


if __name__ == "__main__":
    import sys
    import tempfile
    torch.set_num_threads(4)
    torch.use_deterministic_algorithms(True)
    if "--bf16-only" in sys.argv:   # G7c was added later; the other fixtures are not regenerated
        te = import_reference()
        with tempfile.TemporaryDirectory() as tmp:
            gen_trainer_bf16(te, tmp)
        sys.exit(0)
    gen_affinities()
    te = import_reference()
    with tempfile.TemporaryDirectory() as tmp:
        gen_trainer(te, tmp)
    with tempfile.TemporaryDirectory() as tmp:
        gen_trainer_amp(te, tmp)
    with tempfile.TemporaryDirectory() as tmp:
        gen_trainer_bf16(te, tmp)
