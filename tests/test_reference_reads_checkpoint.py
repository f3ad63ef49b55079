"""Build container only (needs /root/reference; skipped elsewhere): the REFERENCE's own `torch_em.util.load_model`
(util/util.py:408-460) and `DefaultTrainer.from_checkpoint` (trainer/default_trainer.py:288-330) read a checkpoint that
THIS repo's trainer wrote on an MI355X (tests/golden/ckpt_tiny, made by scripts/make_tiny_checkpoint.py) -- SURVEY.md
8(f)-4, checkpoint wire-format compatibility.  Runs in a subprocess: importing the reference package needs stand-in
modules for packages this image lacks (tests/golden/gen_golden_trainer.py, route B)."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT

CKPT = os.path.join(GOLDEN, "ckpt_tiny", "checkpoints", "tiny")
CODE = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests", "golden"))
import torch
import gen_golden_trainer as gg
torch_em = gg.import_reference()
from torch_em.util import load_model
from torch_em.trainer import DefaultTrainer
ck = torch.load(os.path.join({ckpt!r}, "latest.pt"), weights_only=False, map_location="cpu")
model = load_model({ckpt!r}, name="latest", device="cpu")          # class + kwargs from the `init` record, then the state
model = getattr(model, "_orig_mod", model)   # the reference wraps the model in torch.compile (compile_model=None: auto)
assert type(model).__name__ == "UNet2d" and type(model).__module__.startswith("torch_em_amd"), type(model)
sd = {{k: v for k, v in model.state_dict().items()}}
assert sorted(sd) == sorted(ck["model_state"]) and all(torch.equal(sd[k].cpu(), ck["model_state"][k].cpu()) for k in sd)
# the same state in the REFERENCE's UNet2d reproduces the prediction our engine stored next to the checkpoint
from torch_em.model import UNet2d
kw = {{k: v for k, v in ck["init"]["model_kwargs"].items() if not k.endswith("_impl")}}   # the reference's own block classes
ref_model = UNet2d(**kw)
ref_model.load_state_dict(ck["model_state"])
ref_model.eval()
io = torch.load(os.path.join({ckpt!r}, "io.pt"), weights_only=False)
with torch.no_grad():
    err = float((ref_model(io["x"]) - io["pred"]).abs().max() / io["pred"].abs().max())
assert err < 1e-4, err
trainer = DefaultTrainer.from_checkpoint({ckpt!r}, name="latest", device="cpu")
assert trainer._iteration == ck["iteration"] == 4 and trainer._epoch == ck["epoch"]
assert type(trainer.optimizer).__name__ == "FusedAdamW" and len(trainer.optimizer.state_dict()["state"]) == len(sd)
print("REFERENCE_READS_CHECKPOINT_OK", err)
"""


@pytest.mark.skipif(not os.path.isdir("/root/reference/torch_em"), reason="the reference only exists in the build container")
@pytest.mark.skipif(not os.path.exists(os.path.join(CKPT, "latest.pt")), reason="tests/golden/ckpt_tiny not generated yet")
def test_reference_loads_our_checkpoint():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    res = subprocess.run([sys.executable, "-c", CODE.format(root=ROOT, ckpt=CKPT)], capture_output=True, text=True, env=env,
                         cwd="/tmp", timeout=600)
    assert res.returncode == 0 and "REFERENCE_READS_CHECKPOINT_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
