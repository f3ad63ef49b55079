"""Train UNet2d(1, 2, depth=2, initial_features=4) for 4 iterations with THIS repo's trainer on the GPU and leave the
checkpoint folder under gpurun_out/ckpt_tiny (it is then committed as tests/golden/ckpt_tiny/ and read back by the
REFERENCE's load_model / DefaultTrainer.from_checkpoint in the build container: tests/test_reference_reads_checkpoint.py)."""
import os
import shutil
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch_em_amd  # noqa: E402
from torch_em_amd.model import UNet2d  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "ckpt_tiny")
shutil.rmtree(out, ignore_errors=True)
g = torch.Generator().manual_seed(5)
x = torch.randn(4, 1, 32, 32, generator=g)
y = (torch.rand(4, 2, 32, 32, generator=g) > 0.5).float()
loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(x, y), batch_size=2, shuffle=False)
torch.manual_seed(0)
model = UNet2d(1, 2, depth=2, initial_features=4)
trainer = torch_em_amd.default_segmentation_trainer("tiny", model, loader, loader, device="cuda", logger=None,
                                                    mixed_precision=False, save_root=out)
trainer.fit(iterations=4)
with torch.no_grad():
    model.eval()
    pred = model(x[:2].cuda()).cpu()
torch.save({"x": x[:2], "pred": pred}, os.path.join(out, "checkpoints", "tiny", "io.pt"))
print(sorted(os.listdir(os.path.join(out, "checkpoints", "tiny"))))
