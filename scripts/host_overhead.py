"""Host time per training step: how long the Python thread needs to ENQUEUE one step (no synchronisation inside the timed
region) versus how long the GPU needs to run it, eager and as a replayed HIP graph (torch_em_amd/graph.py).
    python scripts/host_overhead.py [--size 128] [--batch 2] [--steps 20] [--precision split16|amp]
Prints one JSON line.  The step is host-bound when host_ms >= gpu_ms."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from torch_em_amd.graph import GraphedTrainStep  # noqa: E402
from torch_em_amd.loss import DiceLoss  # noqa: E402
from torch_em_amd.model import UNet3d  # noqa: E402
from torch_em_amd.model import engine  # noqa: E402
from torch_em_amd.optim import FusedAdamW  # noqa: E402


def measure(fn, steps):
    for _ in range(3):
        fn()
    # host: the enqueue of ONE step into an idle queue (with several steps in flight the hardware queue fills up and the
    # enqueue call blocks for as long as the GPU needs: that would measure the GPU again)
    host = 0.0
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        host += time.perf_counter() - t0
    torch.cuda.synchronize()
    # gpu: back-to-back steps between two events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return host / steps * 1e3, e0.elapsed_time(e1) / steps


def run(size=128, batch=2, steps=20, precision=None):
    """-> {"workload", "eager": {host_ms, gpu_ms}, "hip_graph": {host_ms, gpu_ms}} for the cfg-2 network at batch x size^3"""
    prev = engine.PRECISION
    if precision:
        engine.set_precision(precision)
    try:
        torch.manual_seed(0)
        model = UNet3d(1, 2, depth=4, initial_features=32).to("cuda")
        opt = FusedAdamW(model.parameters(), lr=1e-4)
        loss_fn = DiceLoss()
        x = torch.randn(batch, 1, size, size, size, device="cuda")
        y = (torch.rand(batch, 2, size, size, size, device="cuda") > 0.5).float()

        def eager():
            opt.zero_grad()
            loss = loss_fn(model(x), y)
            loss.backward()
            opt.step()

        eh, eg = measure(eager, steps)
        step = GraphedTrainStep(model, loss_fn, opt, x, y)
        gh, gg = measure(lambda: step(x, y), steps)
        return {"workload": f"UNet3d(1->2, 32 features, depth 4) {batch}x1x{size}^3, {engine.PRECISION}",
                "eager": {"host_ms": round(eh, 3), "gpu_ms": round(eg, 3)},
                "hip_graph": {"host_ms": round(gh, 3), "gpu_ms": round(gg, 3)}}
    finally:
        engine.set_precision(prev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--precision", default=None)
    a = ap.parse_args()
    print(json.dumps(run(a.size, a.batch, a.steps, a.precision)))


if __name__ == "__main__":
    main()
