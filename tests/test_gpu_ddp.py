"""GPU, world_size 2 on ONE device over gloo: the real engine backward + GradSync + fused AdamW under the DDP wrapper.
(RCCL needs one GPU per rank; the driver's 8-GPU run covers that transport, this covers everything above it.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d
    from torch_em_amd.multi_gpu_training import DDP, cleanup, setup
    from torch_em_amd.optim import FusedAdamW
    setup(rank, world, backend="gloo", port=port)
    try:
        dev = "cuda:0"
        torch.manual_seed(0)
        net = UNet3d(1, 2, depth=2, initial_features=32).to(dev)
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(1, 1, 16, 16, 16, generator=g).to(dev)
        y = (torch.rand(1, 2, 16, 16, 16, generator=g) > 0.5).float().to(dev)
        # local gradient without any exchange
        DiceLoss()(net(x), y).backward()
        local = torch.cat([p.grad.flatten() for p in net.parameters()]).clone()
        both = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        expect = sum(both) / world
        net.zero_grad()
        ddp = DDP(net, device_ids=[0])
        opt = FusedAdamW(net.parameters(), lr=1e-3)
        before = torch.cat([p.detach().flatten() for p in net.parameters()]).clone()
        opt.zero_grad()
        DiceLoss()(ddp(x), y).backward()
        got = torch.cat([p.grad.flatten() for p in net.parameters()])
        err = float((got - expect).norm() / expect.norm())
        opt.step()
        after = torch.cat([p.detach().flatten() for p in net.parameters()])
        allp = [torch.empty_like(after) for _ in range(world)]
        dist.all_gather(allp, after)
        q.put((rank, err, bool(torch.equal(allp[0], allp[1])), float((after - before).abs().max())))
    finally:
        cleanup()


def test_ddp_two_ranks_one_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
    for rank, err, same, moved in res:
        assert err < 1e-5, (rank, err)      # averaged gradient == mean of the two local gradients
        assert same                          # both ranks hold bit-identical parameters after the step
        assert moved > 0


def _worker_bn(rank, world, port, q):
    """norm="BatchNorm" under DDP: rank 0's running statistics are what every rank holds when a forward pass starts."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d
    from torch_em_amd.multi_gpu_training import DDP, cleanup, setup
    setup(rank, world, backend="gloo", port=port)
    try:
        dev = "cuda:0"
        torch.manual_seed(0)
        net = UNet3d(1, 2, depth=2, initial_features=16, norm="BatchNorm").to(dev)
        ddp = DDP(net, device_ids=[0])
        g = torch.Generator().manual_seed(7 + rank)
        x = torch.randn(2, 1, 16, 16, 16, generator=g).to(dev)
        y = (torch.rand(2, 2, 16, 16, 16, generator=g) > 0.5).float().to(dev)
        DiceLoss()(ddp(x), y).backward()          # the ranks now hold DIFFERENT running statistics (different batches)
        mine = torch.cat([b.detach().float().reshape(-1) for b in net.buffers()]).clone()
        allb = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allb, mine)
        differ = not torch.equal(allb[0], allb[1])
        seen = {}
        def snapshot(m, a):  # (returns None: a pre-hook's return value would replace the input)
            seen["b"] = torch.cat([b.detach().float().reshape(-1) for b in m.buffers()]).clone()

        h = net.register_forward_pre_hook(snapshot)
        ddp(x)                                    # second forward: starts from rank 0's buffers on every rank
        h.remove()
        q.put((rank, differ, bool(torch.equal(seen["b"], allb[0]))))
    finally:
        cleanup()


def test_ddp_broadcasts_buffers_every_forward():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_bn, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
    for rank, differ, synced in res:
        assert differ          # the test is meaningful: local statistics had diverged
        assert synced, rank    # and the next forward saw rank 0's


def _worker_nccl(port, q):
    """ONE rank over the real RCCL communicator: init_process_group("nccl"), GradSync's in-place ReduceOp.AVG ranges on
    the collective stream, finish() before the fused optimizer (N > 1 on hardware is the driver's scaling run)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d
    from torch_em_amd.multi_gpu_training import DDP, cleanup, setup
    from torch_em_amd.optim import FusedAdamW
    setup(0, 1, backend="nccl", port=port)
    try:
        dev = "cuda:0"
        torch.manual_seed(0)
        net = UNet3d(1, 2, depth=2, initial_features=32).to(dev)
        x = torch.randn(1, 1, 32, 32, 32).to(dev)
        y = (torch.rand(1, 2, 32, 32, 32) > 0.5).float().to(dev)
        DiceLoss()(net(x), y).backward()
        local = torch.cat([p.grad.flatten() for p in net.parameters()]).clone()
        net.zero_grad()
        ddp = DDP(net, device_ids=[0], bucket_mb=0.25)   # small ranges: several overlapped collectives per step
        opt = FusedAdamW(net.parameters(), lr=1e-3)
        opt.zero_grad()
        DiceLoss()(ddp(x), y).backward()
        got = torch.cat([p.grad.flatten() for p in net.parameters()])
        before = torch.cat([p.detach().flatten() for p in net.parameters()]).clone()
        opt.step()
        torch.cuda.synchronize()
        after = torch.cat([p.detach().flatten() for p in net.parameters()])
        q.put((bool(torch.equal(got, local)), float((after - before).abs().max()), bool(ddp.sync._avg)))
    finally:
        cleanup()


def test_ddp_one_rank_over_rccl():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_nccl, args=(_free_port(), q))
    p.start()
    same, moved, avg = q.get(timeout=300)
    p.join(60)
    assert avg            # the NCCL (= RCCL) backend averages in the collective
    assert same           # AVG over one rank is the identity, bit for bit
    assert moved > 0


def _worker_nccl_graph(port, q):
    """The data-parallel step as ONE HIP graph: forward, backward, GradSync's RCCL all-reduces (collective stream forked
    from / joined to the capturing stream) and the fused optimizer; replays must equal the eager DDP steps bit for bit."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from torch_em_amd.graph import GraphedTrainStep
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d
    from torch_em_amd.multi_gpu_training import DDP, cleanup, setup
    from torch_em_amd.optim import FusedAdamW
    setup(0, 1, backend="nccl", port=port)
    try:
        dev = "cuda:0"
        g = torch.Generator().manual_seed(3)
        xs = [torch.randn(1, 1, 32, 32, 32, generator=g).to(dev) for _ in range(4)]
        ys = [(torch.rand(1, 2, 32, 32, 32, generator=g) > 0.5).float().to(dev) for _ in range(4)]
        loss_fn = DiceLoss()

        def make():
            torch.manual_seed(0)
            net = UNet3d(1, 2, depth=2, initial_features=16).to(dev)
            return net, DDP(net, device_ids=[0], bucket_mb=0.05), FusedAdamW(net.parameters(), lr=1e-3)

        net0, ddp0, opt0 = make()
        losses0 = []
        for x, y in zip(xs, ys):
            opt0.zero_grad()
            loss = loss_fn(ddp0(x), y)
            loss.backward()
            opt0.step()
            losses0.append(float(loss))
        ncoll = ddp0.sync.stats["n_collectives"]
        net1, ddp1, opt1 = make()
        step = GraphedTrainStep(ddp1, loss_fn, opt1, xs[0], ys[0])
        losses1 = [float(step(x, y)[1]) for x, y in zip(xs, ys)]
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(net0.state_dict().values(), net1.state_dict().values()))
        q.put((same, losses0 == losses1, ncoll, step.replays))
    finally:
        cleanup()


def test_ddp_step_as_one_hip_graph_over_rccl():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_nccl_graph, args=(_free_port(), q))
    p.start()
    same, same_loss, ncoll, replays = q.get(timeout=300)
    p.join(60)
    assert ncoll >= 2          # several overlapped collectives per step were part of the capture
    assert replays == 4 and same_loss and same
