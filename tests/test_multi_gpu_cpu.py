"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (GradSync / DDP wrapper) that the
N>1 path uses, exercised without a GPU on a fake gradient arena."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from torch_em_amd.multi_gpu_training import DDP, GradSync, cleanup, setup
    setup(rank, world, backend="gloo", port=port)
    try:
        n = 10000
        flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
        sync = GradSync(bucket_mb=0.01)            # ~2.6k elements per bucket: several launches
        # ranges arrive out of order and leave a hole [3000, 3500) that nobody announces
        for lo, hi in ((8000, 10000), (6000, 8000), (3500, 6000), (0, 1000), (1000, 3000)):
            sync.ready(flat, lo, hi)
        sync.finish(flat)
        expect = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        ok_sync = torch.allclose(flat, expect)
        # DDP wrapper: parameter broadcast from rank 0, attribute passthrough, state_dict prefix
        torch.manual_seed(rank)
        from torch_em_amd.model import UNet3d
        net = UNet3d(1, 2, depth=1, initial_features=4)
        ddp = DDP(net, device_ids=None)
        w = net.out_conv.weight.detach().clone()
        gathered = [torch.empty_like(w) for _ in range(world)]
        dist.all_gather(gathered, w)
        ok_bcast = all(torch.equal(gathered[0], t) for t in gathered)
        ok_attr = ddp.out_channels == 2 and ddp.init_kwargs["depth"] == 1 and \
            all(k.startswith("module.") for k in ddp.state_dict())
        # per-forward buffer broadcast (torch DDP's broadcast_buffers=True): rank 0's running statistics win
        bn = torch.nn.Sequential(torch.nn.Conv3d(1, 3, 1), torch.nn.BatchNorm3d(3))
        dbn = DDP(bn, device_ids=None)
        with torch.no_grad():
            bn[1].running_mean.fill_(float(rank + 1))
            bn[1].num_batches_tracked.fill_(10 * (rank + 1))
        dbn.eval()
        dbn(torch.zeros(1, 1, 2, 2, 2))
        ok_buf = bool((bn[1].running_mean == 1.0).all()) and int(bn[1].num_batches_tracked) == 10
        q.put((rank, ok_sync, ok_bcast, ok_attr, getattr(net, "_tem_grad_sync", None) is ddp.sync, ok_buf))
    finally:
        cleanup()


def test_gradsync_and_ddp_wrapper_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(r[0] for r in res) == [0, 1]
    for r in res:
        assert all(r[1:]), r


def test_arena_layout_and_flat_grads_detection():
    from torch_em_amd.arena import ParamArena, arena_layout
    from torch_em_amd.model import UNet3d
    net = UNet3d(1, 2, depth=1, initial_features=4)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    ar = ParamArena(net)
    assert ar.is_current() and all(o % 4 == 0 for o, _ in ar.offsets.values())
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k])
    # gradients that are consecutive views of one flat buffer are detected as an arena ...
    offs, total = arena_layout(ar.params)
    flat = torch.randn(total)
    for p in ar.params:
        o, n = offs[id(p)]
        p.grad = flat[o:o + n].view(p.shape).detach()
    assert ar.grads_flat() is None          # nobody registered an arena
    ar.params[0]._tem_grad_flat = flat      # what engine.UNetFunction.backward does
    g = ar.grads_flat()
    assert g is not None and g.data_ptr() == flat.data_ptr() and g.numel() == total
    # ... and anything else is not
    ar.params[1].grad = torch.randn_like(ar.params[1])
    assert ar.grads_flat() is None
    net.to(torch.float64)  # replaces the storages
    assert not ar.is_current()
