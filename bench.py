#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X: voxels/s of one training step
(zero_grad -> forward -> DiceLoss -> backward -> AdamW) of UNet3d(1->2, initial_features=32,
depth=4, InstanceNorm) on synthetic 2x1x128^3 batches per GPU (cfg 2; cfg 4 = the same per
rank under data parallelism, weak scaling).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the fp32-MFMA 3x3x3
implicit-GEMM convolution), its duration measured live with HIP events on the launch stream
inside the timed region; `cpu_baseline` times the oracle (oracle/unet_ref.py, a torch-CPU
restatement pinned to the reference) on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md:41
PEAK_BF16_MFMA_TFLOPS = 2500.0  # ibid. :42 (dense)
PEAK_HBM_TBS = 8.0              # ibid. :35
STEP_GFLOP = 5700.8             # conv fwd+bwd of cfg 2 per GPU
STEP_GB = 25.58                 # conv-centric fp32 traffic of cfg 2 per GPU


# live-event tags -> kernel names as rocprofv3 prints them (dominant template instantiation of each tag)
RP_NAMES = {
    "k_conv_zr_f16x3<3,3,3>": "k_conv_zr<2, true, 1, false, false>",      # <NS, F16, MODE (1 = fused statistics), KSPLIT, WIDE>
    "k_conv_zr_bf16x3<3,3,3>": "k_conv_zr<2, false, 0, false, false>",
    "k_conv_zr_f16<3,3,3>": "k_conv_zr<2, true, 1, false, true>",         # one-term modes: 32 channels per phase
    "k_conv_zr_bf16<3,3,3>": "k_conv_zr<2, false, 1, false, true>",
    "k_conv_zr_fp32<3,3,3>": "k_conv_zr<2, false, 1, false, false, float, true, false>",   # exact fp32 on the z-reuse structure (round 6, --precision fp32)
    "k_conv_pp_bf16x3<3,3,3,CT=2>": "k_conv_pp<3, 3, 3, 4, 8, 8, 2, 2, 2, false>",
    "k_conv_pp_bf16x3<3,3,3,CT=1>": "k_conv_pp<3, 3, 3, 4, 8, 8, 1, 1, 2, false>",
    "k_conv_pp_f16x3<3,3,3,CT=2>": "k_conv_pp<3, 3, 3, 4, 8, 8, 2, 2, 2, true>",
    "k_conv_pp_f16x3<3,3,3,CT=1>": "k_conv_pp<3, 3, 3, 4, 8, 8, 1, 1, 2, true>",
    "k_conv_fwd_bf16x6<3,3,3,NR=2>": "k_conv_fwd_bfsplit<3, 3, 3, 4, 8, 8, 2, 3, false>",
    "k_conv_fwd_bf16x6<3,3,3,NR=1>": "k_conv_fwd_bfsplit<3, 3, 3, 4, 8, 8, 1, 3, false>",
    "k_conv_fwd_bf16x3<3,3,3,NR=2>": "k_conv_fwd_bfsplit<3, 3, 3, 4, 8, 8, 2, 2, false>",
    "k_conv_fwd_bf16x3<3,3,3,NR=1>": "k_conv_fwd_bfsplit<3, 3, 3, 4, 8, 8, 1, 2, false>",
    "k_conv_fwd_f16x3<3,3,3,NR=2>": "k_conv_fwd_bfsplit<3, 3, 3, 4, 8, 8, 2, 2, true>",
    "k_conv_fwd_f16x3<3,3,3,NR=1>": "k_conv_fwd_bfsplit<3, 3, 3, 4, 8, 8, 1, 2, true>",
    "k_conv_fwd_f16<3,3,3,NR=2>": "k_conv_fwd_bfsplit<3, 3, 3, 4, 8, 8, 2, 1, true>",
    "k_conv_fwd_f16<3,3,3,NR=1>": "k_conv_fwd_bfsplit<3, 3, 3, 4, 8, 8, 1, 1, true>",
    "k_conv_wgrad_f16<3,3,3,NCO=2>": "k_conv_wgrad_tr<1>",       # round 4: one instantiation per arithmetic (no NCO variants)
    "k_conv_wgrad_f16<3,3,3,NCO=1>": "k_conv_wgrad_tr<1>",
    "k_conv_wgrad_bf16x3<3,3,3,NCO=2>": "k_conv_wgrad_tr<0>",
    "k_conv_wgrad_bf16x3<3,3,3,NCO=1>": "k_conv_wgrad_tr<0>",
    "k_conv_wgrad_f16x2<3,3,3,NCO=2>": "k_conv_wgrad_tr<3>",
    "k_conv_wgrad_f16x2<3,3,3,NCO=1>": "k_conv_wgrad_tr<3>",
}

SUSTAINED_F16_MFMA_TFLOPS = 1640.0   # measured: profiles/r03_mfma_sustained.txt (pure MFMA stream, random operands, all CUs)

PRECISION_DTYPE = {
    "fp32": "f32",
    "mixed": "f32 (forward: exact fp32 MFMA; backward convs: fp32 operands split into 2 bf16 terms, 3 bf16 MFMAs per "
             "product, fp32 accumulate -- gradient error vs float64 identical to exact fp32, tests/test_gpu_unet.py)",
    "split": "f32-class (forward: fp32 operands split into 3 bf16 terms, 6 bf16 MFMAs per product = 24-bit products; "
             "backward convs: 2 terms / 3 MFMAs; fp32 accumulate)",
    "split16": "f32-class (forward convs on normalised activations: fp32 operands split into 2 fp16 terms = 22 mantissa "
               "bits, lo plane scaled by 2^12 with its own fp32 accumulator, 3 fp16 MFMAs per product; other forward "
               "convs 3 bf16 terms / 6 MFMAs; data-gradient convs 2 bf16 terms / 3 MFMAs (16-bit products); WEIGHT-gradient "
               "convs of the pre-normalised 3x3x3 layers: xhat 2 fp16 terms x g ONE fp16 term after a power-of-two prescale from "
               "max|g| = 2 MFMAs per product, ~2e-4 unbiased relative noise on those dw tensors (profiles/"
               "r04_backward_arith_sim.txt; TEM_WGRAD_ARITH=bf16x3 restores 3 MFMAs: key wgrad_bf16x3_ms_per_step); fp32 "
               "accumulate and storage; gradient error vs float64 = that of the fp32 reference path, tests/test_gpu_unet.py)",
    "bf16x3": "bf16x3 (all MFMA convs split-bf16, fp32 accumulate)",
    "amp": "f16 (REDUCED PRECISION, not the headline configuration: activations and their gradients STORED as fp16 between "
           "the kernels of the step, conv operands fp16, one fp16 MFMA per product; fp32 accumulate, statistics, parameters, "
           "gradient arena and network output -- the counterpart of the reference's torch.autocast(float16); no loss scaling "
           "in this synthetic step; TEM_AMP_STORAGE=32 keeps fp32 tensors as in rounds 1-4)",
    "amp_bf16": "bf16 (REDUCED PRECISION, not the headline configuration: activations and their gradients STORED as bf16, "
                "conv operands bf16, one bf16 MFMA per product; fp32 accumulate, statistics, parameters, gradient arena and "
                "network output -- the counterpart of torch.autocast(bfloat16))",
}


def cpu_baseline(max_threads):
    """Oracle fwd+bwd on the host cores on the FULL cfg-2 batch (2x1x128^3): one timed step after a 32^3 thread-count
    probe (about 20-30 s of CPU work in total).  torch's CPU backend does not scale to every hardware thread of a large
    host (on the 2x64-core EPYC box 16 threads beat 64 by 3x and 256 by 600x), so the thread count is picked by the
    probe and the winner is what `cores` reports."""
    from oracle import unet_ref
    from torch_em_amd.model import UNet3d
    torch.manual_seed(0)
    sd = {k: v.detach().clone() for k, v in UNet3d(1, 2, initial_features=32, depth=4).state_dict().items()}
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(1, 1, 32, 32, 32, generator=g)
    ys = (torch.rand(1, 2, 32, 32, 32, generator=g) > 0.5).float()
    best_t, threads = None, 1
    for th in [t for t in (8, 16, 32, 64) if t <= max(max_threads, 8)]:
        torch.set_num_threads(th)
        unet_ref.unet_loss_and_grads(sd, xs, ys, [2, 2, 2, 2])
        t0 = time.perf_counter()
        unet_ref.unet_loss_and_grads(sd, xs, ys, [2, 2, 2, 2])
        dt = time.perf_counter() - t0
        if best_t is None or dt < best_t:
            best_t, threads = dt, th
    torch.set_num_threads(threads)
    x = torch.randn(2, 1, 128, 128, 128, generator=g)
    y = (torch.rand(2, 2, 128, 128, 128, generator=g) > 0.5).float()
    unet_ref.unet_loss_and_grads(sd, x, y, [2, 2, 2, 2])   # warm-up at the measured size (first-touch pages, oneDNN primitives)
    dts = []
    for _ in range(2):
        t0 = time.perf_counter()
        unet_ref.unet_loss_and_grads(sd, x, y, [2, 2, 2, 2])
        dts.append(time.perf_counter() - t0)
    dt = min(dts)
    return {"value": 2 * 128 ** 3 / dt, "unit": "voxels/s", "cores": threads, "kind": "port",
            "sample": "oracle (torch-CPU fp32 restatement) zero_grad+fwd+DiceLoss+bwd of the same UNet3d on the full cfg-2 "
                      f"batch 2x1x128^3, best of 2 steps after one warm-up step at this size: {dt:.2f} s/step (no optimizer step: AdamW is "
                      "~3 % of the reference's CPU step, BASELINE.md section 2); thread count chosen by a 32^3 probe over "
                      f"8/16/32/64 of {max_threads} hardware threads"}


def extra_measurements(step, args, engine):
    """Driver-visible numbers for what DESIGN.md claims besides the headline (rank 0, N = 1, after the timed region):
    the SAME cfg-2 step with exact-fp32 MFMA arithmetic and with the opt-in mixed-precision arithmetic (3 steps each
    after 2 warm-ups; `step` is the benchmark's own closure, the packed weights follow the precision switch), and the
    training steps of the other BASELINE configs (scripts/bench_workloads.py).  None of these is `value`."""
    import importlib.util
    out = {}
    prev = engine.PRECISION

    def timed(n=3, w=2):
        for _ in range(w):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    try:
        if args.batch == 2 and args.size == 128:
            vox = args.batch * args.size ** 3
            out["modes"] = {}
            for prec, key in (("fp32", "exact_fp32_ms_per_step"), ("amp", "amp_ms_per_step"),
                              ("amp_bf16", "amp_bf16_ms_per_step")):
                engine.set_precision(prec)
                out[key] = t = timed(n=5 if prec != "fp32" else 3)
                # the same line the headline carries, for the other arithmetics of the same step (none of them is `value`)
                m = {"ms_per_step": t, "value": vox / (t / 1e3), "unit": "voxels/s", "dtype": PRECISION_DTYPE[prec]}
                if prec == "fp32":
                    m["step_roofline"] = {"flops_frac_fp32_mfma": STEP_GFLOP / t / PEAK_FP32_MFMA_TFLOPS,
                                          "hbm_frac_8TBs": (STEP_GB / (t / 1e3)) / (PEAK_HBM_TBS * 1e3)}
                else:
                    st16 = engine.act_dtype() != torch.float32
                    gb = STEP_GB / 2 if st16 else STEP_GB
                    m["step_roofline"] = {
                        "flops_frac_16bit_mfma": STEP_GFLOP / t / PEAK_BF16_MFMA_TFLOPS,
                        "hbm_frac_8TBs": (gb / (t / 1e3)) / (PEAK_HBM_TBS * 1e3),
                        "note": f"{gb:.2f} GB algorithmic per step: the 25.58 GB of SURVEY.md 8(d) "
                                + ("halved: every activation / gradient tensor is 2 bytes per element (a LOWER bound on the bytes: "
                                   "network input, prediction, statistics, packed weights and the gradient arena stay fp32, so "
                                   "hbm_frac is slightly understated)" if st16 else
                                   "(fp32 tensors: TEM_AMP_STORAGE=32)")
                                + "; one 16-bit MFMA per product => the step is HBM-bound in this mode"}
                out["modes"]["exact_fp32" if prec == "fp32" else prec] = m
            # the default arithmetic with the three-product (bf16x3) weight gradients of rounds 1-3
            engine.set_precision(prev)
            f16x2, engine._WGRAD_F16X2 = engine._WGRAD_F16X2, False
            try:
                out["wgrad_bf16x3_ms_per_step"] = timed(n=5)
            finally:
                engine._WGRAD_F16X2 = f16x2
        engine.set_precision(prev)
        spec = importlib.util.spec_from_file_location("bench_workloads", os.path.join(ROOT, "scripts", "bench_workloads.py"))
        wl = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(wl)
        torch.cuda.empty_cache()
        for key, fn in (("cfg1_ms_per_step", wl.cfg1), ("cfg3_ms_per_step", wl.cfg3), ("cfg5_ms_per_step", wl.cfg5)):
            res = fn()
            out[key] = res["ms_per_step"]
            if "hip_graph_ms_per_step" in res:
                out[key.replace("_ms_per_step", "_hip_graph_ms_per_step")] = res["hip_graph_ms_per_step"]
            torch.cuda.empty_cache()
        spec = importlib.util.spec_from_file_location("host_overhead", os.path.join(ROOT, "scripts", "host_overhead.py"))
        ho = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ho)
        # the step as ONE replayed HIP graph (torch_em_amd/graph.py): host enqueue time and step time, at the benchmark size
        # (GPU-bound either way) and at 32^3 (host-bound when enqueued launch by launch)
        out["hip_graph"] = {"cfg2": ho.run(args.size, args.batch, 5), "patch32": ho.run(32, args.batch, 10)}
        torch.cuda.empty_cache()
        out["extras_note"] = ("exact_fp32 / amp: the cfg-2 step of this run under engine.set_precision('fp32' / 'amp'), 3 steps "
                              "after 2 warm-ups; cfg1/cfg3/cfg5: BASELINE configs 1, 3, 5 (per-GPU step incl. on-device "
                              "targets), scripts/bench_workloads.py; hip_graph: scripts/host_overhead.py (host_ms = enqueue of one step into an idle "
                              "queue, gpu_ms = back-to-back steps); all at the engine's default precision unless named")
    except Exception as e:  # the headline must survive a failing extra
        out["extras_error"] = f"{type(e).__name__}: {e}"
    finally:
        engine.set_precision(prev)
    return out


def self_launch(args_gpus):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run with one rank per GPU on
    127.0.0.1 (what the reference's train_multi_gpu does with mp.spawn, multi_gpu_training.py:172-190).  The ranks
    inherit stdout, so rank 0's JSON line is this process's JSON line."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_rank_to_gpu_numa_node(local_rank, world):
    """One rank per GPU under torch.distributed.run: the launcher exports OMP_NUM_THREADS=1 and leaves every rank free to
    run on any core.  Each rank is pinned to the CPUs of the NUMA node its GPU hangs off (PCI address from the device
    properties -> /sys/bus/pci/devices/<bdf>/numa_node), ranks that share a node split its CPUs evenly, and the OpenMP /
    torch intra-op thread count is set explicitly.  Returns what was done (the `ddp.cpu_affinity` object of the line)."""
    info = {"numa_node": None, "cpus": None, "omp_threads": None}
    try:
        allowed = os.sched_getaffinity(0)
        nodes = []
        for d in range(world):
            pr = torch.cuda.get_device_properties(d)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            try:
                nodes.append(int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read()))
            except (OSError, ValueError):
                nodes.append(-1)
        node = nodes[local_rank]
        cpus = set(allowed)
        if node >= 0:
            try:
                cpus = _cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()) & allowed or set(allowed)
            except OSError:
                pass
        peers = [r for r in range(world) if nodes[r] == node]
        mine = sorted(cpus)
        share = max(len(mine) // len(peers), 1)
        k = peers.index(local_rank)
        mine = mine[k * share:(k + 1) * share] or mine
        os.sched_setaffinity(0, mine)
        threads = max(1, min(len(mine), 8))
        os.environ["OMP_NUM_THREADS"] = str(threads)
        torch.set_num_threads(threads)
        info = {"numa_node": node, "cpus": len(mine), "omp_threads": threads}
    except Exception as e:  # an unusual /sys layout must not cost the run
        info["error"] = f"{type(e).__name__}: {e}"
    return info


def main():
    if "WORLD_SIZE" not in os.environ:
        pre = argparse.ArgumentParser(add_help=False)
        pre.add_argument("--gpus", type=int, default=1)
        n = pre.parse_known_args()[0].gpus
        if n > 1:
            sys.exit(self_launch(n))
    # The contract is ONE JSON line on stdout.  RCCL prints its version banner to fd 1 from C when the communicator is
    # created, and libraries may print warnings: route fd 1 to stderr for the whole run and keep the real stdout
    # for the final line only.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2, help="per-GPU batch (cfg 2: 2)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--norm", default="InstanceNorm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra keys (exact-fp32 / mixed-precision step times of cfg 2, cfg 1/3/5 step times)")
    ap.add_argument("--precision", default=None, choices=["fp32", "mixed", "split", "split16", "bf16x3", "amp", "amp_bf16"],
                    help="MFMA conv arithmetic (default: engine default = split16)")
    ap.add_argument("--kernel-table", default=None, help="write the per-kernel timing table to this file")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="developer A/B: a dispatch option of the library (include/tem_hip.h, tem_set_option), e.g. zr_wide=0")
    args = ap.parse_args()

    import torch.distributed as dist
    from torch_em_amd import ops
    from torch_em_amd.loss import DiceLoss
    from torch_em_amd.model import UNet3d, engine
    from torch_em_amd.multi_gpu_training import DDP
    from torch_em_amd.optim import FusedAdamW

    if args.precision:
        engine.set_precision(args.precision)
    for kv in args.option:
        from torch_em_amd import _lib
        name, _, val = kv.partition("=")
        _lib.set_option(name, int(val))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    affinity = pin_rank_to_gpu_numa_node(local_rank, world) if (world > 1 or os.environ.get("TEM_BENCH_PIN", "0") == "1") else None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    torch.manual_seed(0)
    net = UNet3d(1, 2, initial_features=32, depth=4, norm=args.norm).to(dev)
    force_ddp = os.environ.get("TEM_BENCH_FORCE_DDP", "0") == "1"  # exercise the N>1 code path on one GPU
    if force_ddp and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=0, world_size=1)
    model = DDP(net, device_ids=[local_rank]) if (world > 1 or force_ddp) else net
    if isinstance(model, DDP):
        model.sync.measure = True   # HIP events around the join of the gradient exchange (ddp.allreduce_exposed_ms)
    opt = FusedAdamW(net.parameters(), lr=1e-3)
    loss_fn = DiceLoss()
    g = torch.Generator().manual_seed(rank)
    S = args.size
    x = torch.randn(args.batch, 1, S, S, S, generator=g).to(dev)
    y = (torch.rand(args.batch, 2, S, S, S, generator=g) > 0.5).float().to(dev)

    def step():
        opt.zero_grad()
        loss = loss_fn(model(x), y)
        loss.backward()
        opt.step()
        return loss

    # Live HIP-event timing costs host time and GPU bubbles (two events per launch, ~200 per step: 3.5-6 ms/step), so
    # every conv launch is instrumented in ONE untimed step (the last warm-up step, or an extra one if --warmup 0) to
    # build the kernel table and find the dominant kernel; inside the timed region only that kernel carries events.
    noprof = os.environ.get("TEM_BENCH_NOPROF", "0") == "1"
    # A freshly booted box runs its first seconds of GPU work ~9 % slower (clock / power-state ramp: three consecutive
    # bench.py processes on one fresh box measured 31.9, 29.3, 28.5 ms/step).  Inference-only forward passes for a
    # fixed wall-clock budget bring the GPU to its steady state; they are NOT training steps and are not counted in
    # --warmup / --steps.
    prewarm_s = float(os.environ.get("TEM_BENCH_PREWARM_S", "3"))
    if prewarm_s > 0:
        tw = time.perf_counter()
        with torch.no_grad():
            while time.perf_counter() - tw < prewarm_s:
                net(x)
                torch.cuda.synchronize()
    for _ in range(max(args.warmup - 1, 0)):
        step()
    ops.PROFILER = [] if (rank == 0 and not noprof) else None
    step()
    torch.cuda.synchronize()
    table_prof, ops.PROFILER = ops.PROFILER or [], None
    dom_tag = None
    if table_prof:
        per_tag = {}
        for (tag, _shape), _fl, e0, e1 in table_prof:
            per_tag[tag] = per_tag.get(tag, 0.0) + e0.elapsed_time(e1)
        dom_tag = max(per_tag.items(), key=lambda kv: kv[1])[0]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # Events on the dominant kernel's launches in a SAMPLE of the timed steps (default 2 of them, evenly spaced): an
    # event pair costs ~35 us of dispatch overlap, and the kernel that leads the step now has 14 launches per step
    # (0.5 ms/step = 2 % of `value` if every step carried them).  TEM_BENCH_EVENT_STEPS=<n> samples n steps (>= steps: all).
    n_ev = max(1, min(args.steps, int(os.environ.get("TEM_BENCH_EVENT_STEPS", "2"))))
    ev_steps = {(i * args.steps) // n_ev for i in range(n_ev)} if (rank == 0 and dom_tag is not None) else set()
    if ev_steps:
        ops.PROFILER = []
    if isinstance(model, DDP):
        model.sync.exposed_ms()   # drop the warm-up steps' events
    # TEM_HIP_GRAPH=1: the timed steps are replays of ONE captured HIP graph (torch_em_amd/graph.py; with DDP the RCCL
    # all-reduces are nodes of it).  Per-launch events cannot be taken inside a replay, so the dominant kernel's events and
    # the exposed all-reduce time come from `ev_eager` eager steps right before the timed region.
    graphed = None
    graph_error = None
    ddp_exposed_eager = None
    # N > 1: the graph replay is the DEFAULT (eight ranks share one host: ~5 ms of Python enqueue per step and rank against
    # one hipGraphLaunch); TEM_HIP_GRAPH=0 / 1 overrides in both directions.
    use_graph = os.environ.get("TEM_HIP_GRAPH", "1" if world > 1 else "0") == "1"
    if use_graph:
        from torch_em_amd.graph import GraphedTrainStep
        for i in range(n_ev):   # on EVERY rank: the eager steps carry collectives
            ops.PROFILER_FILTER = {dom_tag} if ev_steps else set()
            step()
        torch.cuda.synchronize()
        if isinstance(model, DDP):
            ddp_exposed_eager = model.sync.exposed_ms()
        eager_dom_prof, ops.PROFILER, ops.PROFILER_FILTER = ops.PROFILER or [], None, None
        # the capture (incl. the RCCL all-reduces at N > 1) and two replays; a rank whose capture raises falls back to eager
        # launches -- TOGETHER with every other rank (the flag is MIN-reduced), so that a first contact with a multi-GPU RCCL
        # capture that does not work costs the graph, not the run
        try:
            graphed = GraphedTrainStep(model, loss_fn, opt, x, y)
            for _ in range(2):
                graphed(x, y)
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001
            graph_error, graphed = f"{type(e).__name__}: {e}", None
        if world > 1:
            ok = torch.tensor([0.0 if graph_error else 1.0], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 0.0 and graphed is not None:
                graph_error, graphed = "another rank could not capture the step", None
            dist.barrier()
        if graphed is not None:
            ev_steps = set()
        elif rank == 0:
            print(f"bench.py: HIP-graph capture failed ({graph_error}); timing eager launches", file=sys.stderr)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if ev_steps:
            ops.PROFILER_FILTER = {dom_tag} if i in ev_steps else set()
        loss = graphed(x, y)[1] if graphed is not None else step()
    # wall time of the Python loop per step, before the final synchronize: eager = host cost of enqueueing ~220 launches
    # (while the GPU queue has room); graph = one replay per step, but the loop is held back by the GPU (the 4-slot pinned
    # ring of optimizer scalars waits for the replay that used a slot): the graph's own host cost is `hip_graph.*.host_ms`
    host_loop_ms = (time.perf_counter() - t0) / args.steps * 1e3
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dom_prof, ops.PROFILER, ops.PROFILER_FILTER = ops.PROFILER or [], None, None
    if use_graph and graphed is None and not dom_prof:
        dom_prof = eager_dom_prof   # capture failed: the eager steps in front of the timed region carried the events
    if graphed is not None:
        dom_prof, ev_steps = eager_dom_prof, (set(range(n_ev)) if (rank == 0 and dom_tag is not None) else set())
    per_rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank_ms = [float(v.item()) / args.steps * 1e3 for v in gathered]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss)
    ddp_info = None
    if isinstance(model, DDP):
        ddp_info = {"bytes": model.sync.stats["bytes"], "n_collectives": model.sync.stats["n_collectives"],
                    "allreduce_exposed_ms": ddp_exposed_eager if graphed is not None else model.sync.exposed_ms(),
                    "ranks": dist.get_world_size(),
                    "backend": dist.get_backend(), "per_rank_ms_per_step": per_rank_ms, "cpu_affinity": affinity,
                    "note": "bytes / collectives of one step's gradient exchange (in-place all-reduce of arena ranges on "
                            "RCCL's stream, overlapped with backward); allreduce_exposed_ms = mean time the compute stream "
                            "waited at the join before the optimizer (HIP events), over the timed steps"}

    if rank == 0:
        voxels = args.batch * S ** 3
        ms = elapsed / args.steps * 1e3
        value = world * voxels * args.steps / elapsed
        # ---- per-kernel table from the live HIP events ----
        table, detail = {}, {}
        for (tag, shape), flops, e0, e1 in table_prof:
            dt = e0.elapsed_time(e1)
            for tab, key in ((table, tag), (detail, (tag, shape))):
                d = tab.setdefault(key, {"launches": 0, "ms": 0.0, "flops": 0.0})
                d["launches"] += 1
                d["ms"] += dt
                d["flops"] += flops
        rows = sorted(table.items(), key=lambda kv: -kv[1]["ms"])
        lines = [f"{'kernel':58s} {'launches':>8s} {'avg_ms':>9s} {'total_ms/step':>13s} {'TFLOP/s':>9s}"]
        for tag, d in rows:
            lines.append(f"{tag:58s} {d['launches']:8d} {d['ms'] / d['launches']:9.4f} "
                         f"{d['ms']:13.3f} {d['flops'] / d['ms'] / 1e9:9.2f}")
        conv_ms = sum(d["ms"] for d in table.values())
        lines.append(f"conv kernels {conv_ms:.2f} ms in the instrumented warm-up step; timed region {ms:.2f} ms/step")
        lines.append("")
        lines.append(f"{'kernel / layer (NxDxHxW cin->cout)':78s} {'launches':>8s} {'avg_ms':>9s} {'TFLOP/s':>9s}")
        for (tag, shape), d in sorted(detail.items(), key=lambda kv: -kv[1]["ms"]):
            lines.append(f"{tag + '  ' + shape:78s} {d['launches']:8d} {d['ms'] / d['launches']:9.4f} "
                         f"{d['flops'] / d['ms'] / 1e9:9.2f}")
        print("\n".join(lines), file=sys.stderr)
        if args.kernel_table:
            with open(args.kernel_table, "w") as f:
                f.write("\n".join(lines) + "\n")
        if not rows:  # TEM_BENCH_NOPROF=1 (A/B of the live-event overhead): no kernel table, no roofline
            os.write(real_stdout, (json.dumps({"ms_per_step": ms, "value": value, "note": "live profiling disabled"}) + "\n").encode())
            return
        # the dominant kernel, timed live over the timed region (HIP events on its launch stream)
        dom = {"launches": 0, "ms": 0.0, "flops": 0.0}
        for (_tag, _shape), flops, e0, e1 in dom_prof:
            dom["launches"] += 1
            dom["ms"] += e0.elapsed_time(e1)
            dom["flops"] += flops
        achieved = dom["flops"] / dom["ms"] / 1e9  # TFLOP/s (algorithmic: 2*MACs of the convolution)
        split = 6 if "bf16x6" in dom_tag else (3 if ("bf16x3" in dom_tag or "f16x3" in dom_tag) else
                                               2 if "f16x2" in dom_tag else 1 if "_f16<" in dom_tag else 0)
        traffic_file = next((f for f in (os.path.join(ROOT, "profiles", f"r0{r}_traffic_bytes_per_launch.json") for r in (6, 5, 4, 3, 2, 1))
                             if os.path.exists(f)), "")
        # split-bf16 kernels execute 3 (or 6) bf16 MFMAs per algorithmic product: effective peak = dense bf16 peak / 3 (6)
        peak = PEAK_BF16_MFMA_TFLOPS / split if split else PEAK_FP32_MFMA_TFLOPS
        standard = (args.batch == 2 and S == 128)
        # HBM bytes per launch of the dominant kernel: measured with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
        # passes (scripts/summarize_profiles.py -> profiles/r02_traffic_bytes_per_launch.json; gfx950 x2 FETCH correction)
        traffic = None
        try:
            tj = json.load(open(traffic_file))
            key = RP_NAMES.get(dom_tag.split("(")[0])
            for cand in (key, key[:-1] + ", float>" if key else None, key[:-1] + ", float, false>" if key else None,
                         key[:-1] + ", float, false, false>" if key else None):   # round 5: the kernels carry their element type; round 6: k_conv_zr its X32 / XS flags
                if cand in tj:
                    traffic = tj[cand]
                    break
        except (OSError, ValueError):
            pass
        out = {
            "metric": "voxels/sec fwd+bwd, UNet3d 1x128^3 bs=2",
            "value": value, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "host_loop_ms_per_step": host_loop_ms,
            "step_mode": ("one HIP graph replay per step (" + ("default for --gpus > 1" if "TEM_HIP_GRAPH" not in os.environ
                                                                else "TEM_HIP_GRAPH=1")
                          + "; roofline events from eager steps before the timed region)" if graphed is not None else "eager launches"),
            "dtype": PRECISION_DTYPE[engine.PRECISION], "data": "synthetic",
            "config": {"workload": f"UNet3d(1->2, initial_features=32, depth=4, norm={args.norm}) + DiceLoss, "
                                   f"zero_grad+fwd+loss+bwd+AdamW, per-GPU batch {args.batch}x1x{S}^3"
                                   + ("" if standard else " (NON-STANDARD SIZE)"),
                       "parallelism": f"dp{world}", "global_batch": world * args.batch, "final_loss": final_loss},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic,
                         "traffic_source": (os.path.relpath(traffic_file, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                            "an earlier run of this command, not measured in this run)") if traffic is not None else None,
                         "kernel": dom_tag,
                         "peak_note": (f"dense bf16 MFMA peak 2500 TFLOP/s / {split} MFMAs per product (split-bf16, fp32 "
                                       "accumulate); executed-MFMA fraction of 2500 = frac" if split else
                                       "exact-fp32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
                         # what this chip SUSTAINS: a register-only stream of v_mfma_f32_32x32x16_f16 on all 256 CUs with
                         # random operands runs at 1.65 GHz (power-limited; 2.2 GHz / 2150 TF on all-zero operands):
                         # scripts/proto/memtime_cal.hip, profiles/r03_mfma_sustained.txt
                         "sustained_peak": (SUSTAINED_F16_MFMA_TFLOPS / split) if split else None,
                         "frac_of_sustained": (achieved / (SUSTAINED_F16_MFMA_TFLOPS / split)) if split else None,
                         "launches_per_step": dom["launches"] // max(len(ev_steps), 1),
                         "event_steps": sorted(ev_steps),
                         "avg_launch_ms": dom["ms"] / dom["launches"],
                         "flops_per_launch_avg": dom["flops"] / dom["launches"]},
        }
        if graph_error is not None:
            out["step_mode"] += f" (the HIP-graph replay that is the default for --gpus > 1 could not be captured: {graph_error})"
        if ddp_info is not None:
            out["ddp"] = ddp_info
        if standard:
            out["step_roofline"] = {
                "flops_frac_fp32_mfma": STEP_GFLOP / ms / PEAK_FP32_MFMA_TFLOPS,
                "hbm_frac_8TBs": (STEP_GB / (ms / 1e3)) / (PEAK_HBM_TBS * 1e3),
                "note": "5700.8 GFLOP and 25.58 GB algorithmic per step (SURVEY.md 8d); fp32 arithmetic => MFMA-bound"}
        if world == 1 and not args.no_extras:
            out.update(extra_measurements(step, args, engine))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(os.cpu_count() or 1)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
