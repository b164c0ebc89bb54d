"""Benchmark of the DDPM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W            (one rank per GPU, RCCL)

A "step" is one full training step of `DiffusionModel.train_step` (label drop, q_sample, U-Net forward, loss,
backward, bucketed gradient all-reduce, fused Adam) on one batch of synthetic MNIST-shaped images that is already
resident in HBM.  Workload = BASELINE.json configs[1]: 1x28x28, C=128, per-GPU batch 1024, bf16 compute with fp32
master weights (weak scaling: the per-GPU batch is fixed).  Rank 0 prints ONE JSON line:
  value            whole-job train images/s (all ranks), max-over-ranks wall time around exactly K steps
  sampler          reverse-diffusion steps/s (DDIM, guidance off, batch 1024/GPU), timed separately
  roofline         dominant kernel = the MFMA implicit-GEMM convolution (forward + data-gradient launches):
                   algorithmic FLOPs of every launch in the timed region / their HIP-event durations, vs the dense
                   bf16 MFMA peak
  cpu_baseline     the oracle (CPU restatement) timed on the host cores on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 (guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=28)
    ap.add_argument("--in_channels", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--attention", type=int, default=0, help="1: self-attention block at the S/4 level (BASELINE config 5 shape)")
    ap.add_argument("--sampler_steps", type=int, default=20)
    ap.add_argument("--cpu_seconds", type=float, default=15.0)
    ap.add_argument("--no_cpu", action="store_true")
    ap.add_argument("--no_profile", action="store_true", help="skip the per-launch HIP events")
    return ap.parse_args()


def synthetic_batch(B, C, S, device, seed):
    g = torch.Generator().manual_seed(seed)
    raw = torch.rand((B, C, S, S), generator=g) * 2 - 1
    ink = torch.rand((B, C, S, S), generator=g) < 0.15
    x = torch.where(ink, raw, -torch.ones_like(raw))       # MNIST-like: 85 % of pixels exactly -1
    y = torch.randint(0, 10, (B,), generator=g)
    return x.to(device), y.to(device)


def cpu_baseline(seconds, S, in_channels):
    """Oracle train step (forward + autograd backward + Adam restatement) on the host cores, cfg1 shape B=32."""
    from oracle import diffusion_ref as D
    from oracle import unet_ref as U
    # the threads this process may actually run on (a GPU box gives one GPU's share of the host, 16 cores)
    ncores = min(len(os.sched_getaffinity(0)), int(os.environ.get("GMK_CPU_THREADS", "16")))
    torch.set_num_threads(max(1, ncores))
    B = 32
    params = {k: v.requires_grad_(True) for k, v in U.reference_init_params(128, in_channels).items()}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v2 = {k: torch.zeros_like(v) for k, v in params.items()}
    x, y = synthetic_batch(B, in_channels, S, "cpu", 7)
    g = torch.Generator().manual_seed(3)
    n, t_total, step = 0, 0.0, 0
    while True:
        u = torch.rand((B,), generator=g); eps = torch.randn(x.shape, generator=g)
        t0 = time.perf_counter()
        loss = D.training_losses(params, x, y, u, eps)["loss"].mean()
        grads = torch.autograd.grad(loss, [p for k, p in params.items() if not k.startswith("cond_w_embed")])
        step += 1
        with torch.no_grad():
            for (k, p), gr in zip([(k, p) for k, p in params.items() if not k.startswith("cond_w_embed")], grads):
                pn, m[k], v2[k] = D.adam_step(p, gr, m[k], v2[k], step)
                p.copy_(pn)
        dt = time.perf_counter() - t0
        if step > 1:                      # first iteration is the warm-up
            n += 1; t_total += dt
        if t_total >= seconds or step >= 200 or (step == 1 and dt > seconds):
            if n == 0:
                n, t_total = 1, dt
            break
    # reverse-diffusion beside it: DDIM, guidance off, B=32, a few steps of the T=1000 schedule
    with torch.no_grad():
        pd = {k: v.detach() for k, v in params.items()}
        init = torch.randn(x.shape, generator=g)
        D.sample(pd, init, y, 2, sampler="ddim", record=False)             # warm-up
        ns = 8
        t0 = time.perf_counter()
        D.sample(pd, init, y, ns, sampler="ddim", record=False)
        ts = time.perf_counter() - t0
    return {"value": round(B * n / t_total, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle (torch-CPU restatement) train step, B=32, {in_channels}x{S}x{S}, C=128, fp32, "
                      f"{n} steps in {t_total:.1f} s after 1 warm-up",
            "sampler_steps_per_sec": round(ns / ts, 2), "sampler_sample": f"oracle DDIM, guidance off, B=32, {ns} steps in {ts:.2f} s"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GMK_DIST_BACKEND", "nccl") != "nccl":
        local = local % max(1, torch.cuda.device_count())              # rehearsal: several ranks may share a GPU
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        backend = os.environ.get("GMK_DIST_BACKEND", "nccl")       # "gloo": rehearsal of the N > 1 path on one GPU (tests)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from generative_models_amd import common, ops, parallel
    Model = common.discover_models()["diffusion"]
    G = common.AttrDict(dict(Model.DG))
    G.update(lr=3e-4, pad32=0, device=str(dev), timesteps=1000, bs=a.batch, compute_dtype=a.dtype,
             in_channels=a.in_channels, seed=0, attention=a.attention)
    model = Model(G).to(dev)
    model.size = a.size
    if world > 1:
        parallel.GradSync(model.net).broadcast_params(0)
    x, y = synthetic_batch(a.batch, a.in_channels, a.size, dev, 1000 + rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    model.train()
    for _ in range(a.warmup):
        model.train_step(x, y.clone())
    barrier()
    # per-launch HIP events are taken on every 5th step of the timed region only: each event pair costs the command
    # processor its launch overlap (~3 % of the step when every launch of every step carries one)
    prof = None if a.no_profile else []
    nprof = 0
    t0 = time.perf_counter()
    side = ops.WGRAD_STREAM
    for i in range(a.steps):
        ops.PROFILE = prof if (prof is not None and i % 5 == 0) else None
        nprof += ops.PROFILE is not None
        # a launch's HIP-event duration is only that kernel's own time if nothing else shares the chip: the profiled steps keep
        # the weight gradients on the main stream (they run ~4 % slower than the other steps, which is inside `value`)
        ops.WGRAD_STREAM = side and ops.PROFILE is None
        model.train_step(x, y.clone())
    ops.WGRAD_STREAM = side
    barrier()
    elapsed = time.perf_counter() - t0
    ops.PROFILE = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    images_per_s = world * a.batch * a.steps / elapsed

    # ---- roofline of the dominant kernel from the per-launch HIP events of the timed region (rank 0)
    roofline = None
    if prof:
        torch.cuda.synchronize()
        peak = MFMA_BF16_PEAK_TFLOPS if a.dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
        by = {}
        for name, s, e, f in prof:
            d = by.setdefault(name, [0.0, 0.0, 0])
            d[0] += s.elapsed_time(e); d[1] += f; d[2] += 1
        dom = max(by, key=lambda k: by[k][0])          # dominant kernel = largest total HIP-event time
        desc = {"conv3x3_halo_ws_kernel": "3x3 stride-1 bf16 convolution, LDS-resident halo, wave-specialised (forward + data gradients)",
                "conv3x3_halo_kernel": "3x3 stride-1 bf16 convolution, LDS-resident halo (forward + data gradients)",
                "conv3x3_halo_ws_kernel[zero-stuffed transposed conv]": "data gradient of the stride-2 convs = the same kernel on the zero-stuffed gradient (2 launches/step; algorithmic FLOPs are 1/4 of its MFMA work)",
                "conv_igemm_dma_kernel": "im2col LDS-DMA convolution (1x1, strided, upsampled, fp32)",
                "conv_igemm_kernel": "im2col register-staged convolution (small problems)",
                "conv_wgrad_slots_kernel": "3x3 weight gradient over padded slots (+ slab reduce)",
                "conv_wgrad_kernel": "im2col split-K weight gradient (+ slab reduce)"}
        ms, fl, n = by[dom]
        ach = fl / (ms * 1e-3) / 1e12
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")     # PMC FETCH/WRITE_SIZE of the same command (rocprofv3)
        if os.path.exists(tfile):
            traffic = json.load(open(tfile)).get(dom, {}).get("hbm_bytes_per_launch")
        roofline = {"kernel": dom, "what": desc.get(dom, ""), "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
                    "launches_per_step": n // nprof, "avg_launch_us": round(ms * 1e3 / n, 2),
                    "share_of_step_time": round(ms * 1e-3 / (elapsed * nprof / a.steps), 3),
                    "profiled_steps": nprof,
                    "other_kernels": {k: {"achieved": round(v[1] / (v[0] * 1e-3) / 1e12, 2), "avg_launch_us": round(v[0] * 1e3 / v[2], 2),
                                          "launches_per_step": v[2] // nprof,
                                          "share_of_step_time": round(v[0] * 1e-3 / (elapsed * nprof / a.steps), 3)}
                                      for k, v in by.items() if k != dom}}

    # ---- reverse-diffusion steps/s: DDIM, guidance off (the `evaluate` path), trajectories not recorded
    from functools import partial
    model.eval()
    model.diffusion.num_steps = a.sampler_steps
    init = model._aux_rng.normal((a.batch, a.in_channels, a.size, a.size), dev)
    model.diffusion.sample(net=partial(model.net, guide=y), init_x=init, record=False)     # warm-up
    ts = float("inf")
    for _ in range(2):                               # two timed passes, the faster one is reported (the chip re-clocks after the train loop)
        barrier()
        t0 = time.perf_counter()
        model.diffusion.sample(net=partial(model.net, guide=y), init_x=init, record=False)
        barrier()
        ts = min(ts, time.perf_counter() - t0)
    if world > 1:
        t = torch.tensor([ts], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ts = float(t)
    sampler = {"steps_per_sec": round(a.sampler_steps / ts, 2),
               "image_steps_per_sec": round(world * a.batch * a.sampler_steps / ts, 1),
               "batch_per_gpu": a.batch, "mode": "ddim, guidance off, 1 U-Net forward per step",
               "timed_steps": a.sampler_steps, "timing": "faster of two passes after one warm-up pass"}

    def time_sampler(kind, cond_w):              # the other two modes of SURVEY M1(ii), same batch and step count
        model.diffusion.sampler = kind
        run = lambda: model.diffusion.sample(net=partial(model.net, guide=y), init_x=init, cond_w=cond_w, record=False)
        run()
        t = float("inf")
        for _ in range(2):
            barrier()
            t0 = time.perf_counter()
            run()
            barrier()
            t = min(t, time.perf_counter() - t0)
        if world > 1:
            tt = torch.tensor([t], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt)
        return round(a.sampler_steps / t, 2)

    if a.sampler_steps > 0:
        sampler["other_modes_steps_per_sec"] = {
            "ddim, guidance on (the `sample` path: conditional + unconditional forward per step)": time_sampler("ddim", 0.5),
            "noisy (ancestral), guidance off": time_sampler("noisy", None)}
        model.diffusion.sampler = "ddim"

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:
        cpu = cpu_baseline(a.cpu_seconds, a.size, a.in_channels)

    if rank == 0:
        fwd_gflop = {(1, 28): 4.3913, (3, 32): 5.7447, (3, 64): 22.9754}.get((a.in_channels, a.size))
        line = {
            "metric": "ddpm_train_images_per_sec", "value": round(images_per_s, 1), "unit": "images/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"DDPM train step, {a.in_channels}x{a.size}x{a.size} images, SimpleUnet C=128"
                                   f"{' + self-attention' if a.attention else ''}, per-GPU batch {a.batch}, T=1000"
                                   f"{' (BASELINE.json configs[1])' if (a.in_channels, a.size, a.batch, a.attention) == (1, 28, 1024, 0) else ''}",
                       "global_batch": world * a.batch, "parallelism": f"dp{world}",
                       "optimizer": "fused Adam lr=3e-4", "mean_type": "v"},
            "sampler": sampler,
        }
        if fwd_gflop:
            line["model_tflops"] = round(3 * fwd_gflop * images_per_s / 1e3, 2)   # 3x forward FLOPs per train image
        if roofline:
            line["roofline"] = roofline
        if cpu:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
