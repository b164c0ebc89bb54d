"""Per-step wall times of one bench config (synchronised every step) + allocator statistics: tells a kernel-side slowdown from
allocator churn (retries / growth) when a long run is slower than a short one.
    python tools/step_times.py cfg3 30"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from generative_models_amd import common, parallel

key, n = sys.argv[1], int(sys.argv[2])
cin, S, B, attention, _ = bench.CONFIGS[key]
dev = torch.device("cuda", 0)
Model = common.discover_models()["diffusion"]
G = common.AttrDict(dict(Model.DG))
G.update(lr=3e-4, pad32=0, device=str(dev), timesteps=1000, bs=B, compute_dtype="bf16", in_channels=cin, seed=0, attention=attention)
model = Model(G).to(dev); model.size = S; model.train()
batches = [bench.synthetic_batch(B, cin, S, dev, 1000 + k) for k in range(4)]
for i in range(n):
    x, y = batches[i % 4]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    model.train_step(x, y.clone())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    print(f"step {i:3d} {dt * 1e3:8.2f} ms  reserved {st['reserved_bytes.all.current'] / 2**30:7.1f} GiB  allocated-peak {st['allocated_bytes.all.peak'] / 2**30:7.1f} GiB  "
          f"retries {st['num_alloc_retries']}  mallocs {st['num_device_alloc']}  frees {st['num_device_free']}", flush=True)
if len(sys.argv) > 3:          # then: per-step times of N sampler iterations (DDIM, guidance off) right behind the training phase
    from functools import partial
    from generative_models_amd import ops
    from generative_models_amd.diffusion.gaussian_diffusion import logsnr_schedule_cosine_host, sampler_times
    model.eval()
    x, y = batches[0]
    z = model._aux_rng.normal((B, cin, S, S), dev)
    T = int(sys.argv[3])
    for i in range(T)[::-1]:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        u_t, u_s = sampler_times(i, T)
        lt, ls = logsnr_schedule_cosine_host(u_t), logsnr_schedule_cosine_host(u_s)
        v = model.net.forward_hip(z, torch.full((B,), float(lt), device=dev), y, None)
        z, _, _ = ops.sampler_step(v, z, lt, ls, i == 0)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = torch.cuda.memory_stats()
        print(f"sampler i={i:3d} {dt * 1e3:8.2f} ms  reserved {st['reserved_bytes.all.current'] / 2**30:7.1f} GiB  active {st['active_bytes.all.current'] / 2**30:6.1f} GiB  "
              f"retries {st['num_alloc_retries']}  mallocs {st['num_device_alloc']}  frees {st['num_device_free']}", flush=True)
