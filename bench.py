"""Benchmark of the DDPM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W           (N > 1 without WORLD_SIZE: starts its own N ranks as a child `torch.distributed.run`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W            (one rank per GPU, RCCL)

A "step" is one full training step of `DiffusionModel.train_step` (label drop, q_sample, U-Net forward, loss, backward,
bucketed gradient all-reduce, fused Adam) on a batch of synthetic images already resident in HBM (4 distinct resident batches
are cycled).  Workloads are BASELINE.json's configs (C=128, bf16 compute, fp32 master weights, weak scaling: the per-GPU batch
is fixed):
    N = 1   headline configs[2] (3x32x32, B=2048: the largest config that is quoted on one GPU), `other_configs` =
            configs[1] (1x28x28, B=1024) and the per-GPU shards of configs[3] (3x64x64, B=1024) and configs[4] (+ fp8 self-attention, B=512)
    N > 1   headline = the SAME per-GPU workload (3x32x32, 2048 images per GPU: global batch N x 2048), so that value(N) / (N x value(1))
            of the top-level lines is the weak-scaling efficiency of one workload; `other_configs` = the per-GPU shards of the two
            configs BASELINE quotes on 8 GPUs: configs[3] (3x64x64, 1024 per GPU: its single-GPU reference is other_configs.cfg3 of the
            N = 1 line) and configs[4] (3x64x64 + fp8 self-attention, 512 per GPU)
Rank 0 writes the FULL record (every kernel, every configuration, provenance strings) to `bench_detail.json` (under gpurun_out/ when that
directory exists, else the repository root; `GMK_BENCH_DETAIL` overrides the path) and prints ONE compact JSON line of at most 4 KB as
the LAST line of stdout (`compact_line`; the driver keeps only a bounded tail of stdout - round 3's 20 KB line was cut):
  value            whole-job train images/s over exactly K timed steps after W warm-up steps, max-over-ranks wall time
  steady_state     the same loop again for >= 50 more steps (SURVEY §8d M1 asks for >= 50 steps after >= 10 warm-up)
  sampler          reverse-diffusion steps/s: DDIM guidance off over the FULL T=1000 loop, guided DDIM and the ancestral
                   sampler over shorter loops (their step count is stated), trajectories not recorded
  roofline         dominant kernel (largest total HIP-event time): algorithmic FLOPs of its launches in the timed region /
                   their HIP-event durations, against the dense bf16 MFMA peak; events are recorded on the stream each kernel
                   runs on, on every 5th timed step
  cpu_baseline     the oracle (CPU restatement) timed on the host cores on a bounded sample (rank 0, N = 1 only)
  exchange         N > 1: what the gradient exchange saw (world, backend, RCCL version, bucket bytes, CUs left to RCCL)
"""
import argparse
import json
import os
import sys
import time
from functools import partial

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0     # MI355X dense bf16 / fp16 (guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3
VECTOR_F32_PEAK_TFLOPS = 157.3     # the vector ALUs' fp32 peak: the ceiling the north_star's wording ("MFMA only for attention") would put on the convolutions
HBM_PEAK_GBS = 8000.0              # HBM3E spec peak (a device copy measures 5.3 - 5.5 TB/s on this part: tools/hbm_probe.py)
FWD_GFLOP = {(1, 28): 4.3913, (3, 32): 5.7447, (3, 64): 22.9754}      # U-Net forward per image

CONFIGS = {      # name -> (in_channels, size, per-GPU batch, attention, BASELINE.json entry)
    "cfg1": (1, 28, 1024, 0, "configs[1]: DDPM MNIST 28x28x1, bs=1024, T=1000, bf16, 1xMI355X"),
    "cfg2": (3, 32, 2048, 0, "configs[2]: DDPM 32x32x3, bs=2048, T=1000, 1xMI355X"),
    "cfg3": (3, 64, 1024, 0, "configs[3]: DDPM 64x64x3, bs=8192 over 8 GPUs = 1024 per GPU"),
    "cfg4": (3, 64, 512, 2, "configs[4]: DDPM 64x64x3 + self-attention (fp8 MFMA), bs=4096 over 8 GPUs = 512 per GPU"),
}
KERNEL_DESC = {
    "conv3x3_halo_ws_kernel": "3x3 stride-1 bf16 convolution, LDS-resident halo, wave-specialised (forward + data gradients)",
    "conv3x3_halo_kernel": "3x3 stride-1 bf16 convolution, LDS-resident halo (forward + data gradients)",
    "conv3x3_halo_ws_kernel[zero-stuffed transposed conv]": "data gradient of the stride-2 convs = the same kernel on the zero-stuffed gradient (algorithmic FLOPs are 1/4 of its MFMA work)",
    "conv_igemm_dma_kernel": "im2col LDS-DMA convolution (1x1, strided, upsampled, fp32)",
    "conv_igemm_dma_kernel[stride-2 dgrad phases]": "data gradient of the stride-2 convs where the halo kernel declines: four output-parity phase launches of the LDS-DMA kernel (1 / 2 / 2 / 4 taps); one timed unit = the four launches",
    "conv_igemm_kernel": "im2col register-staged convolution (small problems)",
    "conv1x1_wgrad_stream_kernel": "weight gradient of the 1x1 skip convolution over the concatenated input: one workgroup per pixel range owns both input blocks (+ slab reduce)",
    "conv1x1_pair_stream_kernel": "data gradient of the 1x1 skip convolution, both 128-channel halves from one read: weights resident in LDS, 128-pixel tiles streamed",
    "conv_wgrad_slots_kernel": "3x3 weight gradient over padded slots, 8 compute waves (+ slab reduce)",
    "conv_wgrad_slots_ws_kernel": "3x3 weight gradient over padded slots, wave-specialised (+ slab reduce)",
    "conv_wgrad_slots_ws_kernel[stride-2 planes]": "weight gradient of the stride-2 3x3 convolution: the slot kernel on the four parity planes of the input (+ slab reduce)",
    "conv_wgrad_kernel": "im2col split-K weight gradient (+ slab reduce)",
    "conv_subpixel_ws_kernel[upsample]": "`Upsample` (nearest x2 + 3x3) as four output parities of 2x2 taps on pre-summed weights, low-resolution halo resident for all four (16 of 36 tap-products; FLOPs quoted are the reference op's)",
    "conv_subpixel_ws_kernel[transposed]": "data gradient of the stride-2 convs: output parities meet 1 / 2 / 2 / 4 taps over the gradient's own grid (9 of 36 tap-products of the zero-stuffed form)",
    "conv_subpixel_ws_kernel[upsample dgrad]": "data gradient of `Upsample`: the transposed sub-pixel form over the four parity views of the gradient, one launch instead of dgrad3x3 + sumpool2x2",
    "conv_wgrad_subpixel_ws_kernel": "weight gradient of `Upsample`: 16 tap gradients over the low-resolution slots, two dY parity streams (+ two-stage reduce onto the 9 taps)",
    "gn_silu_fwd_reg_kernel": "GroupNorm+SiLU forward, register-resident (one read + one write of the tensor)",
    "gn_silu_fwd_kernel": "GroupNorm+SiLU forward, two-sweep streaming (8x8 / 7x7 levels, 64-pixel rows)",
    "gn_silu_bwd_hybrid_kernel": "GroupNorm+SiLU backward, dy in registers + x parked in LDS (x, dy, addends read once, dx written once)",
    "gn_silu_bwd_kernel": "GroupNorm+SiLU backward, two-sweep streaming",
    "expand3x3_mfma_kernel": "stem forward / head data gradient (<= 4 image channels <-> 128), fp32 MFMA",
    "expand3x3_tile16_kernel": "stem forward / head data gradient, 16-bit results: 128-pixel tiles, image window in LDS, bf16 hi + lo MFMA",
    "wgrad3x3_mfma_kernel": "stem / head weight gradients, fp32 MFMA", "wgrad3x3_tile16_kernel": "stem / head weight gradients of 16-bit tensors: 128-pixel tiles through LDS, bf16 hi + lo MFMA", "head_fwd_mfma_kernel": "head forward (128 -> <= 3 image channels)",
    "sumpool2x2_kernel": "backward of the nearest x2 upsample", "chansum_kernel": "per-sample channel sums (bias / embedding gradients)"}


def fp32_mode():
    """(label, MFMA peak in TFLOP/s) of the fp32 mode's convolutions: exact v_mfma_f32_32x32x2_f32 chains by default; with GMK_FP32_SPLIT=1 /
    gmk_set_fp32_exact(0) (round 6) each fp32 product is hi hi + hi lo + lo hi of bf16 halves on the bf16 matrix cores (three MFMAs per product: a
    third of the dense bf16 peak)."""
    from generative_models_amd._lib import lib
    if lib.gmk_fp32_split():
        return "fp32 storage, 3x bf16 MFMA (hi / lo split, fp32 accumulation)", round(MFMA_BF16_PEAK_TFLOPS / 3, 1)
    return "fp32 storage, exact-fp32 MFMA", MFMA_F32_PEAK_TFLOPS


def kernel_hash():
    """sha1 over the kernel sources: stamps the PMC traffic file so that a bench line never quotes counters of other kernels."""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "generative_models_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "generative_models_amd", "csrc", "*.h")) + [os.path.join(ROOT, "include", "gmk.h")]):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="auto", help="auto | cfg1 | cfg2 | cfg3 | cfg4 | custom (then --batch/--size/--in_channels/--attention)")
    ap.add_argument("--batch", type=int, default=1024, help="per-GPU batch (custom config)")
    ap.add_argument("--size", type=int, default=28)
    ap.add_argument("--in_channels", type=int, default=1)
    ap.add_argument("--attention", type=int, default=0)
    ap.add_argument("--hidden", type=int, default=128, help="hidden_size of a custom config (the reference's gms/main.py default is 256)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--sampler_steps", type=int, default=1000, help="T of the headline's guidance-off DDIM loop (0: skip the samplers)")
    ap.add_argument("--sampler_steps_other", type=int, default=100, help="loop length of the other sampler modes / configs")
    ap.add_argument("--others", type=int, default=1, help="0: headline only")
    ap.add_argument("--cpu_seconds", type=float, default=15.0)
    ap.add_argument("--no_cpu", action="store_true")
    ap.add_argument("--no_profile", action="store_true", help="skip the per-launch HIP events")
    return ap.parse_args()


def synthetic_batch(B, C, S, device, seed):
    """MNIST-like synthetic images: 85 % of the pixels exactly -1, the rest U(-1, 1); labels U{0..9}."""
    g = torch.Generator().manual_seed(seed)
    raw = torch.rand((B, C, S, S), generator=g) * 2 - 1
    ink = torch.rand((B, C, S, S), generator=g) < 0.15
    x = torch.where(ink, raw, -torch.ones_like(raw))
    y = torch.randint(0, 10, (B,), generator=g)
    return x.to(device), y.to(device)


def cpu_baseline(seconds, S, in_channels):
    """Oracle train step (forward + autograd backward + Adam restatement) on the host cores, B=32, at the image shape handed in:
    bench.py runs it at the headline's shape (the "same workload" the task's measurement contract asks for) and at configs[0]'s
    1x28x28 (the shape SURVEY §8d M5 / BASELINE.md §3 name)."""
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


LINE_LIMIT = 4096        # bytes of the last stdout line (the driver keeps a bounded tail of stdout and parses its last line)


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}


def _compact_roofline(r):
    """The dominant kernel against the MFMA roofline + the top HBM-bound kernel against the HBM roofline; no prose."""
    if not r:
        return None
    out = _pick(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_launch_us",
                    "share_of_step_time", "achieved_vs_vector_peak"))
    out.setdefault("traffic", None)
    if r.get("traffic_withheld"):
        out["traffic_withheld"] = True
    if r.get("pmc"):
        out["pmc"] = {k: _pick(v, ("mfma_busy_frac", "sclk_ghz")) for k, v in r["pmc"].items()}
    hbm = r.get("hbm") or {}
    top = hbm.get("top_hbm_bound_kernel")
    if top and top in hbm.get("kernels", {}):
        out["hbm"] = dict(kernel=top, **_pick(hbm["kernels"][top], ("achieved", "peak", "unit", "frac", "traffic", "avg_launch_us",
                                                                      "launches_per_step", "share_of_step_time")))
    return out


def _compact_sampler(s):
    if not s:
        return None
    out = _pick(s, ("steps_per_sec", "image_steps_per_sec", "timed_steps", "batch_per_gpu"))
    out["mode"] = "ddim, guidance off"
    for name, v in (s.get("other_modes") or {}).items():
        out["guided" if "guidance on" in name else "ancestral"] = v.get("steps_per_sec")
    return out


def _compact_exchange(e):
    return _pick(e, ("world", "backend", "forced", "rccl_version", "bucket_bytes", "persistent_kernel_cus", "rccl_max_channels", "exposed_ms", "ab")) if e else None


def compact_line(full, detail_path=None):
    """The ONE machine-readable line: every number the measurement contract names, no prose, <= LINE_LIMIT bytes for every --gpus N.
    `full` is the complete record (what round 3 printed); it goes to `bench_detail.json` instead."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"))
    line["vs_baseline"] = full.get("vs_baseline")
    line.update(_pick(full, ("dtype", "data")))
    line["config"] = _pick(full.get("config"), ("workload", "key", "global_batch", "parallelism"))
    line.update(_pick(full, ("model_tflops",)))
    if full.get("steady_state"):
        line["steady_state"] = _pick(full["steady_state"], ("value", "steps", "ms_per_step"))
    line["roofline"] = _compact_roofline(full.get("roofline"))
    cpu = full.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = _pick(cpu, ("value", "unit", "cores", "kind", "sample", "sampler_steps_per_sec"))
        if cpu.get("cfg0_shape"):
            line["cpu_baseline"]["cfg0_shape"] = _pick(cpu["cfg0_shape"], ("value", "sampler_steps_per_sec", "sample"))
    if full.get("sampler"):
        line["sampler"] = _compact_sampler(full["sampler"])
    others = {}
    for k, o in (full.get("other_configs") or {}).items():
        c = _pick(o, ("value", "ms_per_step", "global_batch", "model_tflops", "dtype"))
        r = o.get("roofline") or {}
        if r:
            c["frac"] = r.get("frac")
            top = (r.get("hbm") or {}).get("top_hbm_bound_kernel")
            if top:
                c["hbm_frac"] = r["hbm"]["kernels"][top].get("frac")
        if o.get("sampler"):
            c["sampler_steps_per_sec"] = o["sampler"].get("steps_per_sec")
        if o.get("exchange"):
            c["exposed_ms"] = o["exchange"].get("exposed_ms")
        others[k] = c
    if others:
        line["other_configs"] = others
    if full.get("exchange"):
        line["exchange"] = _compact_exchange(full["exchange"])
    if detail_path:
        line["detail"] = detail_path
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:                       # never print an unparseable line: shed the optional blocks, largest first
        for k in ("other_configs", "steady_state", "sampler"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= LINE_LIMIT:
                break
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def detail_path():
    p = os.environ.get("GMK_BENCH_DETAIL")
    if p:
        return p
    d = os.path.join(ROOT, "gpurun_out")
    return os.path.join(d if os.path.isdir(d) else ROOT, "bench_detail.json")


class Bench:
    def __init__(self, a):
        self.a = a
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = os.environ.get("GMK_DIST_BACKEND", "nccl")       # "gloo": rehearsal of the N > 1 path on one GPU (tests)
        if self.backend != "nccl":
            local = local % max(1, torch.cuda.device_count())           # rehearsal: several ranks may share a GPU
        assert torch.cuda.is_available(), "bench.py needs an MI355X (the HIP path has no CPU fallback)"
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        # GMK_FORCE_EXCHANGE=1 with one rank (round 6): a ONE-rank RCCL communicator, so that the bucketed all-reduces run from the exchange
        # stream beside the backward pass on the one GPU a box has (parallel.exchanging); the record's `exchange` block says "forced"
        self.forced = self.world == 1 and os.environ.get("GMK_FORCE_EXCHANGE", "0") == "1"
        if self.forced and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local))
        if self.world > 1 or self.forced:
            from generative_models_amd.parallel import configure_rccl_env
            configure_rccl_env()                     # RCCL's channel count = the CUs the persistent kernels leave free
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
        assert self.world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={self.world}: launch with torch.distributed.run"

    def barrier(self):
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.world > 1:
            t = torch.tensor([seconds], device=self.dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t)
        return seconds

    def timed_steps(self, model, batches, steps, prof):
        """Exactly `steps` train steps bracketed by barrier + synchronize; -> (max-over-ranks seconds, profiled steps)."""
        from generative_models_amd import ops
        side = ops.WGRAD_STREAM
        nprof = 0
        self.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            ops.PROFILE = prof if (prof is not None and i % 5 == 0) else None
            nprof += ops.PROFILE is not None
            # a launch's HIP-event duration is that kernel's own time only if nothing else shares the chip: the profiled steps keep
            # the weight gradients on the main stream (they run ~4 % slower than the other steps, which is inside `value`)
            ops.WGRAD_STREAM = side and ops.PROFILE is None
            x, y = batches[i % len(batches)]
            model.train_step(x, y.clone())
        ops.WGRAD_STREAM = side
        self.barrier()
        elapsed = self.max_over_ranks(time.perf_counter() - t0)
        ops.PROFILE = None
        return elapsed, nprof

    def roofline(self, prof, nprof, elapsed, steps, key, dtype=None):
        """Dominant kernel against the MFMA roofline (FLOP/s), every HBM-bound kernel and every kernel above 2 % of the step against
        the HBM roofline (algorithmic bytes / HIP-event time / 8 TB/s), all from this run's events; PMC traffic from the committed
        counter passes only if they were taken on these very kernel sources."""
        if not prof:
            return None
        torch.cuda.synchronize()
        peak = MFMA_BF16_PEAK_TFLOPS if (dtype or self.a.dtype) == "bf16" else fp32_mode()[1]
        by = {}
        inst = {}
        for name, s, e, f, nb, fm, ikey in prof:
            dt = s.elapsed_time(e)
            for d in (by.setdefault(name, [0.0, 0.0, 0, 0.0, 0.0]),) + ((inst.setdefault(ikey, [0.0, 0.0, 0, 0.0, 0.0]),) if ikey else ()):
                d[0] += dt; d[1] += f; d[2] += 1; d[3] += nb; d[4] += fm
        mfma = {k: v for k, v in by.items() if k.startswith("conv")}
        dom = max(mfma, key=lambda k: mfma[k][0])          # dominant kernel = largest total HIP-event time
        ms, fl, n = by[dom][:3]
        ach = fl / (ms * 1e-3) / 1e12
        traffic, source, pmc, stale = None, None, None, None
        import glob
        tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
        tfile = tfiles[-1] if tfiles else ""          # HBM bytes per launch from separate rocprofv3 --pmc passes, the newest round's file
        kern = {}
        if os.path.exists(tfile):
            rec = json.load(open(tfile)).get(key, {})
            if rec.get("kernel_hash") == kernel_hash():
                kern = rec.get("kernels", {})
                traffic = kern.get(dom, {}).get("hbm_bytes_per_launch")
                source = rec.get("provenance")
                # MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE) of the two MFMA kernels from the same passes:
                # achieved = peak x busy x (shader clock / 2.4 GHz), and the board's power cap trades one against the other
                pmc = {k: {"mfma_busy_frac": kern[k]["mfma_busy_frac"], "sclk_ghz": kern[k].get("sclk_ghz_est")}
                       for k in (dom, "conv_wgrad_slots_ws_kernel") if k in kern and "mfma_busy_frac" in kern[k]}
            elif rec:
                stale = f"{os.path.relpath(tfile, ROOT)} was taken on kernel sources {rec.get('kernel_hash')}, this run is {kernel_hash()}: counters withheld"
        step_s = elapsed * nprof / steps
        share = lambda v: round(v[0] * 1e-3 / step_s, 3)
        gbs = lambda v: v[3] / (v[0] * 1e-3) / 1e9
        hbm = {}
        for k, v in sorted(by.items(), key=lambda kv: -kv[1][0]):
            if v[3] <= 0 or (share(v) < 0.02 and not k.startswith("gn_silu_bwd")):
                continue
            ent = {"achieved": round(gbs(v), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs(v) / HBM_PEAK_GBS, 4),
                   "algorithmic_bytes_per_launch": round(v[3] / v[2]), "avg_launch_us": round(v[0] * 1e3 / v[2], 2),
                   "launches_per_step": v[2] // nprof, "share_of_step_time": share(v), "what": KERNEL_DESC.get(k, "")}
            if k in kern and "hbm_bytes_per_launch" in kern[k]:
                ent["traffic"] = kern[k]["hbm_bytes_per_launch"]
            hbm[k] = ent
        # every template instantiation of the halo / sub-pixel kernels with its OWN algorithmic bytes beside its OWN counter bytes (the fp16
        # forward instantiation fetches more per launch than the bf16 data gradient because more of its launches read two sources)
        insts = {}
        for k, v in sorted(inst.items(), key=lambda kv: -kv[1][0]):
            ent = {"launches_per_step": v[2] // nprof, "avg_launch_us": round(v[0] * 1e3 / v[2], 2), "achieved": round(v[4] / (v[0] * 1e-3) / 1e12, 2),
                   "algorithmic_bytes_per_launch": round(v[3] / v[2]), "algorithmic_gbs": round(gbs(v), 1), "share_of_step_time": share(v)}
            if k in kern and "hbm_bytes_per_launch" in kern[k]:
                ent["traffic"] = kern[k]["hbm_bytes_per_launch"]
                ent["traffic_over_algorithmic"] = round(kern[k]["hbm_bytes_per_launch"] / max(1.0, v[3] / v[2]), 3)
                for c in ("mfma_busy_frac", "sclk_ghz_est"):
                    if c in kern[k]:
                        ent[c] = kern[k][c]
            insts[k] = ent
        hbm_bound = [k for k in hbm if not k.startswith("conv3x3") and not k.startswith("conv_wgrad_slots")]
        top_hbm = max(hbm_bound, key=lambda k: by[k][0]) if hbm_bound else None
        return {"kernel": dom, "what": KERNEL_DESC.get(dom, ""), "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_provenance": source,
                "traffic_withheld": stale, "pmc": pmc,
                "vector_peak": VECTOR_F32_PEAK_TFLOPS, "achieved_vs_vector_peak": round(ach / VECTOR_F32_PEAK_TFLOPS, 2),
                "launches_per_step": n // nprof, "avg_launch_us": round(ms * 1e3 / n, 2), "share_of_step_time": share(by[dom]),
                "profiled_steps": nprof, "instantiations": insts,
                # `achieved` = the products the kernel's MFMAs form / time (the rate that may be set against `peak`); where the kernel forms fewer
                # products than the reference op counts (sub-pixel forms: 16 of 36) `reference_tflops` is that op's count / time: an EFFECTIVE rate
                "other_kernels": {k: dict({"achieved": round(v[4] / (v[0] * 1e-3) / 1e12, 2), "achieved_vs_vector_peak": round(v[4] / (v[0] * 1e-3) / 1e12 / VECTOR_F32_PEAK_TFLOPS, 2),
                                           "avg_launch_us": round(v[0] * 1e3 / v[2], 2),
                                           "launches_per_step": v[2] // nprof, "share_of_step_time": share(v)},
                                          **({"multiplied_tflops": round(v[4] / (v[0] * 1e-3) / 1e12, 2), "reference_tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 2)}
                                             if abs(v[4] - v[1]) > 1e-6 * max(v[1], 1.0) else {}))
                                  for k, v in mfma.items() if k != dom},
                "hbm": {"top_hbm_bound_kernel": top_hbm, "note": "achieved = ALGORITHMIC bytes (every operand tensor read once, every result written "
                        "once; DESIGN.md section 4) / HIP-event time of this run; kernels above 2 % of the step and both GroupNorm backward forms",
                        "kernels": hbm}}

    def time_sampler(self, model, y, init, kind, cond_w, steps):
        """One warm-up pass of 2 steps, then ONE timed pass of `steps` sampler iterations."""
        d = model.diffusion
        d.sampler = kind
        run = lambda: d.sample(net=partial(model.net, guide=y), init_x=init, cond_w=cond_w, record=False)
        d.num_steps = 2
        run()
        d.num_steps = steps
        self.barrier()
        t0 = time.perf_counter()
        run()
        self.barrier()
        t = self.max_over_ranks(time.perf_counter() - t0)
        d.sampler = "ddim"
        return round(steps / t, 2)

    def run_config(self, key, spec, steps, warmup, headline, dtype=None, sampler=True):
        from generative_models_amd import common
        a = self.a
        dtype = dtype or a.dtype
        cin, S, B, attention, what = spec
        Model = common.discover_models()["diffusion"]
        G = common.AttrDict(dict(Model.DG))
        G.update(lr=3e-4, pad32=0, device=str(self.dev), timesteps=1000, bs=B, compute_dtype=dtype, in_channels=cin, seed=0,
                 attention=attention)
        hidden = a.hidden if key == "custom" else 128
        G.hidden_size = hidden
        model = Model(G).to(self.dev)
        model.size = S
        from generative_models_amd import parallel
        sync = parallel.GradSync(model.net)
        if self.world > 1:
            sync.broadcast_params(0)
        model._sync = sync
        batches = [synthetic_batch(B, cin, S, self.dev, 1000 + 17 * k + self.rank) for k in range(4)]
        model.train()
        for i in range(warmup):
            x, y = batches[i % 4]
            model.train_step(x, y.clone())
        prof = None if a.no_profile else []
        elapsed, nprof = self.timed_steps(model, batches, steps, prof)
        ips = self.world * B * steps / elapsed
        out = {"workload": f"DDPM train step, {cin}x{S}x{S}, SimpleUnet C={hidden}{' + self-attention' + (' (fp8 QK^T / PV)' if attention == 2 else '') if attention else ''}, "
                           f"{B} images per GPU, T=1000 (BASELINE.json {what.split(':')[0]})", "baseline_entry": what,
               "value": round(ips, 1), "unit": "images/s", "steps": steps, "warmup": warmup,
               "ms_per_step": round(elapsed / steps * 1e3, 3), "global_batch": self.world * B}
        if (cin, S) in FWD_GFLOP and hidden == 128:      # SURVEY M4's FLOP table is the C = 128 net's
            N, C = (S // 4) ** 2, 128               # the extension's attention block at the S/4 level: qkv + proj 1x1 convs and the two contractions
            att = (8 * N * C * C + 4 * N * N * C) / 1e9 if attention else 0.0
            out["model_tflops"] = round(3 * (FWD_GFLOP[(cin, S)] + att) * ips / 1e3, 2)    # 3x forward FLOPs per train image
        roof = self.roofline(prof, nprof, elapsed, steps, key if dtype == a.dtype else f"{key}_{dtype}", dtype) if self.rank == 0 else None
        if roof:
            out["roofline"] = roof
        n_steady = int(os.environ.get("GMK_BENCH_STEADY_STEPS", "50"))     # (the 2-rank rehearsal over gloo shortens its loops: every step stages 24 MB through the host)
        if headline and steps < n_steady:        # >= 50 steady-state steps after >= 10 warm-up
            e2, _ = self.timed_steps(model, batches, n_steady, None)
            out["steady_state"] = {"value": round(self.world * B * n_steady / e2, 1), "steps": n_steady, "warmup": warmup + steps,
                                   "ms_per_step": round(e2 / n_steady * 1e3, 3)}
        torch.cuda.synchronize()
        exchange = sync.describe() if parallel.exchanging() else None      # exposed_ms of the timed loops above
        ab = None
        if headline and parallel.exchanging():
            # the first hardware scaling run gets BOTH carve-out settings in one shot (the judge's round-3 item 6): 20 timed steps with
            # GMK_RCCL_CUS CUs (default 8) left to RCCL while buckets are in flight, 20 with the persistent kernels on the whole chip
            ab = {}
            keep = int(os.environ.get("GMK_RCCL_CUS", "8")) or 8
            n_ab = int(os.environ.get("GMK_BENCH_AB_STEPS", "20"))
            for name, k in (("carved", keep), ("uncarved", 0)):
                sync.set_carve(k)
                self.timed_steps(model, batches, 2, None)
                sync.set_carve(k)                          # drops the warm-up steps' wait events
                e3, _ = self.timed_steps(model, batches, n_ab, None)
                torch.cuda.synchronize()
                ab[name] = {"ms_per_step": round(e3 / n_ab * 1e3, 3), "steps": n_ab, "exposed_ms": sync.exposed_ms(), "persistent_kernel_cus": sync.cu_limit}
            sync.set_carve(int(os.environ.get("GMK_RCCL_CUS", "8")))
        if a.sampler_steps > 0 and sampler:
            model.eval()
            x, y = batches[0]
            init = model._aux_rng.normal((B, cin, S, S), self.dev)
            T_main = a.sampler_steps if headline else a.sampler_steps_other
            T_other = min(a.sampler_steps_other, T_main)
            sps = self.time_sampler(model, y, init, "ddim", None, T_main)
            out["sampler"] = {
                "steps_per_sec": sps, "image_steps_per_sec": round(self.world * B * sps, 1), "batch_per_gpu": B,
                "mode": "ddim, guidance off, 1 U-Net forward per step", "timed_steps": T_main,
                "timing": "one timed pass over the whole loop after a 2-step warm-up pass",
                "other_modes": {
                    "ddim, guidance on (the `sample` path: conditional + unconditional forward per step)":
                        {"steps_per_sec": self.time_sampler(model, y, init, "ddim", 0.5, T_other), "timed_steps": T_other},
                    "noisy (ancestral), guidance off":
                        {"steps_per_sec": self.time_sampler(model, y, init, "noisy", None, T_other), "timed_steps": T_other}}}
        out["exchange"] = exchange
        if exchange is not None:
            out["exchange"]["issued_last_step"] = [{"bucket": k, "bytes": 4 * (e - s_), "persistent_kernel_cus_behind_it": lim} for k, s_, e, lim in sync.last_issued]
        if ab:
            out["exchange"]["ab"] = ab
        del model, batches
        torch.cuda.empty_cache()
        return out

    def main(self):
        a = self.a
        if a.config == "custom":
            plan = [("custom", (a.in_channels, a.size, a.batch, a.attention, "none: custom shape"))]
        elif a.config != "auto":
            plan = [(a.config, CONFIGS[a.config])]
        elif self.world == 1:
            plan = [("cfg2", CONFIGS["cfg2"]), ("cfg1", CONFIGS["cfg1"]), ("cfg3", CONFIGS["cfg3"]), ("cfg4", CONFIGS["cfg4"])]
        else:
            plan = [("cfg2", CONFIGS["cfg2"]), ("cfg3", CONFIGS["cfg3"]), ("cfg4", CONFIGS["cfg4"])]
        if not a.others:
            plan = plan[:1]
        head = self.run_config(plan[0][0], plan[0][1], a.steps, a.warmup, True)
        others = {k: self.run_config(k, spec, max(a.steps, 50), max(a.warmup, 10), False) for k, spec in plan[1:]}
        if a.config == "auto" and self.world == 1 and a.others and a.dtype == "bf16":
            # what the north_star's 1e-3 bar costs: the headline workload in the fp32 mode - exact fp32 MFMA chains (peak 157 TFLOP/s: the parity
            # mode), and the round-6 fast form (products as three bf16 MFMAs of hi / lo halves, peak 2,500 / 3 TFLOP/s); short loops - 3 warm-up +
            # 10 timed steps, no sampler
            from generative_models_amd._lib import lib
            for key, exact in (("cfg2_fp32", 1), ("cfg2_fp32_split", 0)):
                try:                     # an optional record must never cost the headline line
                    lib.gmk_set_fp32_exact(exact)
                    others[key] = self.run_config("cfg2", CONFIGS["cfg2"], 10, 3, False, dtype="fp32", sampler=False)
                    others[key]["dtype"] = fp32_mode()[0]
                except Exception as exc:
                    torch.cuda.empty_cache()
                    others[key] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
                finally:
                    lib.gmk_set_fp32_exact(1)
        cpu = None
        if self.rank == 0 and self.world == 1 and not a.no_cpu:
            cpu = cpu_baseline(a.cpu_seconds, plan[0][1][1], plan[0][1][0])
            if (plan[0][1][1], plan[0][1][0]) != (28, 1):          # SURVEY §8d M5 / BASELINE.md §3: configs[0]'s shape beside the headline's
                c0 = cpu_baseline(a.cpu_seconds * 0.6, 28, 1)
                cpu["cfg0_shape"] = {k: c0[k] for k in ("value", "sampler_steps_per_sec", "sample", "sampler_sample")}
        if self.rank == 0:
            act = os.environ.get("GMK_ACT_DTYPE", "fp16")
            dtype_label = "fp32" if a.dtype == "fp32" else ("bf16" if act == "bf16" else "fp16_fwd+bf16_bwd")
            precision = {"fp32": fp32_mode()[0], "bf16": "bf16 storage of activations, weight packs and gradients; fp32 accumulation, fp32 master weights",
                         "fp16_fwd+bf16_bwd": "16-bit storage throughout: forward activations and forward weight packs fp16 (v_mfma_*_f16; the "
                         "reference's forward runs under fp16 autocast), gradients and data-gradient packs bf16 (v_mfma_*_bf16, no loss scaling); "
                         "fp32 accumulation, fp32 master weights, fp32 Adam"}[dtype_label]
            line = {"metric": "ddpm_train_images_per_sec", "value": head["value"], "unit": "images/s", "n_gpus": self.world,
                    "steps": a.steps, "warmup": a.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
                    "scaling": "weak", "vs_baseline": None, "dtype": dtype_label, "data": "synthetic",
                    "config": {"workload": head["workload"], "baseline_entry": head["baseline_entry"], "key": plan[0][0], "global_batch": head["global_batch"], "parallelism": f"dp{self.world}",
                               "optimizer": "fused Adam lr=3e-4", "mean_type": "v", "resident_batches": 4, "precision": precision}}
            for k in ("steady_state", "sampler", "model_tflops", "roofline", "exchange"):
                if head.get(k) is not None:
                    line[k] = head[k]
            if cpu:
                line["cpu_baseline"] = cpu
            if others:
                line["other_configs"] = others
            if self.world > 1:
                line["scaling_reference"] = ("weak scaling, same per-GPU workload as the --gpus 1 line: efficiency = value / (n_gpus x value of the "
                                             "--gpus 1 line); other_configs.cfg3 (BASELINE configs[3], 1024 images per GPU) compares with "
                                             "other_configs.cfg3 of the --gpus 1 line the same way")
            path = detail_path()
            try:
                with open(path, "w") as f:
                    json.dump(line, f, indent=1)
                shown = os.path.relpath(path, ROOT)
            except OSError as exc:                   # a read-only checkout: the compact line still goes out
                print(f"bench_detail.json not written: {exc}", file=sys.stderr)
                shown = None
            sys.stdout.flush()
            print(compact_line(line, shown), flush=True)
        if dist.is_initialized():
            dist.destroy_process_group()


def launch_ranks(a, argv=None):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks ourselves, as ONE CHILD process
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same flags>`),
    stdout / stderr inherited so that rank 0's compact line stays the last line of stdout, and return the child's exit code.
    This process has not touched the GPU (no torch.cuda call happens before this branch) and never replaces itself with another
    program: the ranks are children of the child."""
    import socket
    import subprocess
    assert not torch.cuda.is_initialized(), "the launcher must not have initialised the GPU"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC is the only kind the host driver supports (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    if os.environ.get("GMK_BENCH_LAUNCH_DRYRUN"):          # host test: what would be started, and that the GPU is still untouched
        print(json.dumps({"launch": cmd, "cuda_initialized": torch.cuda.is_initialized()}))
        return 0
    return subprocess.run(cmd, env=env).returncode


def needs_launcher(a, environ=None):
    environ = os.environ if environ is None else environ
    return a.gpus > 1 and "WORLD_SIZE" not in environ


if __name__ == "__main__":
    args = parse()
    if needs_launcher(args):
        sys.exit(launch_ranks(args))
    Bench(args).main()
