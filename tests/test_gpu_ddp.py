"""Data-parallel equivalence on the real HIP network: 2 ranks (both on cuda:0, gloo transport) each take half of a
batch; after the bucketed all-reduce + fused Adam(grad_scale=1/world) the parameters equal a single-rank step on the
whole batch.  RCCL itself runs with ONE rank (round 6, GMK_FORCE_EXCHANGE=1: communicator, the four bucket all-reduces on the exchange
stream, the carve-out); more than one rank needs more than one GPU and is the driver's scaling run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    """A TCP port nobody listens on right now: fixed rendezvous ports collide between test runs."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])

_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from functools import partial
from generative_models_amd import parallel
from generative_models_amd.diffusion.simple_unet import SimpleUnet
from generative_models_amd.diffusion.gaussian_diffusion import GaussianDiffusion
from generative_models_amd.diffusion.optim import FusedAdam
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
torch.manual_seed(0)
g = torch.Generator().manual_seed(0)
B = 4
x = (torch.rand((B, 1, 12, 12), generator=g) * 2 - 1).cuda(); y = torch.randint(0, 10, (B,), generator=g).cuda()
u = torch.rand((B,), generator=g).cuda(); eps = torch.randn((B, 1, 12, 12), generator=g).cuda()
def make():
    torch.manual_seed(1)
    net = SimpleUnet(128, 0.0, compute_dtype=torch.float32)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if ".out_layers.3.weight" in n:
                p.uniform_(-0.02, 0.02)
    return net.cuda()
# single-rank reference on the whole batch
ref = make(); dref = GaussianDiffusion(mean_type="v", num_steps=4); oref = FusedAdam(ref)
dref.train_forward_backward(net=partial(ref, guide=y), x=x, grad_scale=1.0 / B, u=u, eps=eps)
gref = ref.flat_grads.clone()
oref.step()
# sharded
net = make(); sync = parallel.GradSync(net); sync.broadcast_params(0)
d = GaussianDiffusion(mean_type="v", num_steps=4); opt = FusedAdam(net)
sl = slice(r * B // w, (r + 1) * B // w)
d.train_forward_backward(net=partial(net, guide=y[sl]), x=x[sl], grad_scale=1.0 / (B // w), u=u[sl], eps=eps[sl],
                         on_grads_ready=sync.hook, join_side_before_ready=False)      # as DiffusionModel.train_step calls it
sync.finish()
torch.cuda.synchronize()
assert sync.exposed_ms() is not None and sync._comm is not None
# mean over ranks of the per-shard mean gradients == gradient of the whole-batch mean (equal shards)
gdiff = float((net.flat_grads / w - gref).abs().max()) / float(gref.abs().max())
assert gdiff < 1e-4, gdiff
opt.step(grad_scale=1.0 / w)
gerr = float((net.flat_params - ref.flat_params).abs().max())
delta = float((ref.flat_params - make().flat_params).abs().max())
# the first Adam step is ~ lr * sign(g): elements whose gradient is rounding noise may differ by a fraction of lr
assert delta > 1e-5 and gerr < 0.1 * delta, (gerr, delta)
dist.destroy_process_group()
print("rank", r, "ok", gerr, delta)
"""


def test_two_rank_step_equals_single_rank(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", free_port(), str(script), ROOT]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2


@pytest.mark.parametrize("form", ["torch.distributed.run", "plain"])
def test_bench_two_ranks_rehearsal(form):
    """Both of the driver's scaling command lines with N = 2 ranks sharing this GPU over gloo (RCCL refuses two ranks on one
    device): `torch.distributed.run ... bench.py --gpus N`, and plain `python bench.py --gpus N` (no WORLD_SIZE: bench.py starts
    its own ranks as a child process).  Barrier, bucketed all-reduce from inside backward, max-over-ranks timing, one JSON line
    from rank 0 with the whole-job rate."""
    import json
    flags = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "custom", "--batch", "64", "--sampler_steps", "2",
             "--sampler_steps_other", "2", "--no_cpu"]
    if form == "plain":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + flags
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", free_port(), os.path.join(ROOT, "bench.py")] + flags
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(env, GMK_DIST_BACKEND="gloo", OMP_NUM_THREADS="2", GMK_BENCH_STEADY_STEPS="6", GMK_BENCH_AB_STEPS="4"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert r.stdout.strip().splitlines()[-1] == lines[0] and len(lines[0]) <= 4096, len(lines[0])    # the LAST stdout line, within the driver's window
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    assert d["value"] > 0 and "cpu_baseline" not in d and d["roofline"]["kernel"].startswith("conv")
    ex = d["exchange"]
    assert ex["world"] == 2 and ex["backend"] == "gloo" and len(ex["bucket_bytes"]) == 4 and ex["persistent_kernel_cus"] == 248
    assert ex["exposed_ms"] is not None and ex["exposed_ms"] >= 0.0
    assert "rccl_max_channels" not in ex         # the RCCL channel cap is opt-in (GMK_RCCL_CAP=1) until a node has measured it
    # both carve-out settings in one run, so the first hardware scaling curve can be read with either
    ab = ex["ab"]
    assert ab["carved"]["persistent_kernel_cus"] == 248 and ab["uncarved"]["persistent_kernel_cus"] is None
    assert ab["carved"]["ms_per_step"] > 0 and ab["uncarved"]["ms_per_step"] > 0 and ab["uncarved"]["exposed_ms"] is not None
    assert d["steady_state"]["steps"] == 6 and d["sampler"]["timed_steps"] == 2          # (loops shortened for the rehearsal: gloo stages every bucket through the host)
    full = json.load(open(os.path.join(ROOT, d["detail"])))
    assert full["value"] == d["value"] and "hbm" in full["roofline"] and full["exchange"]["ab"] == ab


_RCCL_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from functools import partial
from generative_models_amd import parallel
from generative_models_amd._lib import lib
from generative_models_amd.diffusion.simple_unet import SimpleUnet
from generative_models_amd.diffusion.gaussian_diffusion import GaussianDiffusion
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)            # "nccl" IS RCCL on ROCm: loads librccl, one-rank communicator
assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
g = torch.Generator().manual_seed(0)
B, S = 256, 28
x = (torch.rand((B, 1, S, S), generator=g) * 2 - 1).cuda(); y = torch.randint(0, 10, (B,), generator=g).cuda()
u = torch.rand((B,), generator=g).cuda(); eps = torch.randn((B, 1, S, S), generator=g).cuda()
torch.manual_seed(1)
net = SimpleUnet(128, 0.0, compute_dtype=torch.bfloat16)
with torch.no_grad():
    for n, p in net.named_parameters():
        if ".out_layers.3.weight" in n:
            p.uniform_(-0.02, 0.02)
net = net.cuda()
d = GaussianDiffusion(mean_type="v", num_steps=1000)
full = lib.gmk_get_cu_limit()
def step(force, keep):
    os.environ["GMK_FORCE_EXCHANGE"] = "1" if force else "0"
    os.environ["GMK_RCCL_CUS"] = str(keep)
    sync = parallel.GradSync(net)
    out = d.train_forward_backward(net=partial(net, guide=y), x=x, grad_scale=1.0 / B, u=u, eps=eps,
                                   on_grads_ready=sync.hook, join_side_before_ready=False)      # as DiffusionModel.train_step calls it
    inside = lib.gmk_get_cu_limit()
    issued = list(sync.issued)
    sync.finish()
    torch.cuda.synchronize()
    return out["loss"].clone(), net.flat_grads.clone(), sync, inside, issued
l0, g0, s0, in0, is0 = step(False, 8)
assert not is0 and s0._comm is None and in0 == full                                  # unforced, one rank: no exchange at all
# forced, nothing carved: the four bucket all-reduces (a one-rank sum is the identity) from the exchange stream - the SAME bits
l1, g1, s1, in1, is1 = step(True, 0)
assert [k for k, _, _ in is1] == [0, 1, 2, 3] and s1._comm is not None and in1 == full
assert sum(e - s for _, s, e in is1) == net.flat_grads.numel()                        # the buckets tile the arena
assert torch.equal(l0, l1) and torch.equal(g0, g1)
assert s1.exposed_ms() is not None and s1.exposed_ms() >= 0.0
# forced, 8 CUs left to RCCL while buckets fly: the limit is 248 behind every all-reduce, the full chip again after finish();
# kernels launched under the carve-out split their fp32 sums differently (documented): same values to rounding
l2, g2, s2, in2, is2 = step(True, 8)
assert in2 == full - 8 and lib.gmk_get_cu_limit() == full
assert [lim for _, _, _, lim in s2.last_issued] == [full - 8] * 4
assert torch.equal(l0, l2)
rel = float((g2 - g0).abs().max() / g0.abs().max())
assert rel < 1e-4, rel
info = s2.describe()
assert info["forced"] and info["backend"] == "nccl" and info["world"] == 1 and info["persistent_kernel_cus"] == full - 8
ver = info["rccl_version"]
assert ver[0].isdigit() and ver.count(".") >= 2, ver
dist.destroy_process_group()
print("rccl one-rank ok", ver, "exposed_ms", s2.exposed_ms(), "rel", rel)
"""


def test_rccl_runs_the_bucketed_exchange_with_one_rank(tmp_path):
    """RCCL on the one GPU there is (round 6): a one-rank `nccl` process group under `torch.distributed.run --nproc-per-node 1` with
    GMK_FORCE_EXCHANGE=1 - communicator created, the four readiness-ordered buckets all-reduced from the exchange stream behind the
    data-gradient and weight-gradient streams, the persistent kernels under the carved CU limit while they fly (248 inside, 256 after),
    gradients bit-equal to the unforced step where nothing is carved.  What it cannot show is xGMI traffic: there is one rank."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(_RCCL_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", free_port(), str(script), ROOT]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GMK_CU_LIMIT")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "rccl one-rank ok" in r.stdout


def test_bench_forced_exchange_on_one_gpu():
    """`GMK_FORCE_EXCHANGE=1 python bench.py --gpus 1`: the N = 1 line with an `exchange` block measured over RCCL (one rank) - both carve-out
    settings, `exposed_ms`, the RCCL version - so that the overlap machinery of the N > 1 line has run on hardware before a node is available."""
    import json
    flags = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--config", "custom", "--batch", "256", "--sampler_steps", "0", "--no_cpu"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + flags, capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(env, GMK_FORCE_EXCHANGE="1", OMP_NUM_THREADS="2", GMK_BENCH_STEADY_STEPS="6", GMK_BENCH_AB_STEPS="4"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    ex = d["exchange"]
    assert d["n_gpus"] == 1 and ex["world"] == 1 and ex["forced"] and ex["backend"] == "nccl" and ex["persistent_kernel_cus"] == 248
    assert ex["rccl_version"][0].isdigit() and ex["exposed_ms"] is not None
    assert ex["ab"]["carved"]["persistent_kernel_cus"] == 248 and ex["ab"]["uncarved"]["persistent_kernel_cus"] is None
    full = json.load(open(os.path.join(ROOT, d["detail"])))
    assert [b["bucket"] for b in full["exchange"]["issued_last_step"]] == [0, 1, 2, 3]
