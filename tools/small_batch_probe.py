"""(GMK_SAMPLER_GRAPH_PIXELS=0 in the environment: kernel-by-kernel launches instead of the replayed HIP graph.)
Small-batch sampling latency - the reference's own use of the sampler (`evaluate`: 25 images, diffusion_model.py:98-104; `eval_heavy`: chunks
of the test batch size): DDIM steps/s at B images of 1x28x28, interleaved over the variants given as GMK_DEV_VARIANT values.
    python tools/small_batch_probe.py [B=25] [steps=200] [variants=0,8]"""
import sys, time
import torch
sys.path.insert(0, ".")
from functools import partial
from generative_models_amd import common
from generative_models_amd._lib import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
variants = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,8").split(",")]
Model = common.discover_models()["diffusion"]
G = common.AttrDict(dict(Model.DG)); G.update(lr=3e-4, pad32=0, device="cuda", timesteps=n, bs=B)
m = Model(G).cuda().eval(); m.size = 28
y = torch.arange(B, device="cuda") % 10
init = m._aux_rng.normal((B, 1, 28, 28), "cuda")
def run(v, steps):
    lib.gmk_set_dev_variant(v)
    try:
        m.diffusion.num_steps = steps
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m.diffusion.sample(net=partial(m.net, guide=y), init_x=init, record=False)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out
    finally:
        lib.gmk_set_dev_variant(0)
outs = {}
for v in variants:
    run(v, 5)
for rep in range(3):
    for v in variants:
        dt, outs[v] = run(v, n)
        print(f"B={B} variant {v}: {n / dt:8.1f} steps/s  ({dt / n * 1e3:.3f} ms per step)", flush=True)
ref = outs[variants[0]][0][-1] if isinstance(outs[variants[0]], tuple) else outs[variants[0]]
for v in variants[1:]:
    o = outs[v][0][-1] if isinstance(outs[v], tuple) else outs[v]
    print(f"variant {v} vs {variants[0]}: max abs difference of the final images {float((o - ref).abs().max()):.3e}")
