"""In-process A/B of whole train steps under different kernel choices (same device, interleaved rounds).
    python tools/step_ab.py "0,0,0" "2,0,0" ...      (conv,wgrad,gn choices; see include/gmk.h)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from generative_models_amd import common, ops
from generative_models_amd._lib import lib
import bench
variants = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(0, 0, 0)]
Model = common.discover_models()["diffusion"]
G = common.AttrDict(dict(Model.DG)); G.update(lr=3e-4, pad32=0, device="cuda", timesteps=1000, bs=1024)
model = Model(G).to("cuda")
x, y = bench.synthetic_batch(1024, 1, 28, "cuda", 1000)
res = {v: [] for v in variants}
for rnd in range(4):
    for v in variants:
        lib.gmk_set_kernel_choice(*v[:3])
        ops.GN_STATS = (v[3] != 0) if len(v) > 3 else False
        ops.WGRAD_STREAM = (v[4] != 0) if len(v) > 4 else True
        lib.gmk_set_dev_variant(v[5] if len(v) > 5 else 0)
        for _ in range(2):
            model.train_step(x, y.clone())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            model.train_step(x, y.clone())
        torch.cuda.synchronize(); res[v].append((time.perf_counter() - t0) / 8 * 1e3)
for v in variants:
    r = sorted(res[v]); print(v, "ms/step min %.2f median %.2f" % (r[0], r[len(r) // 2]), ["%.2f" % t for t in res[v]])
