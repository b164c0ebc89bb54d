"""Which STAGE of gn_silu_fwd_reg_kernel differs between two identical forwards?  Needs
    tools/build_variant.sh stage -DGMK_GN_XCHECK=3 -DGMK_SHFL_BPERMUTE
    GMK_LIBGMK=generative_models_amd/libgmk_stage.so python tools/gn_stage.py [runs]
Every workgroup of that kernel leaves four checksums (after its last store, so the kernel's timing is otherwise the shipped one):
loaded input dwords | per-lane partial sums | reduced totals handed to LDS | group statistics read back from LDS.  The records of
each forward are compared with those of the first forward."""
import ctypes, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from generative_models_amd import ops
ops.FWD_SIDE = True          # the overlap under investigation (off by default in the product)
from generative_models_amd._lib import lib
from generative_models_amd.diffusion.simple_unet import SimpleUnet

assert hasattr(lib, "gmk_debug_gn_stage"), "needs a -DGMK_GN_XCHECK=3 build (see the docstring)"
NS, NB = 64, 4096
def readout():
    buf = (ctypes.c_uint * (NS * NB * 4))()
    torch.cuda.synchronize()
    assert lib.gmk_debug_gn_stage(buf, 1) == 0
    rec = np.frombuffer(buf, dtype=np.uint32).reshape(NS, NB, 4).copy()
    late = (ctypes.c_uint * (NS * NB))()
    assert lib.gmk_debug_gn_late(late, 1) == 0
    late = np.frombuffer(late, dtype=np.uint32).reshape(NS, NB)
    if late.any():          # -DGMK_GN_XSTAGE_EARLY=1 build: is the input checksum taken at the kernel's end still the one taken right after the loads?
        bad = np.argwhere(late != rec[:, :, 0])
        if len(bad):
            print(f"    REGISTERS CHANGED inside the kernel: {len(bad)} workgroups whose loaded-input checksum at the end differs from the one taken right after the loads, e.g. slot/block {bad[:4].tolist()}")
    return rec

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
torch.manual_seed(0)
net = SimpleUnet(128, 0.0).cuda().eval()
with torch.no_grad():
    for n, p in net.named_parameters():
        if ".out_layers.3.weight" in n:
            p.uniform_(-0.02, 0.02)
net.mark_params_changed()
B = 1024
z = torch.randn(B, 1, 28, 28, device="cuda"); l = torch.randn(B, device="cuda"); y = torch.randint(0, 10, (B,), device="cuda")
names = ["loaded input", "per-lane partial sums", "reduced totals -> LDS", "group statistics <- LDS"]
readout()
ref_out, ref_rec, bad = None, None, 0
for r in range(runs):
    out = net.forward_hip(z, l, y, None).clone()
    rec = readout()
    if ref_out is None:
        ref_out, ref_rec = out, rec
        print("records in use:", int((rec != 0).any(-1).sum()), "of", NS * NB)
        continue
    same = bool(torch.equal(out, ref_out))
    diff = np.argwhere(rec != ref_rec)
    if not same or len(diff):
        bad += 1
        stages = sorted({int(d[2]) for d in diff})
        print(f"run {r}: output {'same' if same else 'DIFFERS'}; {len(diff)} differing checksums; stages {[names[s] for s in stages]}")
        for slot, blk, st in diff[:6]:
            print(f"    slot {slot} block {blk}: {names[st]}: {rec[slot, blk, st]:#010x} vs {ref_rec[slot, blk, st]:#010x}")
print(f"{bad} of {runs - 1} forwards differ from the first")

# ---- second pass: WHICH call, and is its input's final content different too (producer) or only what the kernel loaded (visibility)?
print("\nper-call snapshots (input clone taken in stream order right behind each GroupNorm launch):")
orig = ops.gn_silu_fwd
calls = []
def wrapped(x, gamma, beta, groups, **kw):
    y, mean, rstd = orig(x, gamma, beta, groups, **kw)
    calls.append((x.clone(), mean.clone(), tuple(x.shape), ((mean.data_ptr() >> 9) & 63), x.data_ptr()))
    return y, mean, rstd
ops.gn_silu_fwd = wrapped
import generative_models_amd.diffusion.simple_unet as su
ref_calls = None
for r in range(runs):
    calls.clear()
    out = net.forward_hip(z, l, y, None).clone()
    rec = readout()
    torch.cuda.synchronize()
    if ref_calls is None:
        ref_calls, ref_rec2, ref_out2 = list(calls), rec, out
        continue
    if torch.equal(out, ref_out2):
        continue
    diff = np.argwhere(rec != ref_rec2)
    print(f"run {r}: output differs; checksum diffs at slots/blocks {sorted({(int(a), int(b)) for a, b, _ in diff})[:4]}")
    for ci, (c, c0) in enumerate(zip(calls, ref_calls)):
        xin_same, mean_same = bool(torch.equal(c[0], c0[0])), bool(torch.equal(c[1], c0[1]))
        if not xin_same or not mean_same:
            bad_b = (c[1] != c0[1]).any(1).nonzero().flatten().tolist()[:4]
            print(f"    call {ci:2d} x{c[2]} slot {c[3]}: input clone {'same' if xin_same else 'DIFFERS'}, mean {'same' if mean_same else 'DIFFERS'} (samples {bad_b})")
            for b in bad_b:                      # does the wrong statistic equal that of ANOTHER call's input (a stale kernel argument)?
                hits = [cj for cj, o in enumerate(ref_calls) if o[1].shape == c[1].shape and bool(torch.equal(o[1][b], c[1][b]))]
                print(f"             mean[{b}] = {c[1][b, :3].tolist()} (reference {c0[1][b, :3].tolist()}); bitwise equal to the mean of call(s) {hits}")
            if not xin_same:
                d = (c[0] != c0[0]).flatten(1).any(1).nonzero().flatten().tolist()[:6]
                print(f"             input differs in samples {d}")
            break
